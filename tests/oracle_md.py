"""The CPU oracle's molecular-dynamics trajectory: the integrators of the reference's StandardMD (QPQ leapfrog, its Sexton-Weingarten form, PQP:
/root/reference/src/md/standardMD.jl:126-190) restated on the oracle's own primitives -- link_update, gauge_force, momentum_add_ta, fermi_action,
fermion_force (oracle/oracle.py:146-215) -- as a table of stages.  Test infrastructure: what the device runs through the replayed callers is compared with this."""
import numpy as np


def stages(scheme, nsw=2):
    """One MD step as a list of (leg, coefficient of dtau): "U" = links move with the momenta, "G" / "F" = momenta move with the gauge / fermion force."""
    if scheme == "QPQ":
        return [("U", 0.5), ("G", 1.0), ("F", 1.0), ("U", 0.5)]
    if scheme == "PQP":
        return [("G", 0.5), ("F", 0.5), ("U", 1.0), ("G", 0.5), ("F", 0.5)]
    if scheme == "QPQ_sw":      # the gauge force on a finer time scale around one fermion kick
        inner = [("U", 0.5 / nsw), ("G", 1.0 / nsw), ("U", 0.5 / nsw)] * (nsw // 2)
        return inner + [("F", 1.0)] + inner
    raise ValueError(scheme)


def trajectory(orc, U, P, L, beta, dtau, mdsteps, scheme="QPQ", nsw=2, fermion=None):
    """Evolves copies of (U, P).  fermion = (kind, kappa-or-mass, bc, eta, eps) switches the pseudofermion force on (None: quenched)."""
    U, P = U.copy(), P.copy()
    for _ in range(mdsteps):
        for leg, c in stages(scheme, nsw):
            if leg == "U":
                orc.link_update(U, P, c * dtau, L)
            elif leg == "G":
                orc.momentum_add_ta(P, c * dtau, orc.gauge_force(U, L, beta), L)
            elif fermion is not None:
                kind, km, bc, eta, eps = fermion
                S, X, Y, it, st = orc.fermi_action(kind, U, eta, L, km, bc=bc, eps=eps)
                assert st == 0
                orc.momentum_add_ta(P, c * dtau, orc.fermion_force(kind, U, X, Y, L, km, bc=bc), L)
    return U, P


def hamiltonian(orc, U, P, L, beta, fermion=None):
    H = orc.momentum_action(P, L) + orc.gauge_action(U, L, beta)
    if fermion is not None:
        kind, km, bc, eta, eps = fermion
        H += orc.fermi_action(kind, U, eta, L, km, bc=bc, eps=eps)[0]
    return H
