"""Oracle self-checks for the gauge side of the MD step: the conventions are fixed by requiring that
the forces are the derivatives of the actions and that H = K + S_g (+ S_f) is conserved; nothing else enters."""
import numpy as np
import pytest

KAPPA = 0.141139
BC = (1, 1, 1, -1)


def _herm(rng):
    m = rng.standard_normal((3, 3)) + 1j * rng.standard_normal((3, 3))
    return 0.5 * (m + m.conj().T)


def test_gauge_force_is_the_derivative_of_the_gauge_action(orc):
    from scipy.linalg import expm
    L, beta = (4, 4, 4, 4), 5.7
    U = orc.hot_gauge(L, 201)
    G = orc.gauge_force(U, L, beta)
    rng = np.random.default_rng(202)
    eps = 1e-5
    for (mu, t, z, y, x) in [(0, 1, 2, 3, 0), (3, 3, 0, 1, 3), (1, 0, 3, 0, 2), (2, 2, 1, 3, 1)]:
        T = _herm(rng)
        vals = []
        for sgn in (1, -1):
            Up = U.copy()
            Up[mu, t, z, y, x] = (expm(1j * sgn * eps * T) @ U[mu, t, z, y, x].T).T
            vals.append(orc.gauge_action(Up, L, beta))
        fd = (vals[0] - vals[1]) / (2 * eps)
        an = -2.0 * np.trace(T @ G[mu, t, z, y, x].T).imag
        assert abs(fd - an) < 1e-6 * max(1.0, abs(an)) and abs(an) > 1e-3


def test_link_update_is_unitary_and_reversible(orc):
    L = (4, 4, 2, 2)
    U = orc.hot_gauge(L, 203)
    P = orc.gaussian_momenta(L, 204)
    assert abs(orc.momentum_action(P, L) / (4 * 64 * 8) - 0.5) < 0.1        # <pi^2>/2 per degree of freedom
    U1 = orc.link_update(U.copy(), P, 0.3, L)
    assert orc.unitarity_dev(U1, L) < 1e-14
    U2 = orc.link_update(U1.copy(), P, -0.3, L)
    assert np.abs(U2 - U).max() < 1e-14
    # traceless anti-Hermitian projection: idempotent, and P stays in the algebra
    G = orc.gauge_force(U, L, 5.7)
    Q = orc.momentum_add_ta(np.zeros_like(G), 1.0, G, L)
    Qm = np.swapaxes(Q, -1, -2)
    assert np.abs(Qm + Qm.conj().swapaxes(-1, -2)).max() < 1e-15 and np.abs(np.trace(Qm, axis1=-2, axis2=-1)).max() < 1e-15
    assert np.abs(orc.momentum_add_ta(np.zeros_like(G), 1.0, Q, L) - Q).max() < 1e-15


def _leapfrog(orc, U, P, L, beta, dt, nsteps, eta=None):
    """QPQ leapfrog for H = K + S_g (+ S_f with fixed pseudofermion eta)."""
    def force(U):
        G = orc.gauge_force(U, L, beta)
        if eta is not None:
            S, X, Y, it, st = orc.fermi_action(orc.WILSON, U, eta, L, KAPPA, bc=BC, eps=1e-24)
            G = G + orc.fermion_force(orc.WILSON, U, X, Y, L, KAPPA, bc=BC)
        return G
    U, P = U.copy(), P.copy()
    orc.link_update(U, P, 0.5 * dt, L)
    for k in range(nsteps):
        orc.momentum_add_ta(P, dt, force(U), L)
        orc.link_update(U, P, dt if k < nsteps - 1 else 0.5 * dt, L)
    return U, P


def _H(orc, U, P, L, beta, eta=None):
    H = orc.momentum_action(P, L) + orc.gauge_action(U, L, beta)
    if eta is not None:
        H += orc.fermi_action(orc.WILSON, U, eta, L, KAPPA, bc=BC, eps=1e-24)[0]
    return H


@pytest.mark.parametrize("dynamical", [False, True])
def test_energy_conservation_and_reversibility(orc, lq, dynamical):
    """dH scales like dt^2 for the leapfrog, and the trajectory retraces itself with the momenta flipped."""
    L, beta = (4, 4, 4, 4), 5.7
    import os
    from conftest import GOLDEN
    U = lq.gauge_io.load_ildg(os.path.join(GOLDEN, "wilson_4x4x4x4.ildg"), L)      # the reference's thermalised configuration
    P = orc.gaussian_momenta(L, 205)
    eta = None
    if dynamical:
        xi = orc.gaussian_spinor(orc.wilson_shape(L), 206)
        eta = orc.wilson_D(U, xi, L, KAPPA, 1.0, BC, dagger=True)
    H0 = _H(orc, U, P, L, beta, eta)
    dH = []
    for nsteps in (10, 20):
        U1, P1 = _leapfrog(orc, U, P, L, beta, 0.5 / nsteps, nsteps, eta)
        dH.append(_H(orc, U1, P1, L, beta, eta) - H0)
    assert abs(dH[0]) < 3.0 and abs(dH[1]) < abs(dH[0])         # of an H of about 5000
    assert 3.0 < abs(dH[0] / dH[1]) < 5.0                       # 4 for a second-order integrator
    U2, P2 = _leapfrog(orc, U1, -P1, L, beta, 0.025, 20, eta)
    assert np.abs(U2 - U).max() < 1e-9 and np.abs(P2 + P).max() < 1e-9


def test_oracle_hmc_repeats_the_reference_wilson_test(orc, lq):
    """Pins the ORACLE chain (Dslash, CG, fermion force, gauge force, integrator) to the reference's own end-to-end golden:
    test/runtests.jl:88-99 with test/test_wilson.toml -- start from the reference's thermalised 4^4 configuration, beta = 5.7,
    kappa = 0.141139, dtau = 0.05, 20 MD steps, Sexton-Weingarten N = 10, Nsteps = 10 trajectories; the final plaquette must be
    within 10 % of test/debugplaqdata.txt line 7.  (The criterion is loose but not empty: tests/test_gpu_md.py shows that a
    pseudofermion weight of exp(-S_f/2) fails it.)"""
    import os
    from conftest import GOLDEN
    L, beta, dtau, mdsteps, nsw = (4, 4, 4, 4), 5.7, 0.05, 20, 10
    ref_plaq = 0.5784043949012552                                   # /root/reference/test/debugplaqdata.txt:7
    U = lq.gauge_io.load_ildg(os.path.join(GOLDEN, "wilson_4x4x4x4.ildg"), L)
    rng = np.random.default_rng(111)
    dHs, acc = [], 0

    def Sf_and_force(U, eta, want_force=True):
        S, X, Y, it, st = orc.fermi_action(orc.WILSON, U, eta, L, KAPPA, bc=BC, eps=1e-19)
        assert st == 0
        return S, (orc.fermion_force(orc.WILSON, U, X, Y, L, KAPPA, bc=BC) if want_force else None)

    for traj in range(10):
        Uold = U.copy()
        P = orc.gaussian_momenta(L, 1000 + traj)
        xi = orc.gaussian_spinor(orc.wilson_shape(L), 2000 + traj) * np.sqrt(0.5)     # exp(-xi'xi): <|xi_i|^2> = 1
        eta = orc.wilson_D(U, xi, L, KAPPA, 1.0, BC, dagger=True)
        Hold = orc.momentum_action(P, L) + orc.gauge_action(U, L, beta) + np.vdot(xi, xi).real
        for _ in range(mdsteps):                                                      # runMD_QPQ_sw! (standardMD.jl:146-166)
            for half in range(2):
                for _ in range(nsw // 2):
                    orc.link_update(U, P, 0.5 / nsw * dtau, L)
                    orc.momentum_add_ta(P, dtau / nsw, orc.gauge_force(U, L, beta), L)
                    orc.link_update(U, P, 0.5 / nsw * dtau, L)
                if half == 0:
                    orc.momentum_add_ta(P, dtau, Sf_and_force(U, eta)[1], L)
        Hnew = orc.momentum_action(P, L) + orc.gauge_action(U, L, beta) + Sf_and_force(U, eta, False)[0]
        dHs.append(Hnew - Hold)
        if np.exp(-(Hnew - Hold)) >= rng.random():
            acc += 1
        else:
            U = Uold
    plaq = orc.plaquette(U, L)
    assert abs(plaq - ref_plaq) / ref_plaq < 0.1, (plaq, dHs)
    assert acc >= 6 and np.abs(dHs).max() < 2.0, (acc, dHs)


def test_oracle_hmc_repeats_the_reference_quenched_su3_test(orc, lq):
    """The gauge side alone, pinned to the reference's golden: test/runtests.jl:31-38 with test/test01.toml -- quenched SU(3) HMC from
    the reference's thermalised 4^4 configuration, beta = 5.7, dtau = 1/15, 15 MD steps (plain QPQ leapfrog), 10 trajectories; final
    plaquette within 10 % of test/debugplaqdata.txt line 2."""
    import os
    from conftest import GOLDEN
    L, beta, dtau, mdsteps = (4, 4, 4, 4), 5.7, 1.0 / 15.0, 15
    ref_plaq = 0.55783720583739                                     # /root/reference/test/debugplaqdata.txt:2
    U = lq.gauge_io.load_ildg(os.path.join(GOLDEN, "quenched_su3_4x4x4x4.ildg"), L)
    start = orc.plaquette(U, L)
    rng = np.random.default_rng(112)
    dHs, acc = [], 0
    for traj in range(10):
        Uold = U.copy()
        P = orc.gaussian_momenta(L, 3000 + traj)
        Hold = orc.momentum_action(P, L) + orc.gauge_action(U, L, beta)
        for _ in range(mdsteps):
            orc.link_update(U, P, 0.5 * dtau, L)
            orc.momentum_add_ta(P, dtau, orc.gauge_force(U, L, beta), L)
            orc.link_update(U, P, 0.5 * dtau, L)
        dH = orc.momentum_action(P, L) + orc.gauge_action(U, L, beta) - Hold
        dHs.append(dH)
        if np.exp(-dH) >= rng.random():
            acc += 1
        else:
            U = Uold
    plaq = orc.plaquette(U, L)
    assert abs(plaq - ref_plaq) / ref_plaq < 0.1, (plaq, start, dHs)
    assert acc >= 6 and np.abs(dHs).max() < 2.0 and abs(plaq - start) > 1e-6, (acc, dHs)
