"""Seeded fuzz of the stencil against the oracle: random even lattice shapes (down to extent 2), boundary signs, Wilson
parameter, kernel variants and workgroup maps, both operators, both daggers, parity hops and the 12-real option."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(seed)
    L = tuple(int(2 * rng.integers(1, 7)) for _ in range(4))
    if L[0] * L[1] * L[2] * L[3] > 20000:
        L = (L[0], L[1], 2, 4)
    bc = tuple(int(rng.choice([-1, 1])) for _ in range(4))
    return dict(L=L, bc=bc, r=float(rng.choice([1.0, 1.0, 0.6])), kappa=float(rng.uniform(0.05, 0.15)), mass=float(rng.uniform(0.05, 1.0)),
                variant=int(rng.integers(0, 9)), remap=int(rng.integers(0, 3)), nsub=int(rng.choice([8, 16, 32])), ysplit=int(rng.choice([1, 2, 4])),
                block=int(rng.choice([64, 128, 256])), recon=int(rng.choice([18, 12])), dagger=bool(rng.integers(0, 2)), seed=seed,
                stag_both=int(rng.integers(0, 2)))


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_configuration_matches_oracle(lq, orc, seed):
    assert lq.lib.device_count() > 0
    k = _case(seed)
    L, bc = k["L"], k["bc"]
    lat = lq.Lattice(L)
    for key, val in (("dslash_variant", k["variant"]), ("xcd_remap", k["remap"]), ("xcd_nsub", k["nsub"]), ("xcd_ysplit", k["ysplit"]),
                     ("dslash_block", k["block"]), ("gauge_recon", k["recon"]), ("stag_both", k["stag_both"])):
        lat.set_param(key, val)
    Uh = orc.hot_gauge(L, 1000 + seed)
    U = lq.Gaugefields(lat).upload(Uh)
    for name, kind, okind, km in (("Wilson", lq.WILSON, orc.WILSON, k["kappa"]), ("Staggered", lq.STAGGERED, orc.STAGGERED, k["mass"])):
        r = k["r"] if kind == lq.WILSON else 1.0
        D = lq.Dirac_operator(U, None, {"Dirac_operator": name, "κ": k["kappa"], "mass": k["mass"], "r": r, "boundarycondition": bc})
        psi = orc.gaussian_spinor(lat.fermion_shape(kind), 2000 + seed)
        x = lq.Fermionfields(lat, kind).upload(psi)
        y = x.similar()
        lq.mul_(y, D.adjoint() if k["dagger"] else D, x)
        ref = orc.apply_D(okind, Uh, psi, L, km, r, bc, k["dagger"])
        assert rel_err(y.download(), ref) < 1e-13, k
        lq.mul_(y, lq.DdagD_operator(D), x)
        ref2 = orc.apply_D(okind, Uh, orc.apply_D(okind, Uh, psi, L, km, r, bc, False), L, km, r, bc, True)
        assert rel_err(y.download(), ref2) < 1e-13, k
        if kind == lq.WILSON:
            p = seed & 1
            xin = lq.Fermionfields(lat, kind, lq.ODD if p == 0 else lq.EVEN).upload(psi)
            yout = lq.Fermionfields(lat, kind, lq.EVEN if p == 0 else lq.ODD)
            lq.hop_(yout, D.adjoint() if k["dagger"] else D, xin)
            assert rel_err(yout.download(), orc.wilson_hop_parity(Uh, psi, L, r, bc, k["dagger"], p)) < 1e-13, k
        for o in (x, y, D):
            o.close()
    U.close()
    lat.close()
