"""12-real link compression of the split Dslash kernels (tunable gauge_recon = 12, the default): rows 0 and 1 are read, row 2 is rebuilt as
conj(row0 x row1).  Used only when every link of the current field is unitary to 1e-14, so the result stays within the
fp64 Dslash tolerance (1e-13) of the oracle; anything else silently reads the 18-real field."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu

KAPPA = 0.141139
BC = (1, 1, 1, -1)


@pytest.mark.parametrize("L", [(8, 4, 6, 4), (16, 8, 4, 4)])
def test_recon12_dslash_and_cg_match_oracle(lq, orc, L):
    assert lq.lib.device_count() > 0
    lat = lq.Lattice(L)
    lat.set_param("gauge_recon", 12)
    Uh = orc.hot_gauge(L, 31)
    U = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "boundarycondition": BC, "eps_CG": 1e-19})
    psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 32)
    x = lq.Fermionfields(lat, lq.WILSON).upload(psi)
    y = x.similar()
    for dag in (False, True):
        lq.mul_(y, D.adjoint() if dag else D, x)
        assert lat.get_param("recon_active") == 1
        assert rel_err(y.download(), orc.wilson_D(Uh, psi, L, KAPPA, 1.0, BC, dag)) < 1e-13
    for out_sub, in_sub, p in ((lq.EVEN, lq.ODD, 0), (lq.ODD, lq.EVEN, 1)):
        xin = lq.Fermionfields(lat, lq.WILSON, in_sub).upload(psi)
        yout = lq.Fermionfields(lat, lq.WILSON, out_sub)
        lq.hop_(yout, D, xin)
        assert rel_err(yout.download(), orc.wilson_hop_parity(Uh, psi, L, 1.0, BC, False, p)) < 1e-13
    sol = x.similar()
    it, rr = lq.solve_DinvX_(sol, lq.DdagD_operator(D), x, return_info=True)
    xo, ito, rro, st = orc.cg_DdagD(orc.WILSON, Uh, psi, L, KAPPA, 1.0, BC, eps=1e-19)
    assert st == 0 and abs(it - ito) <= 1 and rel_err(sol.download(), xo) < 1e-9
    # the compressed copy follows the field: new links in the same handle
    Uh2 = orc.hot_gauge(L, 33)
    U.upload(Uh2)
    lq.mul_(y, D, x)
    assert lat.get_param("recon_active") == 1
    assert rel_err(y.download(), orc.wilson_D(Uh2, psi, L, KAPPA, 1.0, BC)) < 1e-13
    # links that are not unitary to 1e-14 are not compressed -- and the result is still that of the 18 stored reals
    Uh3 = Uh2.copy()
    Uh3[2, 1, 0, 1, 2, 0, 0] += 3e-10
    U.upload(Uh3)
    lq.mul_(y, D, x)
    assert lat.get_param("recon_active") == 0
    assert rel_err(y.download(), orc.wilson_D(Uh3, psi, L, KAPPA, 1.0, BC)) < 1e-13
    # a NaN in a link is "not unitary" as well (the comparison must not let it pass)
    Uh4 = Uh2.copy()
    Uh4[1, 0, 1, 0, 1, 2, 2] = np.nan
    U.upload(Uh4)
    lq.mul_(y, D, x)
    assert lat.get_param("recon_active") == 0
    # general r has no split kernel -> no compression
    D2 = lq.Dirac_operator(U.upload(Uh2), None, {"Dirac_operator": "Wilson", "κ": KAPPA, "r": 0.7, "boundarycondition": BC})
    lq.mul_(y, D2, x)
    assert lat.get_param("recon_active") == 0
    assert rel_err(y.download(), orc.wilson_D(Uh2, psi, L, KAPPA, 0.7, BC)) < 1e-13


def test_recon12_on_reference_fixture_stays_exact(lq, orc):
    L = (4, 4, 4, 4)
    Uh = lq.gauge_io.load_ildg(os.path.join(GOLDEN, "wilson_4x4x4x4.ildg"), L)
    lat = lq.Lattice(L)
    lat.set_param("gauge_recon", 12)
    U = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "boundarycondition": BC})
    psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 34)
    x = lq.Fermionfields(lat, lq.WILSON).upload(psi)
    y = x.similar()
    lq.mul_(y, D, x)
    assert rel_err(y.download(), orc.wilson_D(Uh, psi, L, KAPPA, 1.0, BC)) < 1e-13
    print("fixture: recon_active =", lat.get_param("recon_active"), "unitarity dev", orc.unitarity_dev(Uh, L))


def test_recon12_staggered_matches_oracle(lq, orc):
    L = (8, 4, 6, 4)
    lat = lq.Lattice(L)
    lat.set_param("gauge_recon", 12)
    Uh = orc.hot_gauge(L, 35)
    U = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": 0.5, "boundarycondition": BC, "eps_CG": 1e-19})
    psi = orc.gaussian_spinor(lat.fermion_shape(lq.STAGGERED), 36)
    x = lq.Fermionfields(lat, lq.STAGGERED).upload(psi)
    y = x.similar()
    for dag in (False, True):
        lq.mul_(y, D.adjoint() if dag else D, x)
        assert lat.get_param("recon_active") == 1
        assert rel_err(y.download(), orc.staggered_D(Uh, psi, L, 0.5, BC, dag)) < 1e-13
    sol = x.similar()
    it, rr = lq.solve_DinvX_(sol, lq.DdagD_operator(D), x, return_info=True)
    xo, ito, rro, st = orc.cg_DdagD(orc.STAGGERED, Uh, psi, L, 0.5, 1.0, BC, eps=1e-19)
    assert st == 0 and abs(it - ito) <= 1 and rel_err(sol.download(), xo) < 1e-9
