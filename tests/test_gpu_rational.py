"""RHMC building block on the device: (D'D)^(-alpha) phi through one multi-shift solve with the partial fractions of rational.py."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("alpha,times", [(0.5, 2), (0.25, 4)])
def test_inverse_power_composes_to_the_inverse(lq, orc, alpha, times):
    assert lq.lib.device_count() > 0
    L, mass = (8, 4, 6, 4), 0.5
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 701)
    U = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": mass, "eps_CG": 1e-20})
    A = lq.DdagD_operator(D)
    phi_h = orc.gaussian_spinor(lat.fermion_shape(lq.STAGGERED), 702)
    phi = lq.Fermionfields(lat, lq.STAGGERED).upload(phi_h)
    a, b = phi.similar(), phi.similar()
    lq.substitute_fermion_(a, phi)
    for _ in range(times):                       # (x^-alpha)^times = x^-1
        lq.apply_inverse_power_(b, A, a, alpha, mass ** 2, mass ** 2 + 16.0, tol=1e-11)
        a, b = b, a
    xo, ito, rro, st = orc.cg_DdagD(orc.STAGGERED, Uh, phi_h, L, mass, 1.0, (1, 1, 1, -1), eps=1e-24)
    assert st == 0 and rel_err(a.download(), xo) < 1e-8
