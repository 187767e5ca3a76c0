"""GPU: the HIP path against operator-level vectors produced by the reference itself (tests/ref_vectors.py), when they exist:
mul!(y,D,x), mul!(y,D',x) and solve_DinvX!(y, DdagD, x) of LatticeDiracOperators.jl on the reference's own 4^4 fixtures with a
closed-form source.  Tolerances: operator 1e-12 relative (two independent fp64 summation orders), solution 1e-8, and the device's own
true residual below 1e-17.  Without the files the test reports "reference vectors absent" (skip with that reason)."""
import os

import numpy as np
import pytest

import ref_vectors as rv
from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["wilson", "staggered"])
def test_hip_path_matches_reference_vectors(lq, kind):
    assert lq.lib.device_count() > 0, "no HIP device visible: the product has no CPU fallback"
    if not rv.available():
        pytest.skip(rv.ABSENT)
    lat = lq.Lattice(rv.L)
    U = lq.Gaugefields(lat).upload(lq.gauge_io.load_ildg(os.path.join(rv.GOLDEN, rv.FIXTURE[kind]), rv.L))
    name = "Wilson" if kind == "wilson" else "Staggered"
    D = lq.Dirac_operator(U, None, {"Dirac_operator": name, "κ": rv.KAPPA, "mass": rv.MASS, "r": 1.0, "boundarycondition": rv.BC,
                                    "eps_CG": 1e-19, "MaxCGstep": 3000})
    k = lq.WILSON if kind == "wilson" else lq.STAGGERED
    x = lq.Fermionfields(lat, k).upload(rv.closed_form_source(kind))
    y = x.similar()
    for op, which in ((D, "D"), (D.adjoint(), "Ddag")):
        lq.mul_(y, op, x)
        assert rel_err(y.download(), rv.load(kind, which)) < 1e-12, (kind, which)
    sol = x.similar()
    lq.solve_DinvX_(sol, lq.DdagD_operator(D), x)
    assert rel_err(sol.download(), rv.load(kind, "cg_x")) < 1e-8
    r = x.similar()
    lq.mul_(r, lq.DdagD_operator(D), sol)
    lq.add_fermion_(r, -1.0, x)
    assert lq.dot(r, r).real < 1e-17
    assert abs(lq.calculate_Plaquette(U) - rv.meta()["%s_plaquette" % kind]) < 1e-12
