"""Solver edge cases: zero right-hand side, exact initial guess, non-finite input, iteration caps that are not a multiple of the
polling burst -- for the CG, BiCGStab (plain and even-odd), multi-shift and mixed-precision solvers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KAPPA = 0.141139


def _setup(lq, L=(4, 4, 4, 8), name="Wilson", eps=1e-19, maxit=3000):
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=9)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": name, "κ": KAPPA, "mass": 0.5, "eps_CG": eps, "MaxCGstep": maxit})
    kind = lq.WILSON if name == "Wilson" else lq.STAGGERED
    b = lq.Fermionfields(U.lattice, kind)
    return U, D, b, kind


@pytest.mark.parametrize("name", ["Wilson", "Staggered"])
def test_zero_right_hand_side_and_exact_guess(lq, name):
    assert lq.lib.device_count() > 0
    U, D, b, kind = _setup(lq, name=name)
    A = lq.DdagD_operator(D)
    x = b.similar()
    lq.gauss_distribution_fermion_(x, 3)             # a non-zero guess must be driven to the solution x = 0
    it, rr = lq.solve_DinvX_(x, A, b, return_info=True)
    assert rr < 1e-19 and np.abs(x.download()).max() < 1e-8
    lq.clear_fermion_(x)
    it, rr = lq.solve_DinvX_(x, A, b, return_info=True)      # b = 0, x = 0: converged before the first iteration
    assert it == 0 and rr == 0.0 and not x.download().any()
    it, outer, rr = lq.solve_mixed_DinvX_(x, A, b, return_info=True)
    assert it == 0 and outer == 0 and rr == 0.0
    xs = [b.similar() for _ in range(2)]
    it, rr = lq.shiftedcg(xs, [0.1, 1.0], x, A, b, return_info=True)
    assert it == 0 and not xs[0].download().any()
    if kind == lq.WILSON:
        for m in ("bicgstab", "bicgstab_evenodd"):
            D.method_CG = m
            lq.clear_fermion_(x)
            it, rr = lq.solve_DinvX_(x, D, b, return_info=True)
            assert it == 0 and rr == 0.0 and not x.download().any()
    # exact initial guess: A x0 = b0  ->  zero iterations
    lq.gauss_distribution_fermion_(x, 4)
    lq.mul_(b, A, x)
    it, rr = lq.solve_DinvX_(x, A, b, return_info=True)
    assert it == 0 and rr < 1e-19


def test_non_finite_input_is_reported(lq):
    U, D, b, kind = _setup(lq)
    h = np.zeros(U.lattice.fermion_shape(kind), dtype=np.complex128)
    h[0, 0, 0, 0, 0, 0] = np.nan
    b.upload(h)
    x = b.similar()
    with pytest.raises(lq.LQCDError):
        lq.solve_DinvX_(x, lq.DdagD_operator(D), b)
    D.method_CG = "bicgstab"
    lq.clear_fermion_(x)
    with pytest.raises(lq.LQCDError):
        lq.solve_DinvX_(x, D, b)
    lq.clear_fermion_(x)
    with pytest.raises(lq.LQCDError):
        lq.solve_mixed_DinvX_(x, lq.DdagD_operator(D), b)


@pytest.mark.parametrize("maxit", [1, 7, 8, 9, 13])
def test_iteration_cap_is_exact(lq, maxit):
    """The host polls every 8 (CG) / 4 (BiCGStab) iterations; a cap in between must stop exactly there."""
    U, D, b, kind = _setup(lq, eps=1e-30, maxit=maxit)
    lq.gauss_distribution_fermion_(b, 5)
    x = b.similar()
    try:
        lq.solve_DinvX_(x, lq.DdagD_operator(D), b)
        raise AssertionError("expected NotConverged")
    except lq.NotConverged as e:
        assert f"maxsteps = {maxit}" in str(e)
    # the iterate after exactly maxit iterations equals the fixed-window solver's
    ref = b.similar()
    lq.lib.check(lq.lib.lib().lqcd_solve_cg_DdagD_fixed(D._h, ref._h, b._h, maxit))
    assert np.array_equal(x.download(), ref.download())
    D.method_CG = "bicgstab"
    lq.clear_fermion_(x)
    with pytest.raises(lq.NotConverged):
        lq.solve_DinvX_(x, D, b)
