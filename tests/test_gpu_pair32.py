"""GPU parity of the fp32 site-pair Wilson kernel (stencil_pair32.hip: two sites per lane, n and n + T/2, packed fp32 arithmetic) -- the
inner operator of the mixed-precision solvers for plain Wilson r = 1 on unpartitioned lattices (tunable mixed_pair32).
Reached through lqcd_op_apply_f32 (the fp32 operator a mixed solve would use) and through the mixed-precision solvers themselves.
Tolerance of one fp32 application: 2e-6 relative to the oracle's fp64 result (fp32 fields, fp32 accumulation over 8 hops)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
KAPPA = 0.141139


@pytest.fixture(scope="module")
def gpu(lq):
    assert lq.lib.device_count() > 0, "no HIP device visible: the product has no CPU fallback"
    return lq


# z-planes of whole chunks, t-slices that split over 8 XCDs, T a multiple of 4; XH = 8 / 16 / 4 / 6 (magic division by XH per lane)
CASES = [(16, 8, 8, 4), (32, 4, 8, 8), (8, 16, 8, 4), (16, 16, 16, 8), (12, 32, 8, 4)]


@pytest.mark.parametrize("L", CASES)
@pytest.mark.parametrize("bc", [(1, 1, 1, -1), (-1, -1, 1, 1), (1, -1, -1, -1)])
def test_pair32_dslash_matches_oracle_at_fp32_accuracy(gpu, orc, L, bc):
    lq = gpu
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 41)
    U = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "boundarycondition": bc})
    psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 42)
    x = lq.Fermionfields(lat, lq.WILSON).upload(psi)
    y2, y1 = x.similar(), x.similar()
    for dagger in (False, True):
        ref = orc.wilson_D(Uh, psi, L, KAPPA, 1.0, bc, dagger)
        lat.set_param("mixed_pair32", 1)
        lq.mul_f32_(y2, D.adjoint() if dagger else D, x)
        assert lat.get_param("pair32_active") == 1
        lat.set_param("mixed_pair32", 0)
        lq.mul_f32_(y1, D.adjoint() if dagger else D, x)
        assert lat.get_param("pair32_active") == 0
        e2, e1 = rel_err(y2.download(), ref), rel_err(y1.download(), ref)
        assert e2 < 2e-6 and e1 < 2e-6, (L, bc, dagger, e2, e1)
        # two fp32 kernels with different summation orders: they agree with each other as well as with the oracle
        assert rel_err(y2.download(), y1.download()) < 2e-6


def test_pair32_layout_round_trip_is_exact_for_fp32_values(gpu, orc):
    """fp64 -> pair layout -> fp64 through kappa = 0 (D = 1): values that are exactly representable in fp32 come back bit for bit"""
    lq = gpu
    L = (16, 8, 8, 8)
    lat = lq.Lattice(L)
    U = lq.Gaugefields(lat).upload(orc.hot_gauge(L, 43))
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.0})
    psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 44).astype(np.complex64).astype(np.complex128)
    x = lq.Fermionfields(lat, lq.WILSON).upload(psi)
    y = x.similar()
    lat.set_param("mixed_pair32", 1)
    lq.mul_f32_(y, D, x)
    assert lat.get_param("pair32_active") == 1
    assert np.array_equal(y.download(), psi)


@pytest.mark.parametrize("L", [(16, 8, 8, 4), (12, 32, 8, 4)])
def test_mixed_solvers_on_the_pair_kernel(gpu, orc, L):
    lq = gpu
    bc = (1, 1, 1, -1)
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 45)
    U = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "boundarycondition": bc, "eps_CG": 1e-18, "MaxCGstep": 3000})
    psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 46)
    b = lq.Fermionfields(lat, lq.WILSON).upload(psi)
    xo, ito, rro, st = orc.cg_DdagD(orc.WILSON, Uh, psi, L, KAPPA, 1.0, bc, eps=1e-18)
    assert st == 0
    A = lq.DdagD_operator(D)
    for pair in (1, 0):
        lat.set_param("mixed_pair32", pair)
        sol = b.similar()
        it, outer, rr = lq.solve_mixed_DinvX_(sol, A, b, return_info=True)
        assert lat.get_param("pair32_active") == pair
        assert rr < 1e-18 and rel_err(sol.download(), xo) < 1e-8, (pair, it, outer, rr)
        r = b.similar()
        lq.mul_(r, A, sol)
        lq.add_fermion_(r, -1.0, b)
        assert lq.dot(r, r).real < 1e-18
    # mixed-precision multi-shift CG through the same fp32 operator
    lat.set_param("mixed_pair32", 1)
    sig = [0.01, 0.1, 1.0]
    xs = [b.similar() for _ in sig]
    lq.shiftedcg_mixed(xs, sig, None, A, b, eps=1e-16)
    for s, xj in zip(sig, xs):
        r = b.similar()
        lq.mul_(r, A, xj)
        lq.add_fermion_(r, s, xj)
        lq.add_fermion_(r, -1.0, b)
        assert lq.dot(r, r).real < 1e-16, s


def test_pair32_is_not_used_where_it_does_not_apply(gpu, orc):
    """clover term, general r, T not a multiple of 4, half-chunk z-planes: the one-site-per-lane fp32 kernels run (same contract)"""
    lq = gpu
    for L, extra in (((8, 8, 8, 8), {}), ((16, 8, 8, 6), {}), ((16, 8, 8, 4), {"r": 0.8}), ((16, 8, 8, 4), {"Dirac_operator": "WilsonClover", "Clover_coefficient": 1.2})):
        lat = lq.Lattice(L)
        Uh = orc.hot_gauge(L, 47)
        U = lq.Gaugefields(lat).upload(Uh)
        par = {"Dirac_operator": "Wilson", "κ": 0.12, "eps_CG": 1e-16, "MaxCGstep": 3000}
        par.update(extra)
        D = lq.Dirac_operator(U, None, par)
        b = lq.Fermionfields(lat, lq.WILSON)
        lq.gauss_distribution_fermion_(b, 48)
        sol = b.similar()
        it, outer, rr = lq.solve_mixed_DinvX_(sol, lq.DdagD_operator(D), b, return_info=True)
        assert lat.get_param("pair32_active") == 0, (L, extra)
        r = b.similar()
        lq.mul_(r, lq.DdagD_operator(D), sol)
        lq.add_fermion_(r, -1.0, b)
        assert rr < 1e-16 and lq.dot(r, r).real < 1e-16, (L, extra)


@pytest.mark.parametrize("L", [(16, 8, 8, 4), (16, 16, 16, 8)])
def test_x_update_in_the_epilogue_of_the_update_mode_kernel_gives_identical_solutions(gpu, orc, L):
    """mixed_xfuse = 1 (opt-in; measured slower than the separate update, profiles/r03_mixed_precision.log): the fp32 solver's x += alpha p rides in the epilogue of the update-mode D^+ launch of the site-pair kernel and the
    update kernel forms p only -- same operations per element: iteration counts and solutions are bit-identical to the separate x/p update."""
    lq = gpu
    lat = lq.Lattice(L)
    U = lq.Gaugefields(lat).upload(orc.hot_gauge(L, 47))
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "boundarycondition": (1, 1, 1, -1), "eps_CG": 1e-18})
    b = lq.Fermionfields(lat, lq.WILSON).upload(orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 48))
    A = lq.DdagD_operator(D)
    out = []
    for fuse in (1, 0):
        lat.set_param("mixed_xfuse", fuse)
        x = b.similar()
        info = lq.solve_mixed_DinvX_(x, A, b, return_info=True)
        assert lat.get_param("pair32_active") == 1
        out.append((info, x.download()))
    lat.set_param("mixed_xfuse", 0)
    assert out[0][0][:2] == out[1][0][:2] and out[0][0][2] < 1e-18
    assert np.array_equal(out[0][1], out[1][1])
