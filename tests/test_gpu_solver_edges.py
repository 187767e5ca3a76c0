"""Solver edge cases: zero right-hand side, exact initial guess, non-finite input, iteration caps that are not a multiple of the
polling burst -- for the CG, BiCGStab (plain and even-odd), multi-shift and mixed-precision solvers."""
import numpy as np
from conftest import rel_err
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


@pytest.mark.parametrize("parity", [0, 1])
def test_staggered_parity_block_solve_matches_oracle(lq, orc, parity):
    """D'D of the staggered operator is block diagonal in parity: the half-lattice CG on one block equals the full-lattice oracle CG
    on a source that lives on that parity; the other half of the solution field is left alone."""
    L = (8, 4, 6, 4)
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 941)
    U = lq.Gaugefields(lat).upload(Uh)
    mass, bc = 0.1, (1, 1, 1, -1)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": mass, "boundarycondition": bc, "eps_CG": 1e-20})
    bh = orc.gaussian_spinor(lat.fermion_shape(lq.STAGGERED), 942)
    t, z, y, x = np.meshgrid(range(L[3]), range(L[2]), range(L[1]), range(L[0]), indexing="ij")
    mask = ((x + y + z + t) & 1) == parity                       # host layout [.., t, z, y, x, c] -> see fermion_shape
    bp = bh.copy()
    bp.reshape(-1, L[3], L[2], L[1], L[0], 3)[:, ~mask, :] = 0.0
    b = lq.Fermionfields(lat, lq.STAGGERED).upload(bp)
    marker = orc.gaussian_spinor(lat.fermion_shape(lq.STAGGERED), 943)
    mk = marker.copy()
    mk.reshape(-1, L[3], L[2], L[1], L[0], 3)[:, mask, :] = 0.0  # initial guess: zero on the solved parity, a marker on the other
    sol = lq.Fermionfields(lat, lq.STAGGERED).upload(mk)
    it, rr = lq.solve_parity_DinvX_(sol, lq.DdagD_operator(D), b, parity, return_info=True)
    xo, ito, _, st = orc.cg_DdagD(orc.STAGGERED, Uh, bp, L, mass, 1.0, bc, eps=1e-20)
    assert st == 0 and rr < 1e-20 and abs(it - ito) <= 2
    got = sol.download().reshape(-1, L[3], L[2], L[1], L[0], 3)
    ref = xo.reshape(-1, L[3], L[2], L[1], L[0], 3)
    assert np.abs(got[:, mask, :] - ref[:, mask, :]).max() < 1e-9 * np.abs(ref).max()
    assert np.abs(ref[:, ~mask, :]).max() < 1e-12 * np.abs(ref).max()            # the exact solution has no component there
    assert np.array_equal(got[:, ~mask, :], mk.reshape(-1, L[3], L[2], L[1], L[0], 3)[:, ~mask, :])
    with pytest.raises(lq.LQCDError):
        Dw = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.141139})
        w = lq.Fermionfields(lat, lq.WILSON)
        lq.solve_parity_DinvX_(w.similar(), lq.DdagD_operator(Dw), w, 0)


def test_four_taste_action_takes_the_half_lattice_solve(lq, orc):
    """FermiAction(Nf = 4): eta lives on the even sites; lqcd_fermi_action / lqcd_calc_UdSfdU notice the empty odd half and solve on
    half-lattice vectors -- same S_f and force as with the full-lattice CG (tunable staggered_parity_solve = 0)."""
    L = (8, 4, 6, 4)
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 944)
    U = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": 0.1, "boundarycondition": (1, 1, 1, -1), "eps_CG": 1e-22})
    fa = lq.FermiAction(D, {"Nf": 4})
    xi, phi = lq.Fermionfields(lat, lq.STAGGERED), lq.Fermionfields(lat, lq.STAGGERED)
    lq.gauss_sampling_in_action_(xi, U, fa, 945)
    lq.sample_pseudofermions_(phi, U, fa, xi)
    G = lq.Gaugefields(lat)
    out = {}
    for mode in (1, 0):
        lat.set_param("staggered_parity_solve", mode)
        S, it = lq.evaluate_FermiAction(fa, U, phi, return_info=True)
        lq.calc_UdSfdU_(G, fa, U, phi)
        out[mode] = (S, it, G.download())
    lat.set_param("staggered_parity_solve", 1)
    assert abs(out[1][0] - out[0][0]) < 1e-10 * abs(out[0][0]) and abs(out[1][1] - out[0][1]) <= 2
    assert np.abs(out[1][2] - out[0][2]).max() < 1e-9 * np.abs(out[0][2]).max()


@pytest.mark.parametrize("kind_name,L", [("Staggered", (8, 8, 8, 8)), ("Wilson", (8, 8, 8, 8)), ("Wilson", (4, 4, 4, 4)), ("Staggered", (16, 8, 8, 8))])
def test_small_lattice_cg_without_reduction_launches_gives_identical_iterates(lq, orc, kind_name, L):
    """cg_small (default on small unpartitioned lattices): the reductions of a fused CG iteration run in the prologues of the kernels
    that consume them, in reduce_final's summation order -- the solution must be bit-identical to the 5-launch iteration's, with the
    same iteration count and residual, for to-tolerance solves and for the fixed-length window."""
    kind = lq.WILSON if kind_name == "Wilson" else lq.STAGGERED
    lat = lq.Lattice(L)
    lat.set_param("cg_persist", 0)        # this test is about the launch chain (the one-launch form: tests/test_gpu_cg_persist.py)
    U = lq.Gaugefields(lat).upload(orc.hot_gauge(L, 941))
    D = lq.Dirac_operator(U, None, {"Dirac_operator": kind_name, "κ": 0.141139, "mass": 0.5, "boundarycondition": (1, 1, 1, -1), "eps_CG": 1e-18})
    b = lq.Fermionfields(lat, kind)
    lq.gauss_distribution_fermion_(b, 942)
    A = lq.DdagD_operator(D)
    sols, infos = [], []
    for small in (0, 1):
        lat.set_param("cg_small", small)
        x = b.similar()
        infos.append(lq.solve_DinvX_(x, A, b, return_info=True))
        sols.append(x.download())
    assert infos[0][0] == infos[1][0] and infos[0][1] == infos[1][1] and infos[1][1] < 1e-18
    assert np.array_equal(sols[0], sols[1])
    wins = []
    for small in (0, 1):
        lat.set_param("cg_small", small)
        x = b.similar()
        lq.lib.check(lq.lib.lib().lqcd_solve_cg_DdagD_fixed(D._h, x._h, b._h, 7))
        wins.append(x.download())
    assert np.array_equal(wins[0], wins[1])
    # a converged start: zero iterations either way
    lat.set_param("cg_small", 1)
    x = lq.Fermionfields(lat, kind).upload(sols[1])
    it, rr = lq.solve_DinvX_(x, A, b, return_info=True)
    assert it == 0 and rr < 1e-18


@pytest.mark.parametrize("kind_name", ["Wilson", "Staggered"])
@pytest.mark.parametrize("dagger", [False, True])
def test_bicg_matches_oracle(lq, orc, kind_name, dagger):
    """method_CG = "bicg" (the reference's default for solve_DinvX!(y, D, x)): device BiCG = the oracle's, iteration for iteration,
    and the true residual obeys the absolute stopping rule."""
    L, bc = (4, 4, 4, 8), (1, 1, 1, -1)
    kind, okind, km = (lq.WILSON, orc.WILSON, 0.12) if kind_name == "Wilson" else (lq.STAGGERED, orc.STAGGERED, 0.5)
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 951)
    U = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": kind_name, "κ": km, "mass": km, "boundarycondition": bc, "eps_CG": 1e-18, "method_CG": "bicg"})
    bh = orc.gaussian_spinor(lat.fermion_shape(kind), 952)
    b = lq.Fermionfields(lat, kind).upload(bh)
    x, y = b.similar(), b.similar()
    Dd = D.adjoint() if dagger else D
    it, rr = lq.solve_DinvX_(x, Dd, b, return_info=True)
    xo, ito, rro, st = orc.bicg(okind, Uh, bh, L, km, 1.0, bc, dagger, eps=1e-18)
    assert st == 0 and abs(it - ito) <= 1 and rr < 1e-18 and rel_err(x.download(), xo) < 1e-8
    lq.mul_(y, Dd, x)
    lq.add_fermion_(y, -1.0, b)
    assert lq.dot(y, y).real < 1e-17
    Dd.MaxCGstep = 3
    with pytest.raises(lq.NotConverged):
        lq.solve_DinvX_(x.similar(), Dd, b)


@pytest.mark.parametrize("kind_name", ["Wilson", "Staggered"])
def test_deferred_x_update_gives_identical_iterates(lq, orc, kind_name):
    """cg_defer_x: 1 = x is updated every second iteration with both search directions, p ping-pongs between two buffers; K = 3..8 (default 4, round 5) = a ring
    of K search-direction buffers, x += K terms every K-th iteration.  Same operations in the same order per element: the solution, the iteration count and
    windows of every length modulo the ring (one that ends inside the ring leaves pending updates that must be flushed) are bit-identical to the plain fused iteration."""
    kind = lq.WILSON if kind_name == "Wilson" else lq.STAGGERED
    L = (8, 8, 8, 8)
    lat = lq.Lattice(L)
    lat.set_param("cg_small", 0)
    U = lq.Gaugefields(lat).upload(orc.hot_gauge(L, 961))
    D = lq.Dirac_operator(U, None, {"Dirac_operator": kind_name, "κ": 0.141139, "mass": 0.5, "boundarycondition": (1, 1, 1, -1), "eps_CG": 1e-18})
    b = lq.Fermionfields(lat, kind)
    lq.gauss_distribution_fermion_(b, 962)
    A = lq.DdagD_operator(D)
    res = {}
    for defer in (0, 1, 3, 4, 8):
        lat.set_param("cg_defer_x", defer)
        x = b.similar()
        info = lq.solve_DinvX_(x, A, b, return_info=True)
        wins = []
        for niter in (1, 2, 3, 4, 5, 7, 8, 9, 16, 17):
            xw = b.similar()
            lq.lib.check(lq.lib.lib().lqcd_solve_cg_DdagD_fixed(D._h, xw._h, b._h, niter))
            wins.append(xw.download())
        A.MaxCGstep = 7                                     # exhausted solve on an odd count: x still holds the 7-iteration iterate
        xe = b.similar()
        with pytest.raises(lq.NotConverged):
            lq.solve_DinvX_(xe, A, b)
        A.MaxCGstep = 3000
        res[defer] = (info, x.download(), wins, xe.download())
    for defer in (1, 3, 4, 8):
        assert res[0][0] == res[defer][0] and res[defer][0][1] < 1e-18, defer
        assert np.array_equal(res[0][1], res[defer][1]) and np.array_equal(res[0][3], res[defer][3]), defer
        for a, c in zip(res[0][2], res[defer][2]):
            assert np.array_equal(a, c), defer


@pytest.mark.parametrize("L,dagger,csw", [((16, 8, 8, 8), False, 0.0), ((16, 16, 16, 32), True, 0.0), ((8, 8, 8, 16), True, 0.0), ((8, 8, 8, 16), False, 1.3), ((16, 16, 16, 32), True, 1.0)])
def test_evenodd_bicgstab_merged_update_on_the_recurrences(lq, orc, L, dagger, csw):
    """bicg_fused = 4 (opt-in): x / r and p update as ONE launch without a barrier -- rho' = rho - alpha <r0, v> - omega <r0, t> and |r'|^2 = |s|^2 - |<t,s>|^2 / |t|^2
    from inner products that exist before r' does (<r0, t> from a second inner product in the dot epilogue: scalar-addressing kernel, plain direction-split kernel on small
    planes, clover-on-hop kernel of the Wilson-clover Schur operator).  Equal to the two-launch
    chain up to the rounding of the two recurrences: same iteration count, solution to 1e-10, stopping rule on the recursive residual, true residual recomputed here.
    bicg_rec_guard = 0 distrusts the recurrence for |r'|^2 always: the stopping test then waits for the summed |r'|^2 (the next iteration's first streaming kernel)."""
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "WilsonClover" if csw else "Wilson", "Clover_coefficient": csw, "κ": KAPPA, "eps_CG": 1e-19, "MaxCGstep": 3000})
    Dd = D.adjoint() if dagger else D
    Dd.method_CG = "bicgstab_evenodd"
    b = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(b, 112)
    out = {}
    for mode, guard in ((2, 6), (4, 6), (4, 0)):
        lat.set_param("bicg_fused", mode)
        lat.set_param("bicg_rec_guard", guard)
        x = b.similar()
        it, rr = lq.solve_DinvX_(x, Dd, b, return_info=True)
        assert rr < 1e-19 and lat.get_param("bicg_xrp_active") == (2 if mode == 4 else 0), (mode, guard)
        r = b.similar()
        lq.mul_(r, Dd, x)
        lq.add_fermion_(r, -1.0, b)
        assert lq.dot(r, r).real < (1e-17 if csw else 1e-18), (mode, guard)           # the true residual of the full system (the even-odd solve stops on the [A^-1-preconditioned] Schur system's recursive one)
        out[(mode, guard)] = (x.download(), it)
    # the layout of the hops' dot partials ([workgroup][value] or [value][workgroup], tunable bicg_dot_soa) changes addresses, not additions: the same bits
    lat.set_param("bicg_fused", 4)
    lat.set_param("bicg_rec_guard", 6)
    for soa in (0, 2):
        lat.set_param("bicg_dot_soa", soa)
        x = b.similar()
        it, rr = lq.solve_DinvX_(x, Dd, b, return_info=True)
        assert it == out[(4, 6)][1] and np.array_equal(x.download(), out[(4, 6)][0]), soa
    lat.set_param("bicg_dot_soa", 1)
    lat.set_param("bicg_fused", 2)
    lat.set_param("bicg_rec_guard", 6)
    for key in ((4, 6), (4, 0)):
        assert abs(out[key][1] - out[(2, 6)][1]) <= 1 and rel_err(out[key][0], out[(2, 6)][0]) < 1e-10, key
    if L[3] == 8 and not csw:
        xo, ito, rro, st = orc.wilson_bicgstab_eo(U.download(), b.download(), L, KAPPA, 1.0, (1, 1, 1, -1), dagger, eps=1e-19)
        assert st == 0 and abs(ito - out[(4, 6)][1]) <= 1 and rel_err(out[(4, 6)][0], xo) < 1e-9


def test_evenodd_bicgstab_fused_chain_is_bit_identical_to_the_unfolded_one(lq, orc):
    """Tunable bicg_fused.  1: the inner products of an iteration come from the epilogue of the Schur operator's second hop, reductions and scalar steps
    are separate one-block launches; 2 (default): on lattices of <= 1024 chunks per parity they run in the consumers' prologues -- 7 dependent launches
    per iteration instead of 17; 3 (opt-in, round 6): the x / r and the p update as one launch with a grid barrier -- 6 (measured slower).  Same partials, same summation order, same scalar
    expressions: the same BITS in x, the same iteration count.  0 is
    the generic chain (separate dot-product kernels, other partial sums): equal to rounding.  All against the oracle's solution."""
    for L, dagger, csw in (((8, 8, 8, 16), False, 0.0), ((16, 16, 16, 32), True, 0.0), ((4, 4, 4, 8), False, 0.0), ((8, 8, 8, 16), True, 1.3), ((16, 16, 16, 32), False, 1.0)):
        U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
        lat = U.lattice
        # csw != 0: Wilson-clover, the inverse clover blocks applied to the hop sums inside the Schur operator's two launches (forms 1, 2) or by separate passes (form 0)
        D = lq.Dirac_operator(U, None, {"Dirac_operator": "WilsonClover" if csw else "Wilson", "Clover_coefficient": csw, "κ": KAPPA, "eps_CG": 1e-19, "MaxCGstep": 3000})
        Dd = D.adjoint() if dagger else D
        Dd.method_CG = "bicgstab_evenodd"
        b = lq.Fermionfields(lat, lq.WILSON)
        lq.gauss_distribution_fermion_(b, 112)
        out = {}
        for mode in (3, 2, 1, 0):
            lat.set_param("bicg_fused", mode)
            x = b.similar()
            it, rr = lq.solve_DinvX_(x, Dd, b, return_info=True)
            out[mode] = (x.download(), it, rr)
            assert rr < 1e-19
            assert lat.get_param("bicg_xrp_active") == (1 if mode == 3 else 0), mode      # the fused launch did run (every workgroup of it resident) / did not
        lat.set_param("bicg_fused", 2)
        assert out[2][1] == out[1][1] and out[2][2] == out[1][2] and np.array_equal(out[2][0], out[1][0]), L      # folded == unfolded, bit for bit
        assert out[3][1] == out[2][1] and out[3][2] == out[2][2] and np.array_equal(out[3][0], out[2][0]), L      # ... == the x / r / p update as one launch with a grid barrier (round 6)
        assert abs(out[0][1] - out[2][1]) <= 1 and rel_err(out[0][0], out[2][0]) < 1e-10
        if L[0] <= 8:
            Uh = U.download()
            if csw:
                xo, ito, rro, st = orc.wilson_clover_bicgstab_eo(Uh, orc.clover_build(Uh, L, KAPPA, csw), b.download(), L, KAPPA, 1.0, (1, 1, 1, -1), dagger, eps=1e-19)
            else:
                xo, ito, rro, st = orc.wilson_bicgstab_eo(Uh, b.download(), L, KAPPA, 1.0, (1, 1, 1, -1), dagger, eps=1e-19)
            assert st == 0 and abs(ito - out[2][1]) <= 1 and rel_err(out[2][0], xo) < 1e-9
    # a solve that converges in the FIRST half step and one whose right-hand side is zero
    L = (4, 4, 4, 4)
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="cold")
    lat = U.lattice
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.01, "eps_CG": 1e-6, "MaxCGstep": 50})
    D.method_CG = "bicgstab_evenodd"
    b = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(b, 7)
    x = b.similar()
    it, rr = lq.solve_DinvX_(x, D, b, return_info=True)
    assert it <= 2 and rr < 1e-6
    lq.clear_fermion_(b)
    lq.clear_fermion_(x)
    it, rr = lq.solve_DinvX_(x, D, b, return_info=True)
    assert it == 0 and rr == 0.0 and np.abs(x.download()).max() == 0.0
    D.MaxCGstep = 1
    D.eps_CG = 1e-30
    lq.gauss_distribution_fermion_(b, 8)
    with pytest.raises(lq.NotConverged):
        lq.solve_DinvX_(x, D, b)
