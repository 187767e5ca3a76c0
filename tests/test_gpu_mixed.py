"""GPU parity of the mixed-precision CG (fp32 inner solver through the fp32 build of the stencil, fp64 defect correction).
The contract is the fp64 one: the returned x satisfies |b - D^+D x|^2 < eps with the residual recomputed independently in
fp64, and x equals the oracle's fp64 CG solution to the solver tolerance (1e-9 relative)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

KAPPA, MASS = 0.141139, 0.5
BC = (1, 1, 1, -1)


@pytest.fixture(scope="module")
def gpu(lq):
    assert lq.lib.device_count() > 0, "no HIP device visible: the product has no CPU fallback"
    return lq


def _setup(lq, orc, L, kind_name, seed, bc=BC, eps=1e-19):
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, seed)
    Ud = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(Ud, None, {"Dirac_operator": kind_name, "κ": KAPPA, "mass": MASS, "boundarycondition": bc, "eps_CG": eps})
    return lat, Uh, Ud, D


@pytest.mark.parametrize("kind_name", ["Wilson", "Staggered"])
@pytest.mark.parametrize("L,bc", [((4, 4, 4, 8), BC), ((8, 4, 6, 4), (1, -1, 1, 1))])
def test_mixed_cg_matches_oracle_fp64_solution(gpu, orc, kind_name, L, bc):
    lq = gpu
    kind = lq.WILSON if kind_name == "Wilson" else lq.STAGGERED
    okind = orc.WILSON if kind == lq.WILSON else orc.STAGGERED
    km = KAPPA if kind == lq.WILSON else MASS
    lat, Uh, Ud, D = _setup(lq, orc, L, kind_name, 101, bc)
    b_h = orc.gaussian_spinor(lat.fermion_shape(kind), 102)
    b = lq.Fermionfields(lat, kind).upload(b_h)
    x = b.similar()
    A = lq.DdagD_operator(D)
    it, outer, rr = lq.solve_mixed_DinvX_(x, A, b, return_info=True)
    assert rr < 1e-19 and outer >= 2          # more than one correction step is needed to get from fp32 to 1e-19
    xo, ito, rro, st = orc.cg_DdagD(okind, Uh, b_h, L, km, 1.0, bc, eps=1e-19)
    assert st == 0 and rel_err(x.download(), xo) < 1e-9
    assert it < 3 * ito + 20                  # restarts cost iterations, but not many
    # true residual recomputed with the fp64 operator, independently of the solver
    r = b.similar()
    lq.mul_(r, A, x)
    lq.add_fermion_(r, -1.0, b)
    assert lq.dot(r, r).real < 1e-19
    # a good initial guess is used, not overwritten: restart from the solution -> zero outer steps
    it2, outer2, rr2 = lq.solve_mixed_DinvX_(x, A, b, return_info=True)
    assert outer2 == 0 and it2 == 0 and rr2 < 1e-19


def test_mixed_cg_tight_and_loose_targets(gpu, orc):
    lq = gpu
    L = (8, 8, 8, 8)
    for eps in (1e-8, 1e-24):
        lat, Uh, Ud, D = _setup(lq, orc, L, "Wilson", 103, eps=eps)
        b = lq.Fermionfields(lat, lq.WILSON)
        lq.gauss_distribution_fermion_(b, 104)
        x = b.similar()
        A = lq.DdagD_operator(D)
        it, outer, rr = lq.solve_mixed_DinvX_(x, A, b, return_info=True)
        r = b.similar()
        lq.mul_(r, A, x)
        lq.add_fermion_(r, -1.0, b)
        assert lq.dot(r, r).real < eps and rr < eps
    # non-convergence raises like the fp64 solver
    D.MaxCGstep = 3
    A = lq.DdagD_operator(D)
    lq.clear_fermion_(x)
    with pytest.raises(lq.NotConverged):
        lq.solve_mixed_DinvX_(x, A, b)
    with pytest.raises(lq.LQCDError):
        lq.solve_mixed_DinvX_(x, D, b)


@pytest.mark.parametrize("variant", [0, 1, 3, 5])
def test_mixed_cg_under_every_stencil_variant_setting(gpu, orc, variant):
    """The fp32 build of the stencil keeps Wilson spinors and 12-real links as 16-byte component pairs and only has the
    site-per-lane (0) and direction-split (1) kernels; any other dslash_variant is pinned to 1 for the duration of the solve
    and restored afterwards.  Results agree with the oracle's fp64 solution under all of them."""
    lq = gpu
    L = (8, 8, 4, 8)
    lat, Uh, Ud, D = _setup(lq, orc, L, "Wilson", 141)
    lat.set_param("dslash_variant", variant)
    try:
        b_h = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 142)
        b = lq.Fermionfields(lat, lq.WILSON).upload(b_h)
        x = b.similar()
        it, outer, rr = lq.solve_mixed_DinvX_(x, lq.DdagD_operator(D), b, return_info=True)
        assert lat.get_param("dslash_variant") == variant
        xo, ito, rro, st = orc.cg_DdagD(orc.WILSON, Uh, b_h, L, KAPPA, 1.0, BC, eps=1e-19)
        assert st == 0 and rr < 1e-19 and rel_err(x.download(), xo) < 1e-9
    finally:
        lat.set_param("dslash_variant", 1)


@pytest.mark.parametrize("kind_name", ["Wilson", "Staggered"])
@pytest.mark.parametrize("with_base", [True, False])
def test_mixed_multishift_matches_oracle_and_true_residuals(gpu, orc, kind_name, with_base):
    """Mixed-precision shiftedcg: one fp32 multi-shift pass, then fp64 defect correction per shift.  Contract of the fp64 solver:
    every shifted solution equals the oracle's fp64 multi-shift CG to 1e-9 and |b - (D^+D + sigma_j) x_j|^2 < eps, the residual
    recomputed here with the fp64 operator, independently of the solver."""
    lq = gpu
    L = (4, 4, 4, 8)
    kind = lq.WILSON if kind_name == "Wilson" else lq.STAGGERED
    okind = orc.WILSON if kind == lq.WILSON else orc.STAGGERED
    km = KAPPA if kind == lq.WILSON else MASS
    lat, Uh, Ud, D = _setup(lq, orc, L, kind_name, 151)
    b_h = orc.gaussian_spinor(lat.fermion_shape(kind), 152)
    b = lq.Fermionfields(lat, kind).upload(b_h)
    sig = [0.0, 0.004, 0.06, 0.9, 4.0]
    xs = [b.similar() for _ in sig]
    for x in xs:
        lq.gauss_distribution_fermion_(x, 153)          # zero initial guesses are the solver's business: garbage in the outputs is ignored
    x0 = b.similar() if with_base else None
    A = lq.DdagD_operator(D)
    it, outer, worst = lq.shiftedcg_mixed(xs, sig, x0, A, b, eps=1e-20, return_info=True)
    assert worst < 1e-20 and outer >= len(sig)           # fp32 cannot reach 1e-20: every system needs at least one correction
    o0, oxs, oit, oresid, st = orc.multishift_cg(okind, Uh, b_h, L, km, sig, eps=1e-20)
    assert st == 0
    if with_base:
        assert rel_err(x0.download(), o0) < 1e-9
    r = b.similar()
    for x, ox, s_ in zip(xs, oxs, sig):
        assert rel_err(x.download(), ox) < 1e-9
        lq.mul_(r, A, x)
        lq.add_fermion_(r, s_, x, -1.0, b)
        assert lq.dot(r, r).real < 1e-20
    # a loose target is met by the fp32 pass alone (plus the verification of the true residuals): no correction solves
    loose = 1e-9 * lq.dot(b, b).real                       # relative residual 3e-5: within reach of the fp32 recurrence
    it2, outer2, worst2 = lq.shiftedcg_mixed(xs, sig, x0, A, b, eps=loose, return_info=True)
    assert outer2 == 0 and worst2 < loose and it2 < it
    # error paths of the fp64 solver
    with pytest.raises(lq.NotConverged):
        lq.shiftedcg_mixed(xs, sig, x0, A, b, eps=1e-30, maxsteps=3)
    with pytest.raises(lq.LQCDError):
        lq.shiftedcg_mixed(xs[:1], [-1.0], x0, A, b)
    # zero right-hand side: zero solutions, no iterations
    z = b.similar()
    lq.clear_fermion_(z)
    it3, outer3, worst3 = lq.shiftedcg_mixed(xs[:2], sig[:2], None, A, z, eps=1e-20, return_info=True)
    assert it3 == 0 and outer3 == 0 and worst3 == 0.0 and np.abs(xs[0].download()).max() == 0.0


def test_mixed_multishift_rhmc_poles_and_zeta_underflow(gpu, orc):
    """The shifts of a real rational approximation (x^(-1/4) on the staggered spectrum: poles over five decades) and shifts large
    enough to underflow zeta: frozen shifts must neither stall the fp32 pass nor poison the fields."""
    lq = gpu
    L = (4, 4, 4, 4)
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 161)
    Ud = lq.Gaugefields(lat).upload(Uh)
    m = 0.05
    D = lq.Dirac_operator(Ud, None, {"Dirac_operator": "Staggered", "mass": m, "boundarycondition": BC, "eps_CG": 1e-18})
    b_h = orc.gaussian_spinor(lat.fermion_shape(lq.STAGGERED), 162)
    b = lq.Fermionfields(lat, lq.STAGGERED).upload(b_h)
    a0, res, poles, _ = lq.rational.inverse_power_partial_fractions(0.25, m * m, m * m + 16.0, 1e-9)
    sig = [float(p_) for p_ in poles] + [3000.0]
    xs = [b.similar() for _ in sig]
    A = lq.DdagD_operator(D)
    it, outer, worst = lq.shiftedcg_mixed(xs, sig, None, A, b, eps=1e-18, return_info=True)
    _, oxs, oit, _, st = orc.multishift_cg(orc.STAGGERED, Uh, b_h, L, m, sig, bc=BC, eps=1e-18)
    assert st == 0 and worst < 1e-18
    for x, ox in zip(xs, oxs):
        xh = x.download()
        assert np.isfinite(xh).all() and rel_err(xh, ox) < 1e-8


def test_mixed_cg_rccl_self_partition(gpu, orc):
    """fp32 halos (pack / ncclFloat send-recv / exterior from the fp32 stencil build) on the real RCCL path, one GPU."""
    code = textwrap.dedent("""
        import os, sys, numpy as np
        sys.path.insert(0, os.getcwd())
        import latticeqcd_jl_amd as lq
        from oracle import oracle as orc
        L, K, BC = (8, 4, 6, 8), 0.141139, (1, 1, 1, -1)
        lat = lq.Lattice(L)
        lat.comm_init(lq.comm_unique_id())
        U = orc.hot_gauge(L, 111)
        Ud = lq.Gaugefields(lat).upload(U)
        for name, kind, km in (("Wilson", lq.WILSON, K), ("Staggered", lq.STAGGERED, 0.5)):
            D = lq.Dirac_operator(Ud, None, {"Dirac_operator": name, "κ": K, "mass": 0.5, "boundarycondition": BC, "eps_CG": 1e-19})
            psi = orc.gaussian_spinor(lat.fermion_shape(kind), 112)
            b = lq.Fermionfields(lat, kind).upload(psi)
            x = b.similar()
            it, outer, rr = lq.solve_mixed_DinvX_(x, lq.DdagD_operator(D), b, return_info=True)
            xo, ito, rro, st = orc.cg_DdagD(kind, U, psi, L, km, 1.0, BC, eps=1e-19)
            assert st == 0 and rr < 1e-19 and np.abs(x.download() - xo).max() / np.abs(xo).max() < 1e-9, (name, it, outer, rr)
        print("MIXED_SELF_OK")
    """)
    for mask in ("8", "14"):
        env = dict(os.environ, LQCD_FORCE_PARTITION=mask, HSA_ENABLE_IPC_MODE_LEGACY="0", LQCD_HALO_STREAM_MODE=("-1" if mask == "8" else "3"))      # t only: the tuner's choice; the others: the folded one-stream schedule
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and "MIXED_SELF_OK" in r.stdout, (mask, r.stdout[-2000:], r.stderr[-3000:])


def test_mixed_cg_full_size_is_faster_than_fp64(gpu):
    """32^3x64 Wilson (BASELINE size): same true residual target, fewer milliseconds."""
    import time
    lq = gpu
    L = (32, 32, 32, 64)
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "eps_CG": 1e-16})
    A = lq.DdagD_operator(D)
    b = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(b, 112)
    x = b.similar()
    res = {}
    for name, fn in (("fp64", lambda: lq.solve_DinvX_(x, A, b, return_info=True)), ("mixed", lambda: lq.solve_mixed_DinvX_(x, A, b, return_info=True))):
        lq.clear_fermion_(x); fn()            # warm (allocations)
        lq.clear_fermion_(x)
        t0 = time.perf_counter(); info = fn(); dt = time.perf_counter() - t0
        r = b.similar()
        lq.mul_(r, A, x)
        lq.add_fermion_(r, -1.0, b)
        res[name] = (dt, info, lq.dot(r, r).real)
        r.close()
    print("fp64 %.1f ms %s true rr %.2e | mixed %.1f ms %s true rr %.2e" % (1e3 * res["fp64"][0], res["fp64"][1], res["fp64"][2],
                                                                           1e3 * res["mixed"][0], res["mixed"][1], res["mixed"][2]))
    assert res["mixed"][2] < 1e-16 and res["fp64"][2] < 2e-16
    assert res["mixed"][0] < res["fp64"][0]
    for o in (x, b, D, U):
        o.close()


def test_mixed_cg_on_non_unitary_links_uses_the_18_real_inner_operator(gpu, orc):
    """with links that fail the unitarity check the fp32 inner operator must read all 18 reals like the fp64 one (a
    rebuilt third row would differ by O(1) and every outer step would stall into the fp64 fall-back)."""
    lq = gpu
    L = (4, 4, 4, 8)
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 131) * (1.0 + 0.05 * orc.gaussian_spinor((4, L[3], L[2], L[1], L[0], 3, 3), 133).real)   # not SU(3)
    Ud = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(Ud, None, {"Dirac_operator": "Wilson", "κ": 0.12, "boundarycondition": BC, "eps_CG": 1e-18})
    b_h = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 132)
    b = lq.Fermionfields(lat, lq.WILSON).upload(b_h)
    x = b.similar()
    A = lq.DdagD_operator(D)
    it, outer, rr = lq.solve_mixed_DinvX_(x, A, b, return_info=True)
    xo, ito, rro, st = orc.cg_DdagD(orc.WILSON, Uh, b_h, L, 0.12, 1.0, BC, eps=1e-18)
    assert st == 0 and rr < 1e-18 and rel_err(x.download(), xo) < 1e-8
    assert lat.get_param("recon_active") == 0
    assert outer >= 2 and it < 3 * ito + 20          # defect correction converges in fp32 steps: no stall, no fp64 fall-back


@pytest.mark.parametrize("kind_name,L", [("Wilson", (16, 8, 8, 4)), ("Wilson", (4, 4, 4, 8)), ("Staggered", (8, 4, 6, 4)), ("WilsonClover", (4, 4, 4, 8))])
@pytest.mark.parametrize("maxiter", [3000, 5, 6])
def test_deferred_x_update_of_the_fp32_solver_gives_identical_results(gpu, orc, kind_name, L, maxiter):
    """mixed_defer_x (default): the fp32 CG updates x every second iteration with both search directions (two p buffers) -- the same
    operations per element in the same order, so a solve returns the same bits with it and without it, for every fp32 field layout (site pairs,
    component pairs, plain), also when the fp32 solve is cut off behind an even or an odd iteration (the owed alpha p is flushed)."""
    lq = gpu
    kind = lq.STAGGERED if kind_name == "Staggered" else lq.WILSON
    lat = lq.Lattice(L)
    Ud = lq.Gaugefields(lat).upload(orc.hot_gauge(L, 171))
    par = {"Dirac_operator": kind_name, "κ": 0.12, "mass": MASS, "boundarycondition": BC, "eps_CG": 1e-18, "MaxCGstep": maxiter}
    if kind_name == "WilsonClover":
        par["Clover_coefficient"] = 1.0
    D = lq.Dirac_operator(Ud, None, par)
    b = lq.Fermionfields(lat, kind).upload(orc.gaussian_spinor(lat.fermion_shape(kind), 172))
    A = lq.DdagD_operator(D)
    out = []
    for defer in (1, 0):
        lat.set_param("mixed_defer_x", defer)
        x = b.similar()
        try:
            info = lq.solve_mixed_DinvX_(x, A, b, return_info=True)
        except lq.NotConverged:
            info = None
        out.append((info, x.download()))
    lat.set_param("mixed_defer_x", 1)
    assert (out[0][0] is None) == (maxiter < 100) and out[0][0] == out[1][0]
    assert np.array_equal(out[0][1], out[1][1])


def test_mixed_precision_evenodd_bicgstab_wilson_clover(gpu, orc):
    """The same route for the Wilson-clover operator (BASELINE configs[3]): fp32 copies of the packed INVERSE clover blocks applied to the hop sums inside the
    fp32 hops (one-site-per-lane build: the site-pair kernel carries no clover term), fp64 defect correction on the clover Schur system."""
    import os
    lq = gpu
    KAPPA, CSW = 0.141139, 1.2
    L = (8, 8, 8, 16)
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    Uh = U.download()
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "WilsonClover", "Clover_coefficient": CSW, "κ": KAPPA, "eps_CG": 1e-19, "MaxCGstep": 3000})
    D.method_CG = "bicgstab_evenodd"
    b = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(b, 112)
    A = orc.clover_build(Uh, L, KAPPA, CSW)
    for dagger in (False, True):
        Dd = D.adjoint() if dagger else D
        Dd.method_CG = "bicgstab_evenodd"
        x64, x32 = b.similar(), b.similar()
        lq.solve_DinvX_(x64, Dd, b)
        lat.set_param("bicg_mixed", 1)
        it32, rr32 = lq.solve_DinvX_(x32, Dd, b, return_info=True)
        lat.set_param("bicg_mixed", 0)
        assert rr32 < 1e-19 and rel_err(x32.download(), x64.download()) < 1e-9
        res = b.download() - orc.wilson_clover_D(Uh, A, x32.download(), L, KAPPA, 1.0, (1, 1, 1, -1), dagger=dagger)
        assert np.vdot(res, res).real < 1e-19
    fa = lq.FermiAction(D)
    eta = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(eta, 114)
    G0, G1 = lq.Gaugefields(lat), lq.Gaugefields(lat)
    S0 = lq.calc_UdSfdU_(G0, fa, U, eta)
    lat.set_param("mixed_action_solver", 1)
    S1 = lq.calc_UdSfdU_(G1, fa, U, eta)
    lat.set_param("mixed_action_solver", 0)
    assert abs(S1 - S0) < 1e-10 * abs(S0) and rel_err(G1.download(), G0.download()) < 1e-8


@pytest.mark.parametrize("links16,reliable", [(1, 1), (0, 0), (1, 0), (0, 1)])
@pytest.mark.parametrize("L,dagger", [((8, 8, 8, 16), False), ((16, 16, 16, 32), True)])
def test_mixed_precision_evenodd_bicgstab_keeps_the_stopping_rule(gpu, orc, L, dagger, links16, reliable):
    """Tunable bicg_mixed: the even-odd BiCGStab of the plain Wilson operator with an fp32 inner chain (same fused structure as the fp64 one) inside an
    fp64 defect correction.  Contract of lqcd_solve_bicgstab_eo unchanged: r.r < eps for the TRUE fp64 residual -- recomputed here by the oracle --
    and the fp64 solver's solution to solver accuracy; at least two correction steps at eps = 1e-19 (an fp32 recurrence gives ~1e-6 per step).
    links16: the site-pair inner operator (16^3 x 32 here; 8^3 x 16 has no whole-chunk planes and keeps the one-site kernel) reads int16 links; reliable: the
    fp32 chain goes on behind a correction step instead of starting again."""
    import os
    lq = gpu
    KAPPA = 0.141139
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    Uh = U.download()
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "eps_CG": 1e-19, "MaxCGstep": 3000})
    Dd = D.adjoint() if dagger else D
    Dd.method_CG = "bicgstab_evenodd"
    b = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(b, 112)
    x64, x32 = b.similar(), b.similar()
    it64, rr64 = lq.solve_DinvX_(x64, Dd, b, return_info=True)
    lat.set_param("bicg_mixed", 1)
    lat.set_param("mixed_links16", links16)
    lat.set_param("bicg_reliable", reliable)
    it32, rr32 = lq.solve_DinvX_(x32, Dd, b, return_info=True)
    lat.set_param("bicg_mixed", 0)
    lat.set_param("bicg_reliable", 0)
    assert lat.get_param("pair32_active") == (1 if L[0] == 16 else 0)
    assert rr32 < 1e-19 and it32 >= it64           # the fp32 iterations of all correction steps together
    assert rel_err(x32.download(), x64.download()) < 1e-9
    orc.set_threads(os.cpu_count() or 1)
    try:
        res = b.download() - orc.wilson_D(Uh, x32.download(), L, KAPPA, 1.0, (1, 1, 1, -1), dagger)
    finally:
        orc.set_threads(1)
    assert np.vdot(res, res).real < 1e-19
    # through the action: mixed_action_solver = 1 takes this route for the plain Wilson operator
    fa = lq.FermiAction(D)
    eta = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(eta, 114)
    S0 = lq.evaluate_FermiAction(fa, U, eta)
    G0, G1 = lq.Gaugefields(lat), lq.Gaugefields(lat)
    lq.calc_UdSfdU_(G0, fa, U, eta)
    lat.set_param("mixed_action_solver", 1)
    S1 = lq.evaluate_FermiAction(fa, U, eta)
    lq.calc_UdSfdU_(G1, fa, U, eta)
    lat.set_param("mixed_action_solver", 0)
    assert abs(S1 - S0) < 1e-10 * abs(S0) and rel_err(G1.download(), G0.download()) < 1e-8


def test_int16_links_of_the_site_pair_operator(gpu, orc):
    """mixed_links16 = 2: the fp32 site-pair kernel reads its links as int16 fixed point (n / 32767).  One application against the ORACLE's fp64 result: the error is
    that of the 1.5e-5 link rounding (well above fp32's, well below a part in 1e4), D and D^+; with fp32 links the same call is fp32-accurate."""
    lq = gpu
    L, KAPPA = (16, 16, 16, 32), 0.141139
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    Uh = U.download()
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA})
    b = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(b, 112)
    bh = b.download()
    out = b.similar()
    for dagger in (False, True):
        ref = orc.wilson_D(Uh, bh, L, KAPPA, 1.0, (1, 1, 1, -1), dagger)
        Dd = D.adjoint() if dagger else D
        err = {}
        for l16 in (0, 2):
            lat.set_param("mixed_links16", l16)
            lq.mul_f32_(out, Dd, b)
            assert lat.get_param("pair32_active") == 1
            err[l16] = np.abs(out.download() - ref).max() / np.abs(ref).max()
        lat.set_param("mixed_links16", 1)
        assert err[0] < 1e-6 and 1e-6 < err[2] < 1e-4, err


def test_mixed_evenodd_chain_merged_update_and_its_guard(gpu, orc):
    """The fp32 chain of the mixed-precision even-odd BiCGStab under bicg_fused = 4 (x / r / p update as one launch on the recurrences for rho' and |r'|^2, default)
    against bicg_fused = 2, and with bicg_rec_guard = 0 (the stopping test never trusts the recurrence: the summed |r'|^2 decides one kernel later): the true fp64
    residual meets the rule every time, the solutions agree to solver accuracy."""
    lq = gpu
    L = (16, 16, 16, 32)
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.141139, "eps_CG": 1e-19, "MaxCGstep": 3000})
    D.method_CG = "bicgstab_evenodd"
    b = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(b, 112)
    lat.set_param("bicg_mixed", 1)
    got = {}
    for fused, guard in ((2, 6), (4, 6), (4, 0), (4, -1)):
        lat.set_param("bicg_fused", fused)
        lat.set_param("bicg_rec_guard", max(guard, 0) if guard != -1 else 6)
        lat.set_param("mixed_lean_residual", 1 if guard == -1 else 0)      # (-1: every correction step starts from the residual alone, as behind a step that was to be the last)
        x = b.similar()
        it, rr = lq.solve_DinvX_(x, D, b, return_info=True)
        assert rr < 1e-19 and lat.get_param("pair32_active") == 1, (fused, guard)
        r = b.similar()
        lq.mul_(r, D, x)
        lq.add_fermion_(r, -1.0, b)
        assert lq.dot(r, r).real < 1e-18, (fused, guard)
        got[(fused, guard)] = x.download()
    lat.set_param("bicg_mixed", 0); lat.set_param("bicg_fused", 4); lat.set_param("bicg_rec_guard", 6); lat.set_param("mixed_lean_residual", 0)
    assert rel_err(got[(4, 6)], got[(2, 6)]) < 1e-9 and rel_err(got[(4, 0)], got[(2, 6)]) < 1e-9 and rel_err(got[(4, -1)], got[(2, 6)]) < 1e-9
