#!/usr/bin/env python3
"""Secondary measurements for the BASELINE.json configurations that are parity-test cases rather than the bench line:
  configs[1]  8^4 staggered Dslash + CG to 1e-10 (hot start)
  configs[2]  16^3x32 Wilson, even-odd preconditioned BiCGStab vs plain BiCGStab vs CG on D^+D
  8(f) rank 3 multi-shift CG (16^3x32 staggered, 10 shifts)
Prints one JSON object per configuration (time-to-solution, iterations, Dslash roofline fraction).  Not the driver's bench.py."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latticeqcd_jl_amd as lq  # noqa: E402

KAPPA, MASS = 0.141139, 0.5


def timed(fn, reps=3):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best, out


def main():
    res = []
    # ---- configs[1]: 8^4 staggered
    L = (8, 8, 8, 8)
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": MASS, "eps_CG": 1e-10})
    b = lq.Fermionfields(lat, lq.STAGGERED)
    lq.gauss_distribution_fermion_(b, 112)
    x, y = b.similar(), b.similar()
    ms = lq.bench_dslash(D, y, b, warm=50, reps=500)
    V = 8 ** 4

    def solve():
        lq.clear_fermion_(x)
        return lq.solve_DinvX_(x, lq.DdagD_operator(D), b, return_info=True)
    dt, (it, rr) = timed(solve)
    res.append({"config": "8^4 staggered Dslash + CG to 1e-10, fp64, hot start", "dslash_us": 1e3 * ms,
                "dslash_gflops": 570 * V / ms / 1e6, "roofline_frac_672B": 672 * V / ms / 1e6 / 8000, "cg_iters": it,
                "cg_final_rr": rr, "cg_ms": 1e3 * dt, "note": "4096 sites: launch-latency bound, 64 workgroups"})
    # ---- configs[2]: 16^3x32 Wilson, even-odd BiCGStab
    L = (16, 16, 16, 32)
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "eps_CG": 1e-16})
    b = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(b, 112)
    x, y = b.similar(), b.similar()
    ms = lq.bench_dslash(D, y, b, warm=50, reps=500)
    V = 16 ** 3 * 32
    out = {"config": "16^3x32 Wilson Dslash + BiCGStab (even-odd preconditioned), fp64, hot start", "dslash_us": 1e3 * ms,
           "dslash_gflops": 1320 * V / ms / 1e6, "roofline_frac_960B": 960 * V / ms / 1e6 / 8000}
    # even-odd BiCGStab in its three forms (tunable bicg_fused): 2 = inner products from the Schur operator's epilogue + reductions and scalar steps in the
    # consumers' prologues (default), 1 = epilogue products with separate reduction launches, 0 = the generic chain
    D.method_CG = "bicgstab_evenodd"
    for mode in (0, 1, 2):
        lat.set_param("bicg_fused", mode)

        def solve_eo():
            lq.clear_fermion_(x)
            return lq.solve_DinvX_(x, D, b, return_info=True)
        dt, (it, rr) = timed(solve_eo, reps=5)
        out["bicgstab_evenodd_fused%d" % mode] = {"iters": it, "final_rr": rr, "ms": 1e3 * dt, "us_per_iteration": 1e6 * dt / max(it, 1)}
    for method in ("bicgstab_evenodd", "bicgstab"):
        D.method_CG = method

        def solve():
            lq.clear_fermion_(x)
            return lq.solve_DinvX_(x, D, b, return_info=True)
        dt, (it, rr) = timed(solve)
        out[method] = {"iters": it, "final_rr": rr, "ms": 1e3 * dt}

    def solve_cg():
        lq.clear_fermion_(x)
        return lq.solve_DinvX_(x, lq.DdagD_operator(D), b, return_info=True)
    dt, (it, rr) = timed(solve_cg)
    out["cg_DdagD"] = {"iters": it, "final_rr": rr, "ms": 1e3 * dt, "iters_per_s": it / dt}
    res.append(out)
    # ---- multi-shift CG, staggered 16^3x32, 10 shifts
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": 0.05, "eps_CG": 1e-16})
    b = lq.Fermionfields(lat, lq.STAGGERED)
    lq.gauss_distribution_fermion_(b, 112)
    sig = [1e-4 * 3 ** k for k in range(10)]
    xs = [b.similar() for _ in sig]
    x0 = b.similar()
    A = lq.DdagD_operator(D)
    dt, (it, rr) = timed(lambda: lq.shiftedcg(xs, sig, x0, A, b, return_info=True))
    dts, (its, rrs) = timed(lambda: (lq.clear_fermion_(x0), lq.solve_DinvX_(x0, A, b, return_info=True))[1])
    row = {"config": "16^3x32 staggered multi-shift CG, 10 shifts, mass 0.05 (RHMC solver)", "iters": it, "resid": rr,
           "ms": 1e3 * dt, "single_cg_iters": its, "single_cg_ms": 1e3 * dts,
           "cost_vs_10_separate_solves": dt / (10 * dts)}
    # the mixed-precision form (fp32 multi-shift pass + fp64 defect correction per shift) at the same target and at an MD-force target
    bb = lq.dot(b, b).real
    for tag, eps in (("mixed_same_target", 1e-16), ("mixed_force_target_rel1e-6", 1e-12 * bb)):
        dtm, (itm, outm, rrm) = timed(lambda: lq.shiftedcg_mixed(xs, sig, None, A, b, eps=eps, return_info=True))
        dt64, (it64, rr64) = timed(lambda: lq.shiftedcg(xs, sig, None, A, b, eps=eps, return_info=True))
        row[tag] = {"eps": eps, "ms": 1e3 * dtm, "fp32_iters": itm, "correction_solves": outm, "worst_true_rr": rrm,
                    "fp64_ms_same_eps": 1e3 * dt64, "fp64_iters": it64, "speedup": dt64 / dtm}
    res.append(row)
    for o in (U, D, b, x0, *xs):
        o.close()
    # ---- fermion force sweep at 32^3x64 (Wilson) -- 1536 B/site compulsory
    L = (32, 32, 32, 64)
    V = 32 ** 3 * 64
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA})
    X, Y = lq.Fermionfields(lat, lq.WILSON), lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(X, 1)
    lq.gauss_distribution_fermion_(Y, 2)
    G = lq.Gaugefields(lat)
    lq.fermion_force_(G, D, X, Y)
    dt, _ = timed(lambda: [lq.fermion_force_(G, D, X, Y) for _ in range(10)], reps=3)
    ms = 1e3 * dt / 10
    res.append({"config": "32^3x64 Wilson fermion-force sweep (calc_UdSfdU! after the solve)", "ms": ms,
                "algorithmic_GBps_1536B": 1536 * V / ms / 1e6, "roofline_frac": 1536 * V / ms / 1e6 / 8000})
    # ---- one MD step of the 2-flavour Wilson HMC at 32^3x64, everything resident (reference: runMD_QPQ_sw!, N_sw = 10)
    beta, dtau, nsw = 5.7, 0.05, 10
    p = lq.Gaugefields(lat)
    lq.gauss_distribution_(p, 7)
    eta = lq.Fermionfields(lat, lq.WILSON)
    lq.sample_pseudofermions_(eta, U, lq.FermiAction(D), X)
    fa = lq.FermiAction(D)
    D.eps_CG = 1e-16

    def tk(fn, reps=5):
        fn()
        return 1e3 * timed(lambda: [fn() for _ in range(reps)], reps=2)[0] / reps
    lat.set_param("lazy_merge", 0)                 # unit timings: every link update launches at once
    t_gf = tk(lambda: lq.gauge_force_(G, U, beta))
    t_ta = tk(lambda: lq.Traceless_antihermitian_add_(p, 1e-9, G))
    t_pu = tk(lambda: lq.P_update_(U, p, 1e-9, beta))
    t_up = tk(lambda: lq.U_update_(U, p, 1e-9))
    t_ff = tk(lambda: lq.calc_UdSfdU_(G, fa, U, eta), reps=2)
    t_sf = tk(lambda: lq.evaluate_FermiAction(fa, U, eta), reps=2)
    it_eo = lq.evaluate_FermiAction(fa, U, eta, return_info=True)[1]
    lat.set_param("action_eo_solver", 0)        # the reference's form: CG on the normal equations
    t_ff_cg = tk(lambda: lq.calc_UdSfdU_(G, fa, U, eta), reps=2)
    it_cg = lq.evaluate_FermiAction(fa, U, eta, return_info=True)[1]
    lat.set_param("action_eo_solver", 1)
    lat.set_param("mixed_action_solver", 1)        # plain Wilson: the even-odd solves with the fp32 inner chain (bicg_mixed)
    t_ffm = tk(lambda: lq.calc_UdSfdU_(G, fa, U, eta), reps=2)
    it_mx = lq.evaluate_FermiAction(fa, U, eta, return_info=True)[1]
    lat.set_param("action_eo_solver", 0)           # ... and the mixed-precision CG on the normal equations it replaces
    t_ffm_cg = tk(lambda: lq.calc_UdSfdU_(G, fa, U, eta), reps=2)
    lat.set_param("action_eo_solver", 1)
    lat.set_param("mixed_action_solver", 0)

    def md_step():                                 # runMD_QPQ_sw!, one itrj (standardMD.jl:146-166); steps of 1e-9 keep the configuration where it is
        for half in range(2):
            for _ in range(nsw // 2):
                lq.U_update_(U, p, 0.5e-9)
                lq.P_update_(U, p, 1e-9, beta)
                lq.U_update_(U, p, 0.5e-9)
            if half == 0:
                lq.calc_UdSfdU_(G, fa, U, eta)
                lq.Traceless_antihermitian_add_(p, 1e-9, G)
    md_ms = {}
    for merge in (0, 1, 2):                        # 0: every update launches at once; 1: back-to-back link updates merge; 2 (default): + momentum and link update in one sweep
        lat.set_param("lazy_merge", merge)
        for mixed in (0, 1):
            lat.set_param("mixed_action_solver", mixed)
            md_step(); lq.calculate_Plaquette(U)
            md_ms[(merge, mixed)] = 1e3 * timed(lambda: [md_step(), md_step(), md_step(), lq.unitarity_deviation(U)], reps=2)[0] / 3
    lat.set_param("mixed_action_solver", 0)
    res.append({"config": "32^3x64 Wilson HMC, one MD step resident on the device (Sexton-Weingarten N = 10)",
                "md_step_measured_ms": md_ms[(2, 0)], "md_step_measured_mixed_precision_solver_ms": md_ms[(2, 1)],
                "md_step_measured_lazy_merge1_ms": md_ms[(1, 0)], "md_step_measured_lazy_merge1_mixed_ms": md_ms[(1, 1)],
                "md_step_measured_lazy_merge0_ms": md_ms[(0, 0)], "md_step_measured_lazy_merge0_mixed_ms": md_ms[(0, 1)],
                "gauge_force_ms": t_gf, "gauge_force_GBps_1152B": 1152 * V / t_gf / 1e6, "momentum_add_ta_ms": t_ta,
                "P_update_fused_ms": t_pu, "P_update_fused_GBps_1728B": 1728 * V / t_pu / 1e6, "link_exp_update_ms": t_up,
                "link_exp_update_GBps_1728B": 1728 * V / t_up / 1e6,
                "calc_UdSfdU_ms (two even-odd BiCGStab solves to 1e-16 + sweep)": t_ff, "action_solver_iterations_evenodd_bicgstab": it_eo,
                "calc_UdSfdU_ms_action_eo_solver0 (CG to 1e-16 + Y = D X + sweep)": t_ff_cg, "action_solver_iterations_cg": it_cg,
                "evaluate_FermiAction_ms": t_sf,
                "calc_UdSfdU_mixed_precision_solver_ms": t_ffm, "mixed_evenodd_fp32_iterations": it_mx, "calc_UdSfdU_mixed_precision_cg_ms": t_ffm_cg,
                "md_step_sum_of_parts_ms": nsw * (t_pu + 2 * t_up) + t_ff + t_ta,
                "md_step_sum_of_parts_mixed_precision_solver_ms": nsw * (t_pu + 2 * t_up) + t_ffm + t_ta,
                "note": "host<->device traffic per MD step: none (the reference path would move 1.2 GB of links + 2 spinors)"})
    # ---- configs[3]: the same force evaluation with the Wilson-clover operator (even-odd solves through the inverse clover blocks vs CG)
    Dc = lq.Dirac_operator(U, None, {"Dirac_operator": "WilsonClover", "κ": KAPPA, "Clover_coefficient": 1.0, "eps_CG": 1e-16})
    fac = lq.FermiAction(Dc)
    lq.sample_pseudofermions_(eta, U, fac, X)
    t_c1 = tk(lambda: lq.calc_UdSfdU_(G, fac, U, eta), reps=2)
    it_c1 = lq.evaluate_FermiAction(fac, U, eta, return_info=True)[1]
    lat.set_param("action_eo_solver", 0)
    t_c0 = tk(lambda: lq.calc_UdSfdU_(G, fac, U, eta), reps=2)
    it_c0 = lq.evaluate_FermiAction(fac, U, eta, return_info=True)[1]
    lat.set_param("mixed_action_solver", 1)
    t_cm0 = tk(lambda: lq.calc_UdSfdU_(G, fac, U, eta), reps=2)          # mixed-precision CG on the normal equations (action_eo_solver is still 0)
    lat.set_param("action_eo_solver", 1)
    t_cm1 = tk(lambda: lq.calc_UdSfdU_(G, fac, U, eta), reps=2)          # even-odd solves with the fp32 inner chain
    it_cm1 = lq.evaluate_FermiAction(fac, U, eta, return_info=True)[1]
    lat.set_param("mixed_action_solver", 0)
    res.append({"config": "32^3x64 Wilson-clover (c_sw = 1) force evaluation calc_UdSfdU!, eps 1e-16", "two_evenodd_bicgstab_solves_ms": t_c1, "iterations": it_c1,
                "cg_normal_equations_ms": t_c0, "cg_iterations": it_c0, "mixed_precision_cg_ms": t_cm0, "mixed_precision_evenodd_ms": t_cm1, "mixed_evenodd_fp32_iterations": it_cm1})
    # ---- beyond SURVEY 8: the stout layer of the fermion action's links and its back-propagation at 32^3x64; the Domainwall operator at 16^3x32 x L5 = 8
    nn = lq.CovNeuralnet(U)
    nn.push_(lq.STOUT_Layer(["plaquette"], [0.1], U))
    t_sm = tk(lambda: lq.calc_smearedU(U, nn))
    Uout, multi, _ = lq.calc_smearedU(U, nn)
    dS = lq.Gaugefields(lat)
    lq.gauss_distribution_(dS, 9)
    t_bp = tk(lambda: lq.back_prop(dS, nn, multi, U))
    res.append({"config": "32^3x64 stout layer (plaquette, rho 0.1) of the fermion action's links", "calc_smearedU_ms": t_sm, "back_prop_ms": t_bp,
                "note": "smearing = staple sweep + exp; back_prop = 8 link products + staple sweep + Frechet-derivative pass + 24-loop gather"})
    for o in (U, D, X, Y, G, p, eta, dS):
        o.close()
    L = (16, 16, 16, 32)
    V5 = 16 ** 3 * 32
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    L5 = 8
    x5 = lq.Initialize_pseudofermion_fields(U[1], "Domainwall", L5=L5)
    D5 = lq.Dirac_operator(U, x5, {"Dirac_operator": "Domainwall", "mass": 0.05, "L5": L5, "M": -1.8, "eps_CG": 1e-16})
    lq.gauss_distribution_fermion_(x5, 3)
    y5 = x5.similar()
    t_d5 = tk(lambda: lq.mul_(y5, D5, x5), reps=20)
    z5 = x5.similar()
    dt5, (it5, rr5) = timed(lambda: (lq.clear_fermion_(z5), lq.solve_DinvX_(z5, lq.DdagD_operator(D5), x5, return_info=True))[1], reps=2)
    res.append({"config": "16^3x32 x L5 = 8 Domainwall (M = -1.8, m = 0.05): D5 application and CG on D5^+ D5 to 1e-16", "D5_ms": t_d5,
                "D5_GBps_on_minimal_bytes (384 + 384/L5 per 5-d site: psi in + out, 12-real links shared by the slices)": L5 * V5 * (384 + 384 / L5) / t_d5 / 1e6,
                "cg_iters": it5, "cg_ms": 1e3 * dt5, "cg_final_rr": rr5})
    for o in (U, D5, x5, y5, z5):
        o.close()
    # ---- configs[4] geometry on one GPU: 48^3x96 staggered Dslash and CG (fp64)
    L = (48, 48, 48, 96)
    V = 48 ** 3 * 96
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": MASS})
    b = lq.Fermionfields(lat, lq.STAGGERED)
    lq.gauss_distribution_fermion_(b, 112)
    x, y = b.similar(), b.similar()
    ms = lq.bench_dslash(D, y, b, warm=10, reps=100)
    msi = lq.bench_cg(D, x, b, warm=3, niter=50)
    res.append({"config": "48^3x96 staggered Dslash + CG window, fp64, one GPU", "dslash_ms": ms, "dslash_gflops": 570 * V / ms / 1e6,
                "roofline_frac_672B": 672 * V / ms / 1e6 / 8000, "cg_iters_per_s": 1e3 / msi})
    for r in res:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
