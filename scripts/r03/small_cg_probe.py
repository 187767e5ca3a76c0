#!/usr/bin/env python3
"""8^4 (and neighbours) staggered CG to 1e-10 from a hot start: one-launch form (cg_persist = 1) against the launch chain (0); wall time of the
whole lqcd_solve_cg_DdagD call, best of several."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import latticeqcd_jl_amd as lq
for L in ((8, 8, 8, 8), (16, 8, 8, 16), (4, 4, 4, 4)):
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": 0.5, "eps_CG": 1e-10, "MaxCGstep": 3000})
    b = lq.Fermionfields(lat, lq.STAGGERED)
    lq.gauss_distribution_fermion_(b, 112)
    A = lq.DdagD_operator(D)
    x = b.similar()
    for mode in (1, 0, 1):
        lat.set_param("cg_persist", mode)
        best, info = 1e9, None
        for rep in range(30):
            lq.clear_fermion_(x); lat.sync()
            t0 = time.perf_counter(); info = lq.solve_DinvX_(x, A, b, return_info=True); lat.sync()
            best = min(best, 1e3 * (time.perf_counter() - t0))
        if mode in (0, 1):
            tw = {}
            for n in (10, 110):
                bw = 1e9
                for rep in range(20):
                    lq.clear_fermion_(x); lat.sync()
                    t0 = time.perf_counter(); lq.lib.check(lq.lib.lib().lqcd_solve_cg_DdagD_fixed(D._h, x._h, b._h, n)); lat.sync()
                    bw = min(bw, 1e3 * (time.perf_counter() - t0))
                tw[n] = bw
            print("   fixed windows: 10 it %.3f ms, 110 it %.3f ms -> %.2f us / iteration, fixed part %.1f us" % (tw[10], tw[110], 10 * (tw[110] - tw[10]), 1e3 * tw[10] - 100 * (tw[110] - tw[10])), flush=True)
        print("L=%s cg_persist=%d: %.3f ms, %d iterations (%.2f us / iteration), rr %.2e" % (L, mode, best, info[0], 1e3 * best / max(info[0], 1), info[1]), flush=True)
