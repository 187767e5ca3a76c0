#!/usr/bin/env python3
"""mixed-precision CG at 32^3x64: inner tolerance of the fp32 solves against outer steps / inner iterations / time (r.r < 1e-16, hot start)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import latticeqcd_jl_amd as lq
L = (32, 32, 32, 64)
lat = lq.Lattice(L)
U = lq.Gaugefields(lat)
lq.lib.check(lq.lib.lib().lqcd_gauge_hot_start(U._h, ctypes.c_uint64(111)))
D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.141139, "eps_CG": 1e-16, "MaxCGstep": 2000})
b = lq.Fermionfields(lat, lq.WILSON)
lq.gauss_distribution_fermion_(b, 112)
x = b.similar()
A = lq.DdagD_operator(D)
for tol in (0.0, 1e-4, 1e-6, 0.0):
    best, info = 1e9, None
    for rep in range(4):
        lq.clear_fermion_(x); lat.sync()
        t0 = time.perf_counter(); info = lq.solve_mixed_DinvX_(x, A, b, inner_tol=tol, return_info=True); lat.sync()
        best = min(best, 1e3 * (time.perf_counter() - t0))
    print("inner_tol %.0e: %.1f ms, inner %d, outer %d, true rr %.2e" % (tol, best, info[0], info[1], info[2]), flush=True)
