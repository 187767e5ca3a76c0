#!/usr/bin/env python3
"""mixed-precision CG at 32^3x64 with and without the x update fused into the update-mode D^+ of the site-pair kernel (mixed_xfuse)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import latticeqcd_jl_amd as lq
L = (32, 32, 32, 64)
KEY = sys.argv[1] if len(sys.argv) > 1 else "mixed_xfuse"
lat = lq.Lattice(L)
U = lq.Gaugefields(lat)
lq.lib.check(lq.lib.lib().lqcd_gauge_hot_start(U._h, ctypes.c_uint64(111)))
D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.141139, "eps_CG": 1e-16, "MaxCGstep": 2000})
b = lq.Fermionfields(lat, lq.WILSON)
lq.gauss_distribution_fermion_(b, 112)
x = b.similar()
A = lq.DdagD_operator(D)
for fuse in (0, 1, 0, 1, 0, 1):
    lat.set_param(KEY, fuse)
    best, info = 1e9, None
    for rep in range(5):
        lq.clear_fermion_(x); lat.sync()
        t0 = time.perf_counter(); info = lq.solve_mixed_DinvX_(x, A, b, return_info=True); lat.sync()
        best = min(best, 1e3 * (time.perf_counter() - t0))
    print(KEY + " %d: %.1f ms, inner %d, outer %d, true rr %.2e" % (fuse, best, info[0], info[1], info[2]), flush=True)
lq.clear_fermion_(x); lat.sync()
t0 = time.perf_counter(); i64 = lq.solve_DinvX_(x, A, b, return_info=True); lat.sync(); t64 = 1e3 * (time.perf_counter() - t0)
lq.clear_fermion_(x); lat.sync()
t0 = time.perf_counter(); i64 = lq.solve_DinvX_(x, A, b, return_info=True); lat.sync(); t64 = min(t64, 1e3 * (time.perf_counter() - t0))
print("fp64 CG: %.1f ms, %d iterations" % (t64, i64[0]))
