#!/usr/bin/env python3
"""mixed-precision CG against the fp64 CG at 32^3x64 (and optionally 48^3x96 staggered is left to scripts/mixed_probe.py): time to |r|^2 < 1e-16,
several repetitions, plus the fp32 Dslash kernel time from the library's bench entry point.  usage: mixed_ab.py [key=value ...]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import latticeqcd_jl_amd as lq
L = (32, 32, 32, 64)
lat = lq.Lattice(L)
for kv in sys.argv[1:]:
    k, v = kv.split("="); lat.set_param(k, int(v))
U = lq.Gaugefields(lat)
lq.lib.check(lq.lib.lib().lqcd_gauge_hot_start(U._h, ctypes.c_uint64(111)))
D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.141139, "eps_CG": 1e-16, "MaxCGstep": 2000})
b = lq.Fermionfields(lat, lq.WILSON)
lq.gauss_distribution_fermion_(b, 112)
x = b.similar()
def timed(fn):
    lq.lib.check(lq.lib.lib().lqcd_spinor_zero(x._h)); lat.sync(); t0 = time.perf_counter(); r = fn(); lat.sync(); return 1e3 * (time.perf_counter() - t0), r
tm, t64 = [], []
for rep in range(4):
    a, info = timed(lambda: lq.solve_mixed_DinvX_(x, lq.DdagD_operator(D), b, return_info=True)); tm.append(a)
    c, info64 = timed(lambda: lq.solve_DinvX_(x, lq.DdagD_operator(D), b, return_info=True)); t64.append(c)
print(sys.argv[1:], "mixed %s ms (inner %d, outer %d)  fp64 %s ms (%d it)  best ratio %.2f" % (
    ["%.1f" % v for v in tm], info[0], info[1], ["%.1f" % v for v in t64], info64[0], min(t64) / min(tm)), flush=True)
