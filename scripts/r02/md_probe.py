#!/usr/bin/env python3
"""Timing of the gauge side of the MD step at 32^3x64 (P_update fused, gauge force, link update) for tunables given as key=value."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import latticeqcd_jl_amd as lq
L = (32, 32, 32, 64)
lat = lq.Lattice(L)
for kv in sys.argv[1:]:
    k, v = kv.split("="); lat.set_param(k, int(v))
U = lq.Gaugefields(lat); lq.lib.check(lq.lib.lib().lqcd_gauge_hot_start(U._h, ctypes.c_uint64(111)))
P, G = lq.Gaugefields(lat), lq.Gaugefields(lat)
lq.gauss_distribution_(P, 5)
def t(fn, reps=10):
    fn(); lat.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    lat.sync(); return 1e3 * (time.perf_counter() - t0) / reps
print(sys.argv[1:], "P_update_fused %.3f ms  gauge_force %.3f ms  U_update %.3f ms  add_ta %.3f ms" % (
    t(lambda: lq.P_update_(U, P, 1e-6, 5.7)), t(lambda: lq.gauge_force_(G, U, 5.7)), t(lambda: lq.U_update_(U, P, 1e-6)), t(lambda: lq.Traceless_antihermitian_add_(P, 1e-6, G))))
