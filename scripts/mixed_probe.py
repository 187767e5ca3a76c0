#!/usr/bin/env python3
"""32^3x64 Wilson: fp64 CG vs mixed-precision CG to the same true residual (profiling driver)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latticeqcd_jl_amd as lq
L = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "32,32,32,64").split(","))
kind_name = sys.argv[2] if len(sys.argv) > 2 else "Wilson"
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-16
kind = lq.WILSON if kind_name.startswith("Wilson") else lq.STAGGERED
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
if os.environ.get("LQCD_FORCE_PARTITION"):
    lat.comm_init(lq.comm_unique_id())          # world-size-1 communicators: the real RCCL path against this rank itself
for kv in os.environ.get("LQCD_SET", "").split():
    k, v = kv.split("=")
    lat.set_param(k, int(v))
D = lq.Dirac_operator(U, None, {"Dirac_operator": kind_name, "κ": 0.141139, "mass": 0.05, "eps_CG": eps, "Clover_coefficient": 1.0})
A = lq.DdagD_operator(D)
b = lq.Fermionfields(lat, kind)
lq.gauss_distribution_fermion_(b, 112)
x = b.similar()
for name, fn in (("fp64", lambda: lq.solve_DinvX_(x, A, b, return_info=True)), ("mixed", lambda: lq.solve_mixed_DinvX_(x, A, b, return_info=True))):
    lq.clear_fermion_(x); fn()
    best = 1e9
    for _ in range(3):
        lq.clear_fermion_(x)
        t0 = time.perf_counter(); info = fn(); best = min(best, time.perf_counter() - t0)
    r = b.similar(); lq.mul_(r, A, x); lq.add_fermion_(r, -1.0, b)
    print("%s %s %s: %.2f ms info=%s true_rr=%.3e" % (kind_name, L, name, 1e3 * best, info, lq.dot(r, r).real))
