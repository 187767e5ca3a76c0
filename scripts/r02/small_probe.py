#!/usr/bin/env python3
"""8^4 staggered CG to 1e-10 (BASELINE configs[1] scale): solve time, iterations (launch-latency bound: 64 workgroups)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import latticeqcd_jl_amd as lq
L = (8, 8, 8, 8)
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
for kv in os.environ.get("LQCD_SET", "").split():
    k, v = kv.split("="); lat.set_param(k, int(v))
D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": 0.5, "eps_CG": 1e-10})
A = lq.DdagD_operator(D)
b = lq.Fermionfields(lat, lq.STAGGERED); lq.gauss_distribution_fermion_(b, 112)
x = b.similar()
def solve():
    lq.clear_fermion_(x)
    return lq.solve_DinvX_(x, A, b, return_info=True)
solve(); ts = []
for _ in range(20):
    t0 = time.perf_counter(); info = solve(); ts.append(time.perf_counter() - t0)
ts.sort()
print(json.dumps({"config": "8^4 staggered CG to 1e-10", "set": os.environ.get("LQCD_SET", ""), "ms_median": 1e3 * ts[len(ts) // 2], "ms_min": 1e3 * ts[0], "iters": info[0], "rr": info[1]}))
