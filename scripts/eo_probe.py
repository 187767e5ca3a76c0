#!/usr/bin/env python3
"""D x = b for Wilson / Wilson-clover: plain BiCGStab vs the even-odd (Schur) preconditioned one (BASELINE configs[2], configs[3]).
usage: eo_probe.py [L = 16,16,16,32] [eps = 1e-16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latticeqcd_jl_amd as lq
L = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "16,16,16,32").split(","))
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-16
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
b = lq.Fermionfields(lat, lq.WILSON)
lq.gauss_distribution_fermion_(b, 112)
x, r = b.similar(), b.similar()
for name in ("Wilson", "WilsonClover"):
    D = lq.Dirac_operator(U, None, {"Dirac_operator": name, "κ": 0.141139, "Clover_coefficient": 1.0, "eps_CG": eps})
    for method in ("bicgstab", "bicgstab_evenodd"):
        D.method_CG = method
        lq.clear_fermion_(x); lq.solve_DinvX_(x, D, b)
        best = 1e9
        for _ in range(3):
            lq.clear_fermion_(x)
            t0 = time.perf_counter(); info = lq.solve_DinvX_(x, D, b, return_info=True); best = min(best, time.perf_counter() - t0)
        lq.mul_(r, D, x); lq.add_fermion_(r, -1.0, b)
        print("%s %s %-17s %.2f ms iters=%d true_rr=%.3e" % (name, L, method, 1e3 * best, info[0], lq.dot(r, r).real))
    D.close()
