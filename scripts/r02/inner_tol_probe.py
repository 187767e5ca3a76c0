#!/usr/bin/env python3
"""mixed-precision CG: time and outer steps against the tolerance asked of each fp32 solve (32^3x64 Wilson and staggered, 1e-16)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import latticeqcd_jl_amd as lq
L = (32, 32, 32, 64)
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
for kind_name, kind in (("Wilson", lq.WILSON), ("Staggered", lq.STAGGERED)):
    D = lq.Dirac_operator(U, None, {"Dirac_operator": kind_name, "κ": 0.141139, "mass": 0.05, "eps_CG": 1e-16})
    A = lq.DdagD_operator(D)
    b = lq.Fermionfields(lat, kind); lq.gauss_distribution_fermion_(b, 112)
    x = b.similar()
    for tol in (1e-3, 3e-4, 1e-4, 3e-5, 1e-5, 3e-6, 1e-6):
        best = 1e9
        for _ in range(3):
            lq.clear_fermion_(x)
            t0 = time.perf_counter(); info = lq.solve_mixed_DinvX_(x, A, b, inner_tol=tol, return_info=True); best = min(best, time.perf_counter() - t0)
        print("%s inner_tol %.0e: %.2f ms  inner its %d outer %d rr %.2e" % (kind_name, tol, 1e3 * best, *info))
