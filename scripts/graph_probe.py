#!/usr/bin/env python3
"""CG time-to-solution with and without hipGraph replay of the iteration bursts (tunable "graph")."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latticeqcd_jl_amd as lq
import numpy as np
for L, name, eps in (((8, 8, 8, 8), "Staggered", 1e-10), ((8, 8, 8, 8), "Wilson", 1e-19), ((16, 16, 16, 32), "Wilson", 1e-16),
                     ((16, 16, 16, 32), "Staggered", 1e-16), ((32, 32, 32, 64), "Wilson", 1e-16)):
    kind = lq.WILSON if name == "Wilson" else lq.STAGGERED
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    D = lq.Dirac_operator(U, None, {"Dirac_operator": name, "κ": 0.141139, "mass": 0.5, "eps_CG": eps})
    A = lq.DdagD_operator(D)
    b = lq.Fermionfields(lat, kind); lq.gauss_distribution_fermion_(b, 112)
    x = b.similar()
    out = []
    sols = []
    for g in (0, 1):
        lat.set_param("graph", g)
        lq.clear_fermion_(x); lq.solve_DinvX_(x, A, b)
        best = 1e9
        for _ in range(5):
            lq.clear_fermion_(x)
            t0 = time.perf_counter(); info = lq.solve_DinvX_(x, A, b, return_info=True); best = min(best, time.perf_counter() - t0)
        out.append("graph=%d %.3f ms (%d it)" % (g, 1e3 * best, info[0]))
        sols.append(x.download())
    print(name, L, " | ".join(out), "identical" if np.array_equal(sols[0], sols[1]) else "DIFFERENT %.2e" % np.abs(sols[0] - sols[1]).max())
    for o in (x, b, D, U): o.close()
    lat.close()
