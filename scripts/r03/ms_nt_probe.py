#!/usr/bin/env python3
"""Staggered rational action at 48^3x96 (Nf = 2, m = 0.05, 18 poles, fp64 multi-shift CG): non-temporal streams in the shifted update (nt_blas) on / off"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import latticeqcd_jl_amd as lq
L = (48, 48, 48, 96)
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": 0.05, "eps_CG": 1e-14})
fa = lq.FermiAction(D, {"Nf": 2})
phi = lq.Fermionfields(lat, lq.STAGGERED)
lq.gauss_distribution_fermion_(phi, 112)
S = lq.evaluate_FermiAction(fa, U, phi)
for nt in (1, 0, 1, 0):
    lat.set_param("nt_blas", nt)
    lat.sync(); t0 = time.perf_counter(); S, it = lq.evaluate_FermiAction(fa, U, phi, return_info=True); lat.sync(); ta = time.perf_counter() - t0
    print("nt_blas=%d: S_f = %.12e (iters %d) action %.1f ms" % (nt, S, it, 1e3 * ta), flush=True)
