#!/usr/bin/env python3
"""Staggered rational action (RHMC, Nf = 2): action and force with the fp64 multi-shift CG vs per-pole mixed-precision solves
(tunable mixed_action_solver).  usage: rhmc_probe.py [L = 48,48,48,96] [mass = 0.05] [eps = 1e-14]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latticeqcd_jl_amd as lq
L = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "48,48,48,96").split(","))
mass = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-14
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": mass, "eps_CG": eps})
fa = lq.FermiAction(D, {"Nf": 2})
print("poles: action %d, MD %d" % (len(fa.rhmc_action[1]), len(fa.rhmc_MD[1])))
phi = lq.Fermionfields(lat, lq.STAGGERED)
lq.gauss_distribution_fermion_(phi, 112)
G = lq.Gaugefields(lat)
ref = None
for mode in (0, 2, 1):
    lat.set_param("mixed_action_solver", mode)
    S = lq.evaluate_FermiAction(fa, U, phi)
    lat.sync(); t0 = time.perf_counter(); S, it = lq.evaluate_FermiAction(fa, U, phi, return_info=True); lat.sync(); ta = time.perf_counter() - t0
    lq.calc_UdSfdU_(G, fa, U, phi)
    lat.sync(); t0 = time.perf_counter(); lq.calc_UdSfdU_(G, fa, U, phi); lat.sync(); tf = time.perf_counter() - t0
    g = lq.momentum_action(G) if hasattr(lq, "momentum_action") else 0.0
    print("mixed_action_solver=%d: S_f = %.12e (iters %d) action %.1f ms, force %.1f ms, |G|^2-like %.10e" % (mode, S, it, 1e3 * ta, 1e3 * tf, g))
