#!/usr/bin/env python3
"""Staggered 4-taste action (pseudofermion on the even sites): full-lattice CG vs the half-lattice parity-block CG.
usage: parity_probe.py [L = 48,48,48,96] [mass = 0.05] [eps = 1e-14]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latticeqcd_jl_amd as lq
L = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "48,48,48,96").split(","))
mass = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-14
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": mass, "eps_CG": eps})
fa = lq.FermiAction(D, {"Nf": 4})
xi, phi = lq.Fermionfields(lat, lq.STAGGERED), lq.Fermionfields(lat, lq.STAGGERED)
lq.gauss_sampling_in_action_(xi, U, fa, 113)
lq.sample_pseudofermions_(phi, U, fa, xi)
for mode in (0, 1):
    lat.set_param("staggered_parity_solve", mode)
    lq.evaluate_FermiAction(fa, U, phi)
    lat.sync(); t0 = time.perf_counter(); S, it = lq.evaluate_FermiAction(fa, U, phi, return_info=True); lat.sync()
    print("staggered_parity_solve=%d: S_f = %.12e, %d iterations, %.1f ms" % (mode, S, it, 1e3 * (time.perf_counter() - t0)))
