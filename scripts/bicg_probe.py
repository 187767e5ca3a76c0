#!/usr/bin/env python3
"""Even-odd BiCGStab probe (config 3 lattice by default): where an iteration's time goes in each form of the chain (tunable bicg_fused), and the first
iteration at which two forms differ.  usage: bicg_probe.py [mode ...] [--L x,y,z,t] [--diff] [--reps n] [--kappa k] [--csw c] [--set key=value ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latticeqcd_jl_amd as lq  # noqa: E402

args = sys.argv[1:]
L = (16, 16, 16, 32)
if "--L" in args:
    L = tuple(int(v) for v in args[args.index("--L") + 1].split(","))
reps = int(args[args.index("--reps") + 1]) if "--reps" in args else 20
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
while "--set" in args:      # --set key=value: library tunable
    i = args.index("--set")
    k, v = args[i + 1].split("=")
    lat.set_param(k, int(v))
    del args[i:i + 2]
csw = float(args[args.index("--csw") + 1]) if "--csw" in args else 0.0
kappa = float(args[args.index("--kappa") + 1]) if "--kappa" in args else 0.141139
for opt in ("--L", "--reps", "--csw", "--kappa"):      # (what is left are the one-digit modes)
    if opt in args:
        i = args.index(opt)
        del args[i:i + 2]
D = lq.Dirac_operator(U, None, {"Dirac_operator": "WilsonClover" if csw else "Wilson", "Clover_coefficient": csw, "κ": kappa, "eps_CG": 1e-16, "MaxCGstep": 3000})
D.method_CG = "bicgstab_evenodd"
b = lq.Fermionfields(lat, lq.WILSON)
lq.gauss_distribution_fermion_(b, 112)
x = b.similar()
if "--repeat" in args:      # run-to-run determinism of each form at the test's tolerance
    D.eps_CG = 1e-19
    for m in (3, 2, 1, 3, 2, 1, 0, 0):
        lat.set_param("bicg_fused", m)
        for rep in range(3):
            lq.clear_fermion_(x)
            it, rr = lq.solve_DinvX_(x, D, b, return_info=True)
            print("bicg_fused %d run %d: %d iterations, r.r %.17e, |x|^2 %.17e" % (m, rep, it, rr, lq.dot(x, x).real))
    sys.exit(0)
if "--diff" in args:
    D.eps_CG = 1e-40
    for k in range(1, 8):
        D.MaxCGstep = k
        got = {}
        for m in (1, 2):
            lat.set_param("bicg_fused", m)
            lq.clear_fermion_(x)
            try:
                lq.solve_DinvX_(x, D, b)
            except lq.NotConverged:
                pass
            got[m] = x.download()
        d = np.abs(got[1] - got[2]).max()
        print("after %d iterations: max |x(unfolded) - x(folded)| = %.3e (relative %.3e)" % (k, d, d / np.abs(got[1]).max()))
    sys.exit(0)
import time
modes = [int(a) for a in args if a.isdigit() and len(a) == 1] or [2]
for m in modes:
    lat.set_param("bicg_fused", m)
    lq.clear_fermion_(x)
    lq.solve_DinvX_(x, D, b)
    lat.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        lq.clear_fermion_(x)
        it, rr = lq.solve_DinvX_(x, D, b, return_info=True)
    dt = (time.perf_counter() - t0) / reps
    print("bicg_fused %d: %d iterations, %.3f ms per solve, %.1f us per iteration, r.r %.3e" % (m, it, 1e3 * dt, 1e6 * dt / it, rr))
