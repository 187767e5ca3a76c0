#!/usr/bin/env python3
"""Long HMC streams on the device for the statistical parity table of LABNOTES.md section 4.
usage: hmc_stats.py [--ntraj 300] [--therm 30] [--actions a,b,...] [--out file.json]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import latticeqcd_jl_amd as lq
import hmc_harness as hh

ap = argparse.ArgumentParser()
ap.add_argument("--ntraj", type=int, default=300)
ap.add_argument("--therm", type=int, default=30)
ap.add_argument("--actions", default=",".join(hh.ACTIONS))
ap.add_argument("--seed", type=int, default=7)
ap.add_argument("--out", default="")
a = ap.parse_args()
res = {}
for act in a.actions.split(","):
    t0 = time.time()
    r = hh.run_stream(lq, act, a.therm + a.ntraj, a.seed)
    dt = time.time() - t0
    P, eP = hh.binned(r["plaq"][a.therm:])
    E, eE = hh.binned(np.exp(-r["dH"][a.therm:]))
    res[act] = {"plaquette": P, "plaquette_err": eP, "exp_mdH": E, "exp_mdH_err": eE, "acceptance": float(r["accepted"][a.therm:].mean()),
                "mean_abs_dH": float(np.abs(r["dH"][a.therm:]).mean()), "ntraj": a.ntraj, "therm": a.therm, "seconds_per_trajectory": dt / (a.therm + a.ntraj),
                "first10_final_plaquette": float(r["plaq"][9])}
    print("%-26s <P> = %.5f +- %.5f   <exp(-dH)> = %.4f +- %.4f   acc %.2f   <|dH|> %.3f   %.3f s/traj   P(10) = %.4f" %
          (act, P, eP, E, eE, res[act]["acceptance"], res[act]["mean_abs_dH"], res[act]["seconds_per_trajectory"], res[act]["first10_final_plaquette"]), flush=True)
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
