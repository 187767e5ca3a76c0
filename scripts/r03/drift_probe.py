#!/usr/bin/env python3
"""How long does the 12-real link criterion (every link unitary to 1e-14) survive molecular dynamics?  Pure-gauge leapfrog legs of the
reference's Sexton-Weingarten integrator (runMD_QPQ_sw!, standardMD.jl:146-166: N = 10 gauge legs per MD step, dtau = 0.05) from a hot
start at beta = 5.7; after every MD step: max |row2 - conj(row0 x row1)| over all links and whether the Dslash would still read 12 reals.
usage: drift_probe.py [--lattice 16,16,16,32] [--steps 20] [key=value ...]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import latticeqcd_jl_amd as lq  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lattice", default="16,16,16,32")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--traj", type=int, default=1)
ap.add_argument("sets", nargs="*")
a = ap.parse_args()
L = tuple(int(v) for v in a.lattice.split(","))
lat = lq.Lattice(L)
for kv in a.sets:
    k, v = kv.split("=")
    lat.set_param(k, int(v))
U = lq.Gaugefields(lat)
lq.lib.check(lq.lib.lib().lqcd_gauge_hot_start(U._h, ctypes.c_uint64(111)))
p = lq.Gaugefields(lat)
dtau, nsw, beta = 0.05, 10, 5.7
print("lattice", L, a.sets, "start: dev %.3e plaq %.6f" % (lq.unitarity_deviation(U), lq.calculate_Plaquette(U)), flush=True)
nupd = 0
for traj in range(a.traj):
    lq.gauss_distribution_(p, 1000 + traj)
    for step in range(a.steps):
        for _ in range(nsw):
            eps = dtau / nsw
            lq.U_update_(U, p, 0.5 * eps)
            lq.P_update_(U, p, eps, beta)
            lq.U_update_(U, p, 0.5 * eps)
            nupd += 2
        dev = lq.unitarity_deviation(U)
        print("traj %d step %2d  link updates %4d  max dev %.3e  12-real path %s" % (traj, step + 1, nupd, dev, "on" if dev <= 1e-14 else "OFF"), flush=True)
print("final plaquette %.6f" % lq.calculate_Plaquette(U))
