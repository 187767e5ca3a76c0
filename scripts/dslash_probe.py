#!/usr/bin/env python3
"""Micro-driver for profiling: repeated Wilson Dslash (and optionally CG iterations) on a hot-start lattice.
usage: dslash_probe.py [--lattice 32,32,32,64] [--reps 20] [--warm 3] [--cg 0] [--set key=value ...]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latticeqcd_jl_amd as lq  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lattice", default="32,32,32,64")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--warm", type=int, default=3)
ap.add_argument("--cg", type=int, default=0)
ap.add_argument("--kind", default="Wilson")
ap.add_argument("--dagger", type=int, default=0)
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--selfcomm", type=int, default=0, help="1: init world-size-1 RCCL communicators, 2: the peer-mapped backend mapped onto itself (use with LQCD_FORCE_PARTITION)")
a = ap.parse_args()
L = tuple(int(v) for v in a.lattice.split(","))
lat = lq.Lattice(L)
if a.selfcomm == 2:
    lat.comm_init_peer()
elif a.selfcomm:
    lat.comm_init(lq.comm_unique_id())
for kv in a.set:
    k, v = kv.split("=")
    lat.set_param(k, int(v))
U = lq.Gaugefields(lat)
lq.lib.check(lq.lib.lib().lqcd_gauge_hot_start(U._h, ctypes.c_uint64(111)))
D = lq.Dirac_operator(U, None, {"Dirac_operator": a.kind, "κ": 0.141139, "mass": 0.5, "Clover_coefficient": 1.0})
kind = lq.STAGGERED if a.kind == "Staggered" else lq.WILSON
b = lq.Fermionfields(lat, kind)
lq.gauss_distribution_fermion_(b, 112)
y = b.similar()
Dd = D.adjoint() if a.dagger else D
ms = lq.bench_dslash(Dd, y, b, warm=a.warm, reps=a.reps)
V = L[0] * L[1] * L[2] * L[3]
bps, fps = (960, 1320) if kind == lq.WILSON else (672, 570)
if a.kind == "WilsonClover":
    bps, fps = 1536, 1824
print("dslash %s L=%s set=%s ms=%.4f GFLOPs=%.0f algGB/s=%.0f frac=%.3f" % (a.kind, L, a.set, ms, fps * V / ms / 1e6, bps * V / ms / 1e6, bps * V / ms / 1e6 / 8000))
if a.cg:
    x = b.similar()
    msi = lq.bench_cg(D, x, b, warm=2, niter=a.cg)
    print("cg ms/iter=%.4f iter/s=%.1f" % (msi, 1e3 / msi))
