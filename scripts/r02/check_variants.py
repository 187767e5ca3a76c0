#!/usr/bin/env python3
"""Bitwise comparison of the Wilson stencil variants (1 = direction-split is the reference) and a timing sweep.
usage: check_variants.py [--time 1] [--lattice 32,32,32,64]"""
import argparse, ctypes, os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import latticeqcd_jl_amd as lq

ap = argparse.ArgumentParser()
ap.add_argument("--lattice", default="32,32,32,64")
ap.add_argument("--time", type=int, default=1)
ap.add_argument("--variants", default="4,5")
ap.add_argument("--nts", default="0")
ap.add_argument("--reps", type=int, default=200)
a = ap.parse_args()
L = tuple(int(v) for v in a.lattice.split(","))
lat = lq.Lattice(L)
U = lq.Gaugefields(lat)
lq.lib.check(lq.lib.lib().lqcd_gauge_hot_start(U._h, ctypes.c_uint64(111)))
D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.141139, "boundarycondition": (1, 1, 1, -1)})
b = lq.Fermionfields(lat, lq.WILSON)
lq.gauss_distribution_fermion_(b, 112)
y0, y1, d = b.similar(), b.similar(), b.similar()
V = L[0] * L[1] * L[2] * L[3]
variants = [int(v) for v in a.variants.split(",")]
ok = True
for recon in (18, 12):
    lat.set_param("gauge_recon", recon)
    for dag in (0, 1):
        Dd = D.adjoint() if dag else D
        lat.set_param("dslash_variant", 1)
        lq.mul_(y0, Dd, b)
        n0 = lq.dot(y0, y0).real
        for v in variants:
            lat.set_param("dslash_variant", v)
            lq.mul_(y1, Dd, b)
            lq.substitute_fermion_(d, y1)
            lq.add_fermion_(d, -1.0, y0)
            diff = lq.dot(d, d).real
            print("recon=%d dagger=%d variant=%d |y|^2=%.15e |y_v - y_1|^2=%.3e" % (recon, dag, v, n0, diff), flush=True)
            ok = ok and diff <= 1e-24 * n0
print("VARIANTS_OK" if ok else "VARIANTS_DIFFER", flush=True)
if a.time:
    for recon in (18, 12):
        lat.set_param("gauge_recon", recon)
        for v in [1] + variants:
            for nt in [int(x) for x in a.nts.split(",")]:
                lat.set_param("dslash_variant", v)
                lat.set_param("nt_gauge", nt & 3)
                lat.set_param("nt_store", 1 if nt & 4 else 0)
                ms = lq.bench_dslash(D, y1, b, warm=20, reps=a.reps)
                moved = 960 if recon == 18 else 768
                print("time recon=%d variant=%d nt=%d ms=%.4f frac960=%.3f frac_moved=%.3f" % (recon, v, nt, ms, 960 * V / ms / 1e6 / 8000, moved * V / ms / 1e6 / 8000), flush=True)
