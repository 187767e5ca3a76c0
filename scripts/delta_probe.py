#!/usr/bin/env python3
"""Reference-format links (unitary to ~1e-10, not 1e-14): the "12 + delta" link copy of the scalar-addressing Wilson kernel against the 18-real kernel and
against the oracle.  usage: delta_probe.py [--L 32,32,32,64] [--dev 1e-10]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latticeqcd_jl_amd as lq  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (checker only)

args = sys.argv[1:]
L = tuple(int(v) for v in args[args.index("--L") + 1].split(",")) if "--L" in args else (32, 32, 32, 64)
dev = float(args[args.index("--dev") + 1]) if "--dev" in args else 1e-10
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
Uh = U.download()
rng = np.random.default_rng(5)
Uh = Uh + dev * (rng.standard_normal(Uh.shape) + 1j * rng.standard_normal(Uh.shape)) / 3.0      # a text file's 11 significant digits
U.upload(Uh)
print("unitarity deviation of the field: %.3e" % lq.unitarity_deviation(U))
D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.141139})
b = lq.Fermionfields(lat, lq.WILSON)
lq.gauss_distribution_fermion_(b, 112)
y, x = b.similar(), b.similar()
orc.set_threads(os.cpu_count() or 1)
ref = {False: orc.wilson_D(Uh, b.download(), L, 0.141139, 1.0, (1, 1, 1, -1)), True: orc.wilson_D(Uh, b.download(), L, 0.141139, 1.0, (1, 1, 1, -1), dagger=True)}
V = L[0] * L[1] * L[2] * L[3]
for sets in ({"gauge_delta": 0, "dslash_s18": 0}, {"gauge_delta": 0, "dslash_s18": 1}, {"gauge_delta": 0, "dslash_s18": 1, "nt_gauge": 0}, {"gauge_delta": 1}, {"gauge_delta": 1, "nt_gauge": 0}):
    for k, v in sets.items():
        lat.set_param(k, v)
    errs = []
    for dag in (False, True):
        lq.mul_(y, D.adjoint() if dag else D, b)
        errs.append(float(np.abs(y.download() - ref[dag]).max() / np.abs(ref[dag]).max()))
    active = lat.get_param("recon_active")
    ms = lq.bench_dslash(D, y, b, warm=20, reps=300)
    msd = lq.bench_dslash(D.adjoint(), y, b, warm=20, reps=300)
    msi = lq.bench_cg(D, x, b, warm=5, niter=100)
    print("%-34s recon_active %d  D %.4f ms (frac of 8 TB/s by 960 B/site %.3f)  D^+ %.4f ms  CG %.1f iter/s  rel err vs oracle D %.2e D^+ %.2e"
          % (sets, active, ms, 960 * V / ms / 1e6 / 8000, msd, 1e3 / msi, errs[0], errs[1]))
    lat.set_param("nt_gauge", 1)
