#!/usr/bin/env python3
"""A/B on one box: direction-split Wilson kernel (variant 1) against its persistent, software-pipelined form (dslash_pipe = 1).
32^3x64 hot start; per setting: median Dslash time of D and D^+ (200 applications, each between its own HIP events), CG iterations/s
(200-iteration window), and the mixed-precision CG to 1e-16.  usage: pipe_probe.py [--lattice ...] [key=value ...]"""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import latticeqcd_jl_amd as lq  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lattice", default="32,32,32,64")
ap.add_argument("--mixed", type=int, default=1)
ap.add_argument("--cg", type=int, default=200)
ap.add_argument("sets", nargs="*")
a = ap.parse_args()
L = tuple(int(v) for v in a.lattice.split(","))
V = L[0] * L[1] * L[2] * L[3]
lat = lq.Lattice(L)
U = lq.Gaugefields(lat)
lq.lib.check(lq.lib.lib().lqcd_gauge_hot_start(U._h, ctypes.c_uint64(111)))
D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.141139, "eps_CG": 1e-16, "MaxCGstep": 2000})
b = lq.Fermionfields(lat, lq.WILSON)
lq.gauss_distribution_fermion_(b, 112)
y, x = b.similar(), b.similar()
for kv in a.sets:
    k, v = kv.split("=")
    lat.set_param(k, int(v))


def run(label, **params):
    for k, v in params.items():
        lat.set_param(k, v)
    med, mean = lq.bench_dslash_median(D, y, b, warm=20, reps=200)
    medd, meand = lq.bench_dslash_median(D.adjoint(), y, b, warm=20, reps=200)
    msi = lq.bench_cg(D, x, b, warm=5, niter=a.cg) if a.cg else float("nan")
    line = "%-34s D %.4f ms (frac960 %.3f)  D+ %.4f ms  CG %.1f iter/s" % (label, med, 960 * V / med / 1e6 / 8000, medd, 1e3 / msi)
    if a.mixed:
        lq.lib.check(lq.lib.lib().lqcd_spinor_zero(x._h))
        lat.sync(); t0 = time.perf_counter()
        it, outer, rr = lq.solve_mixed_DinvX_(x, lq.DdagD_operator(D), b, return_info=True)
        lat.sync(); tm = 1e3 * (time.perf_counter() - t0)
        lq.lib.check(lq.lib.lib().lqcd_spinor_zero(x._h))
        lat.sync(); t0 = time.perf_counter()
        it64, rr64 = lq.solve_DinvX_(x, lq.DdagD_operator(D), b, return_info=True)
        lat.sync(); t64 = 1e3 * (time.perf_counter() - t0)
        line += "  mixed %.1f ms (%d inner, %d outer)  fp64 %.1f ms (%d it)  x%.2f" % (tm, it, outer, t64, it64, t64 / tm)
    print(line, flush=True)


for recon in (12, 18):
    run("recon%d plain" % recon, gauge_recon=recon, dslash_pipe=0, pipe_per_cu=0)
    for per_cu in (3, 2):
        run("recon%d pipe %d/CU" % (recon, per_cu), gauge_recon=recon, dslash_pipe=1, pipe_per_cu=per_cu)
    run("recon%d scalar addressing" % recon, gauge_recon=recon, dslash_pipe=2, pipe_per_cu=0)
    for n in (2, 4):
        run("recon%d pipelined, %d chunks/WG" % (recon, n), gauge_recon=recon, dslash_pipe=3, pipe_chunks_per_wg=n)
    run("recon%d plain (again)" % recon, gauge_recon=recon, dslash_pipe=0, pipe_per_cu=0)
