#!/usr/bin/env python3
"""Gauge side of the MD step at 32^3x64 through the interface the UNCHANGED reference callers use (per-direction U[mu], p[mu] and temporaries:
AbstractMD.jl:78-118, replayed from their call trace by tests/ref_trace.py) against the fused four-direction entry points"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import latticeqcd_jl_amd as lq
from ref_trace import Replay, standard_md
from test_gpu_reference_callers import plaquette_action
L = (32, 32, 32, 64)
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
md = standard_md(lq, U, plaquette_action(lq, U, 5.7), 0.05, 20)
rp = Replay(lq)
lq.gauss_distribution_(md.p, 7)
def timed(fn, n=20):
    fn(); lat.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    lat.sync(); return 1e3 * (time.perf_counter() - t0) / n
lat.lazy_links = False
print("per-direction, eager: U_update! %.3f ms   P_update! %.3f ms" % (timed(lambda: rp.call("U_update!", U, md.p, 0.05, md)), timed(lambda: rp.call("P_update!", U, md.p, 0.1, md))), flush=True)
lat.lazy_links = True
print("per-direction, lazy:  U_update! %.3f ms   P_update! %.3f ms" % (timed(lambda: rp.call("U_update!", U, md.p, 0.05, md)), timed(lambda: rp.call("P_update!", U, md.p, 0.1, md))), flush=True)
print("fused         U_update! %.3f ms   P_update! %.3f ms" % (timed(lambda: lq.U_update_(U, md.p, 0.0025)), timed(lambda: lq.P_update_(U, md.p, 0.005, 5.7))), flush=True)
