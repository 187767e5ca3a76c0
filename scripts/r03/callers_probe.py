#!/usr/bin/env python3
"""Gauge side of the MD step at 32^3x64 through the interface the UNCHANGED reference callers use (per-direction U[mu], p[mu] and temporaries:
AbstractMD.jl:78-118 transliterated in tests/test_gpu_reference_callers.py) against the fused four-direction entry points"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import latticeqcd_jl_amd as lq
import test_gpu_reference_callers as rc
L = (32, 32, 32, 64)
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
ga = lq.GaugeAction(U)
pl = lq.make_loops_fromname("plaquette", Dim=4)
pl = pl + lq.make_loops_fromname("plaquette", Dim=4, adjoint=True)
ga.push_(5.7 / 2, pl)
md = rc.StandardMD(lq, U, ga, True, 0.05, 20)
lq.gauss_distribution_(md.p, 7)
def timed(fn, n=20):
    fn(); lat.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    lat.sync(); return 1e3 * (time.perf_counter() - t0) / n
lat.lazy_links = False
print("per-direction, eager: U_update! %.3f ms   P_update! %.3f ms" % (timed(lambda: rc.U_update_(U, md.p, 0.05, md)), timed(lambda: rc.P_update_(U, md.p, 0.1, md))), flush=True)
lat.lazy_links = True
print("per-direction, lazy:  U_update! %.3f ms   P_update! %.3f ms" % (timed(lambda: rc.U_update_(U, md.p, 0.05, md)), timed(lambda: rc.P_update_(U, md.p, 0.1, md))), flush=True)
print("fused         U_update! %.3f ms   P_update! %.3f ms" % (timed(lambda: lq.U_update_(U, md.p, 0.0025)), timed(lambda: lq.P_update_(U, md.p, 0.005, 5.7))), flush=True)
