#!/usr/bin/env python3
"""One MD step of the 2-flavour Wilson HMC at 32^3x64 (Sexton-Weingarten N = 10, test_wilson.toml's parameters) THROUGH THE REFERENCE'S OWN CALLERS
(runMD_QPQ_sw!, U_update!, P_update!, P_update_fermion!: standardMD.jl / AbstractMD.jl, replayed from their call trace by
tests/ref_trace.py) on the binding's per-direction interface: lazy evaluation of the link triples on / off, fp64 / mixed solver"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import latticeqcd_jl_amd as lq
from ref_trace import Replay, standard_md
from test_gpu_reference_callers import plaquette_action, wilson_action
L = (32, 32, 32, 64)
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
md = standard_md(lq, U, plaquette_action(lq, U, 5.7), 0.05, 1, fermi_action=wilson_action(lq, U, 0.141139, 1e-16), SextonWeingargten=True, Nsw=10)
rp = Replay(lq)
rp.call("initialize_MD!", U, md)
for lazy in (True, False):
    for mixed in (0, 1):
        lat.lazy_links = lazy
        lat.set_param("mixed_action_solver", mixed)
        rp.call("runMD!", U, md); lat.sync()
        t0 = time.perf_counter(); rp.call("runMD!", U, md); lat.sync(); t = 1e3 * (time.perf_counter() - t0)
        print("lazy_links %s, mixed_action_solver %d: one MD step through the reference's callers %.1f ms (12-real kernel active: %d)" % (lazy, mixed, t, lat.get_param("recon_active")), flush=True)
