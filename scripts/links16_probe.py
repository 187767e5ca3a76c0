#!/usr/bin/env python3
"""16-bit links in the fp32 site-pair inner operator (tunable mixed_links16): time and error of one application, then the mixed solvers and the fermion force
(calc_UdSfdU!, AbstractMD.jl:129) with and without.  usage: links16_probe.py [x,y,z,t]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latticeqcd_jl_amd as lq
L = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "32,32,32,64").split(","))
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
for kv in os.environ.get("LQCD_SET", "").split():
    k, v = kv.split("=")
    lat.set_param(k, int(v))
D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.141139, "eps_CG": 1e-16, "MaxCGstep": 3000})
b = lq.Fermionfields(lat, lq.WILSON)
lq.gauss_distribution_fermion_(b, 112)
ref = b.similar(); lq.mul_(ref, D, b)
refh = ref.download()
out = b.similar()
for l16 in (0, 2, 0, 2):      # 2: int16 links in every mixed-precision solver; 1 (default): in the even-odd BiCGStab of the action / force solves only
    lat.set_param("mixed_links16", l16)
    for dg in (0, 1):
        Dd = D.adjoint() if dg else D
        ms = lq.mul_f32_(out, Dd, b, reps=50)
        err = np.abs(out.download() - refh).max() / np.abs(refh).max() if dg == 0 else 0.0
        print("links16 %d dagger %d: %.4f ms per application (pair32_active %d)%s" % (l16, dg, ms, lat.get_param("pair32_active"), "  max rel err vs fp64 %.2e" % err if dg == 0 else ""), flush=True)
A = lq.DdagD_operator(D)
x = b.similar()
for l16 in (0, 2):
    lat.set_param("mixed_links16", l16)
    lq.clear_fermion_(x); lq.solve_mixed_DinvX_(x, A, b, return_info=True)
    best = 1e9
    for _ in range(3):
        lq.clear_fermion_(x)
        t0 = time.perf_counter(); info = lq.solve_mixed_DinvX_(x, A, b, return_info=True); best = min(best, time.perf_counter() - t0)
    r = b.similar(); lq.mul_(r, A, x); lq.add_fermion_(r, -1.0, b)
    print("mixed CG links16 %d: %.2f ms (iterations, outer, rr)=%s true_rr=%.3e" % (l16, 1e3 * best, info, lq.dot(r, r).real), flush=True)
fa = lq.FermiAction(D)
eta = b.similar(); X = b.similar()
lq.gauss_distribution_fermion_(X, 5)
lq.sample_pseudofermions_(eta, U, fa, X)
G = lq.Gaugefields(lat)
Gh = {}
for mixed, l16, rel in ((0, 0, 1), (1, 0, 0), (1, 0, 1), (1, 1, 0), (1, 1, 1)):
    lat.set_param("mixed_action_solver", mixed); lat.set_param("mixed_links16", l16); lat.set_param("bicg_reliable", rel)
    lq.calc_UdSfdU_(G, fa, U, eta); lat.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        lq.calc_UdSfdU_(G, fa, U, eta)
    lat.sync()
    dt = (time.perf_counter() - t0) / 5
    Gh[(mixed, l16, rel)] = G.download()
    d = np.abs(Gh[(mixed, l16, rel)] - Gh[(0, 0, 1)]).max() / np.abs(Gh[(0, 0, 1)]).max()
    print("calc_UdSfdU mixed %d links16 %d reliable %d: %.2f ms, max rel diff of the force vs fp64 solve %.2e" % (mixed, l16, rel, 1e3 * dt, d), flush=True)
