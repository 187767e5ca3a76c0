#!/usr/bin/env python3
"""fp64 multi-shift CG vs the mixed-precision form (fp32 multi-shift pass + per-shift fp64 defect correction) on a staggered operator,
10 shifts 1e-4 * 3^k, mass 0.05, hot start: tight target (|r|^2 < 1e-20 |b|^2) and an MD-force target (1e-12 |b|^2)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import latticeqcd_jl_amd as lq
L = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "32,32,32,64").split(","))
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": 0.05, "eps_CG": 1e-16, "MaxCGstep": 20000})
b = lq.Fermionfields(lat, lq.STAGGERED)
lq.gauss_distribution_fermion_(b, 112)
bb = lq.dot(b, b).real
sig = [1e-4 * 3 ** k for k in range(10)]
xs = [b.similar() for _ in sig]
A = lq.DdagD_operator(D)
def timed(fn, reps=2):
    fn(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); best = min(best, time.perf_counter() - t0)
    return best, out
for tag, rel in (("tight", 1e-20), ("force", 1e-12), ("loose", 1e-9)):
    eps = rel * bb
    d64, (it64, rr64) = timed(lambda: lq.shiftedcg(xs, sig, None, A, b, eps=eps, return_info=True))
    dm, (itm, outm, rrm) = timed(lambda: lq.shiftedcg_mixed(xs, sig, None, A, b, eps=eps, return_info=True))
    print(json.dumps({"L": L, "target_rel": rel, "fp64_ms": 1e3 * d64, "fp64_iters": it64, "mixed_ms": 1e3 * dm, "mixed_fp32_iters": itm,
                      "corrections": outm, "worst_true_rr_over_bb": rrm / bb, "speedup": d64 / dm}))
