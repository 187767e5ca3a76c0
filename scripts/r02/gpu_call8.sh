#!/bin/bash
mkdir -p gpurun_out/r02
L=gpurun_out/r02/call8.log; : > $L
for recon in 18 12; do
for ntg in 1 5; do
  python scripts/dslash_probe.py --reps 300 --warm 30 --set gauge_recon=$recon --set nt_gauge=$ntg >> $L 2>&1
done
for ns in 8 16 32; do for ys in 1 2 4 8; do
  python scripts/dslash_probe.py --reps 200 --warm 20 --set gauge_recon=$recon --set xcd_nsub=$ns --set xcd_ysplit=$ys >> $L 2>&1
done; done
done
# small lattice: hipGraph replay of the CG bursts
python - >> $L 2>&1 <<'PY'
import time, sys
sys.path.insert(0,'.')
import latticeqcd_jl_amd as lq
for graph in (0, 1):
    U = lq.Initialize_Gaugefields(3, 0, 8, 8, 8, 8, condition="hot", randomseed=111)
    lat = U.lattice
    lat.set_param("graph", graph)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": 0.5, "eps_CG": 1e-10})
    b = lq.Fermionfields(lat, lq.STAGGERED); lq.gauss_distribution_fermion_(b, 112); x = b.similar()
    best = 1e9
    for rep in range(5):
        lq.clear_fermion_(x); lat.sync(); t0 = time.perf_counter(); it, rr = lq.solve_DinvX_(x, lq.DdagD_operator(D), b, return_info=True); dt = time.perf_counter() - t0; best = min(best, dt)
    print("8^4 staggered CG graph=%d iters=%d rr=%.2e best %.3f ms" % (graph, it, rr, 1e3 * best))
PY
grep "^dslash\|8^4" $L
