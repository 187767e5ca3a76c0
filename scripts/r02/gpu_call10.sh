#!/bin/bash
mkdir -p gpurun_out/r02
L=gpurun_out/r02/call10.log; : > $L
timeout 600 python -m pytest tests/test_gpu_solver_edges.py tests/test_gpu_parity.py tests/test_gpu_md.py tests/test_gpu_md_staggered.py -x -q >> $L 2>&1
python - >> $L 2>&1 <<'PY'
import time, sys
sys.path.insert(0,'.')
import latticeqcd_jl_amd as lq
for kind, name in ((lq.STAGGERED, "Staggered"), (lq.WILSON, "Wilson")):
  for small in (0, 1):
    U = lq.Initialize_Gaugefields(3, 0, 8, 8, 8, 8, condition="hot", randomseed=111)
    lat = U.lattice
    lat.set_param("cg_small", small)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": name, "mass": 0.5, "κ": 0.141139, "eps_CG": 1e-10})
    b = lq.Fermionfields(lat, kind); lq.gauss_distribution_fermion_(b, 112); x = b.similar()
    best = 1e9
    for rep in range(7):
        lq.clear_fermion_(x); lat.sync(); t0 = time.perf_counter(); it, rr = lq.solve_DinvX_(x, lq.DdagD_operator(D), b, return_info=True); dt = time.perf_counter() - t0; best = min(best, dt)
    ms = lq.bench_cg(D, x, b, warm=20, niter=400)
    print("8^4 %s CG cg_small=%d iters=%d rr=%.2e solve best %.3f ms; fixed window %.2f us/iteration" % (name, small, it, rr, 1e3 * best, 1e3 * ms))
PY
tail -12 $L
