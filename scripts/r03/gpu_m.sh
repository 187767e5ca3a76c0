#!/bin/bash
# round 3, GPU call M: fp32 site-pair kernel -- parity, fp32 Dslash time, mixed CG A/B
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_m; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pair32.py tests/test_gpu_mixed.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 25 $O/pytest.log
python - <<'PY' 2>&1 | tee $O/f32_dslash.log
import ctypes, sys
sys.path.insert(0, ".")
import latticeqcd_jl_amd as lq
for L in ((32, 32, 32, 64), (48, 48, 48, 96)):
    lat = lq.Lattice(L)
    U = lq.Gaugefields(lat); lq.lib.check(lq.lib.lib().lqcd_gauge_hot_start(U._h, ctypes.c_uint64(111)))
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.141139})
    b = lq.Fermionfields(lat, lq.WILSON); lq.gauss_distribution_fermion_(b, 112); y = b.similar()
    V = L[0] * L[1] * L[2] * L[3]
    for pair in (1, 0, 1, 0):
        lat.set_param("mixed_pair32", pair)
        ms = lq.mul_f32_(y, D, b, reps=200); msd = lq.mul_f32_(y, D.adjoint(), b, reps=200)
        print("L=%s mixed_pair32=%d fp32 D %.4f ms  D+ %.4f ms   moved 384 B/site -> %.2f TB/s (%.3f of 8 TB/s)" % (L, pair, ms, msd, 384 * V / ms / 1e9, 384 * V / ms / 1e9 / 8), flush=True)
    del y, b, D, U, lat
PY
for rep in 1 2; do
  timeout 300 python scripts/r03/mixed_ab.py mixed_pair32=1
  timeout 300 python scripts/r03/mixed_ab.py mixed_pair32=0
done 2>&1 | tee $O/mixed_ab.log
