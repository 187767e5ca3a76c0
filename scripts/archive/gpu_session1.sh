#!/bin/bash
# First GPU session: parity tests, bench, kernel-trace profile, tunable sweep.  Everything lands in gpurun_out/.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --steps 50 --warmup 5 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench.log
echo "== sweep"
for blk in 64 128 256; do for remap in 0 1; do
  timeout 300 python bench.py --steps 20 --warmup 3 --dslash-reps 100 --no-cpu-baseline --set dslash_block=$blk --set xcd_remap=$remap 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
  d=json.loads(l); print('blk=$blk remap=$remap dslash_ms=%.4f GF=%.0f frac=%.3f cg_it/s=%.1f'%(d['dslash_ms'],d['dslash_gflops'],d['roofline']['frac'],d['value']))
except Exception as e: print('blk=$blk remap=$remap FAILED',l[-300:])
"
done; done | tee gpurun_out/sweep.log
echo "== rocprof kernel trace"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof | head -20
