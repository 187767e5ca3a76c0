#!/bin/bash
# A/B on one box: r prefetch in the update-mode Dslash (A = default build, B = -DLQCD_UPD_PREFETCH=0)
cd "$(dirname "$0")/../.."
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/r02/ab26; rm -rf $O; mkdir -p $O
for v in A B A B; do
  if [ $v = B ]; then export LQCD_HIP_LIB=$R/latticeqcd.jl_amd/csrc/liblqcd_hip_b.so; else unset LQCD_HIP_LIB; fi
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['dslash_ms'], d.get('gauge_recon18_all_reals_read',{}).get('cg_iters_per_s'))"
done
for v in A B; do
  if [ $v = B ]; then export LQCD_HIP_LIB=$R/latticeqcd.jl_amd/csrc/liblqcd_hip_b.so; else unset LQCD_HIP_LIB; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o t -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pmc > /dev/null 2>&1)
  f=$(find $O/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "wilson_dirsplit" in r["Name"] or "cg_update" in r["Name"]:
        print("%-70s calls %5s avg %8.1f us" % (r["Name"].replace("lqcd::","")[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
python scripts/mixed_probe.py 32,32,32,64 Wilson 1e-16 2>&1 | tail -1
timeout 300 python -m pytest tests/test_gpu_mixed.py tests/test_gpu_solver_edges.py -x -q 2>&1 | tail -3
