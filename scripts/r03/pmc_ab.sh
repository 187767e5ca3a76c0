#!/bin/bash
# PMC passes (separate runs, kernel-trace only) of the Wilson Dslash under a set of tunables: pmc_ab.sh <label> [key=value ...]
# -> gpurun_out/r03/pmc_<label>.csv (mean per launch of every counter, per kernel)
cd "$(dirname "$0")/../.."
R=$(pwd); L=$1; shift; O=$R/gpurun_out/r03/pmc_$L; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
SETS=""; for kv in "$@"; do SETS="$SETS --set $kv"; done
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
            "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/p$i -o p -- python $R/scripts/dslash_probe.py --reps 5 --warm 1 $SETS > $O/p$i.log 2>&1) || echo "pass $i failed: $pass"
done
python - "$O" "$R/gpurun_out/r03/pmc_$L.csv" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "wilson_dirsplit" in row["Kernel_Name"]:
            k = (row["Kernel_Name"].split("(")[0].replace("void ", "").replace("lqcd::", ""), row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
with open(sys.argv[2], "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for k, v in sorted(acc.items()):
        f.write('"%s",%s,%.6g,%d\n' % (k[0], k[1], v[0] / v[1], v[1]))
print(open(sys.argv[2]).read())
PY
