#!/bin/bash
# PMC passes over the gauge side of the MD step (scripts/r02/md_probe.py): per-kernel means.  pmc_md.sh <label> [key=value ...]
cd "$(dirname "$0")/../.."
R=$(pwd); L=$1; shift; O=$R/gpurun_out/r03/pmcmd_$L; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/p$i -o p -- python $R/scripts/r02/md_probe.py "$@" > $O/p$i.log 2>&1) || echo "pass $i failed"
done
python - "$O" "$R/gpurun_out/r03/pmcmd_$L.csv" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "gauge_force" in n or "link_exp" in n or "momentum_add" in n:
            k = (n.split("(")[0].replace("void ", "").replace("lqcd::", ""), row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
with open(sys.argv[2], "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for k, v in sorted(acc.items()):
        f.write('"%s",%s,%.6g,%d\n' % (k[0], k[1], v[0] / v[1], v[1]))
print(open(sys.argv[2]).read())
PY
