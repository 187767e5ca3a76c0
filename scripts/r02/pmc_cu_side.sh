#!/bin/bash
# CU-side counters of the default Wilson Dslash kernel (separate --pmc passes, kernel-trace only): wave residency, instruction-class
# activity, texture-addresser busy / stalls, L1->L2 read latency, LDS conflicts.  Summary -> gpurun_out/r02/pmc_cu_side.csv
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/r02/pmc_cu; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
            "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
            "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
            "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" \
            "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_CYCLES SQ_INSTS_VALU_ADD_F64"; do
  i=$((i+1))
  (cd /tmp && timeout 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/p$i -o p -- python $R/scripts/dslash_probe.py --reps 5 --warm 1 > $O/p$i.log 2>&1) || echo "pass $i failed: $pass"
done
python - <<'PY'
import csv, glob, os
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/r02/pmc_cu/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "wilson_dirsplit" in row["Kernel_Name"]:
            k = (row["Kernel_Name"].split("(")[0].replace("void ", "").replace("lqcd::", ""), row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
with open("gpurun_out/r02/pmc_cu_side.csv", "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for k, v in sorted(acc.items()):
        f.write('"%s",%s,%.6g,%d\n' % (k[0], k[1], v[0] / v[1], v[1]))
print(open("gpurun_out/r02/pmc_cu_side.csv").read())
PY
