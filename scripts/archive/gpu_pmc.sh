#!/bin/bash
# usage: gpu_pmc.sh <tag> <probe args...> : safe PMC passes (each under its own timeout) on the dslash probe
cd "$(dirname "$0")/../.."
R=$(pwd); TAG=$1; shift; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
n=0
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
            "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
            "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  n=$((n+1))
  (cd /tmp && timeout 90 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/$TAG/p$n -o p -- python $R/scripts/dslash_probe.py --reps 3 --warm 1 "$@" > $R/gpurun_out/$TAG/p$n.log 2>&1) || echo "pass $n [$pass] failed/timeout"
done
python - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(lambda:[0.0,0])
for f in glob.glob("gpurun_out/$TAG/p*/**/*counter_collection.csv",recursive=True):
    for row in csv.DictReader(open(f)):
        if 'wilson' in row['Kernel_Name']:
            acc[row['Counter_Name']][0]+=float(row['Counter_Value']); acc[row['Counter_Name']][1]+=1
print("== $TAG")
for k,v in sorted(acc.items()): print("  %-42s %.6g"%(k,v[0]/v[1]))
PY
