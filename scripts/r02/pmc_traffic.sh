#!/bin/bash
# usage: pmc_traffic.sh <tag> <dslash_probe args...>   HBM/fabric traffic of the Wilson kernel: separate --pmc passes (FETCH_SIZE, WRITE_SIZE,
# TCC hit/miss), kernel-trace only, each under its own timeout.  Prints 2*FETCH_SIZE + WRITE_SIZE (KiB -> GB) per launch.
cd "$(dirname "$0")/../.."
R=$(pwd); TAG=$1; shift; mkdir -p gpurun_out/r02/$TAG; export TMPDIR=/tmp
n=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  n=$((n+1))
  (cd /tmp && timeout 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/r02/$TAG/p$n -o p -- python $R/scripts/dslash_probe.py --reps 5 --warm 2 "$@" > $R/gpurun_out/r02/$TAG/p$n.log 2>&1) || echo "pass $n [$pass] failed/timeout"
done
python - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(lambda:[0.0,0])
for f in glob.glob("gpurun_out/r02/$TAG/p*/**/*counter_collection.csv",recursive=True):
    for row in csv.DictReader(open(f)):
        if 'wilson' in row['Kernel_Name'] or 'staggered' in row['Kernel_Name']:
            acc[(row['Kernel_Name'].split('(')[0],row['Counter_Name'])][0]+=float(row['Counter_Value']); acc[(row['Kernel_Name'].split('(')[0],row['Counter_Name'])][1]+=1
m={k:v[0]/v[1] for k,v in acc.items()}
kern=sorted(set(k[0] for k in m))
for kn in kern:
    g=lambda c: m.get((kn,c),float('nan'))
    tr=(2*g('FETCH_SIZE')+g('WRITE_SIZE'))*1024/1e9
    print("PMC $TAG | %s | read %.3f GB write %.3f GB traffic %.3f GB | TCC hit %.3f | args: $*" % (kn, 2*g('FETCH_SIZE')*1024/1e9, g('WRITE_SIZE')*1024/1e9, tr, g('TCC_HIT_sum')/(g('TCC_HIT_sum')+g('TCC_MISS_sum'))))
PY
