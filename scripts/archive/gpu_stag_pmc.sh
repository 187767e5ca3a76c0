#!/bin/bash
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/stag_pmc; mkdir -p $O; export TMPDIR=/tmp
timeout 120 python scripts/stream_probe.py 2>&1 | tail -4
for L in 48,48,48,96 32,32,32,32; do
for pass in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  n=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && timeout 120 rocprofv3 --pmc $pass --output-format csv -d $O/${L}_$n -o p -- python $R/scripts/dslash_probe.py --kind Staggered --lattice $L --reps 5 --warm 1 > $O/${L}_$n.log 2>&1) || echo "pmc $n failed"
done
done
python - <<PY
import csv,glob
from collections import defaultdict
for L in ("48,48,48,96","32,32,32,32"):
    acc=defaultdict(lambda:[0.0,0])
    for f in glob.glob("gpurun_out/stag_pmc/%s_*/**/*counter_collection.csv"%L,recursive=True):
        for row in csv.DictReader(open(f)):
            if 'staggered' in row['Kernel_Name']:
                acc[row['Counter_Name']][0]+=float(row['Counter_Value']); acc[row['Counter_Name']][1]+=1
    print("==",L)
    for k,v in sorted(acc.items()): print("  %-30s %.6g"%(k,v[0]/v[1]))
PY
find $O -name '*.csv' -size +5M -delete
