#!/bin/bash
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/part_trace; mkdir -p $O; export TMPDIR=/tmp
export LQCD_FORCE_PARTITION=14
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $R/scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 20 --warm 5 > $O/log.txt 2>&1)
python - <<'PY'
import csv,glob
rows=list(csv.DictReader(open(glob.glob("gpurun_out/part_trace/**/t_kernel_trace.csv",recursive=True)[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take the last 3 dslash applications: find last indices of wilson_pack
idx=[i for i,r in enumerate(rows) if 'wilson_pack' in r['Kernel_Name']]
s=idx[-3]
t0=int(rows[s]['Start_Timestamp'])
for r in rows[s:]:
    st=(int(r['Start_Timestamp'])-t0)/1000; en=(int(r['End_Timestamp'])-t0)/1000
    print("%8.1f %8.1f  %6.1f us  q=%s  %s"%(st,en,en-st,r.get('Queue_Id','?'),r['Kernel_Name'][:70]))
PY
