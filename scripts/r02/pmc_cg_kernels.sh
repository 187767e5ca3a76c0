#!/bin/bash
# fabric traffic (PMC, separate passes) of every kernel of a CG iteration: D, D+ in update mode, the two update kernels
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/r02/pmc_cg; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
n=0
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  n=$((n+1))
  (cd /tmp && timeout 150 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/p$n -o p -- python $R/scripts/dslash_probe.py --reps 2 --warm 1 --cg 12 > $O/p$n.log 2>&1) || echo "pass $n failed"
done
python - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0]); dur = defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/r02/pmc_cg/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0].replace("void ", "").replace("lqcd::", ""), row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
V = 32 * 32 * 32 * 64
comp = {"p64::wilson_dirsplit<false, true, false>": 768, "p64::wilson_dirsplit<true, true, false>": 960, "cg_update_even<true, false>": 3 * 192, "cg_update_odd<true, false>": 6 * 192}
print("kernel | FETCH KiB | WRITE KiB | traffic GB = (2 FETCH + WRITE) KiB | compulsory GB | ratio")
for kn in sorted(set(k[0] for k in acc)):
    f, w = acc.get((kn, "FETCH_SIZE")), acc.get((kn, "WRITE_SIZE"))
    if not f or not w or kn not in comp: continue
    fs, ws = f[0] / f[1], w[0] / w[1]
    tr = (2 * fs + ws) * 1024 / 1e9
    c = comp[kn] * V / 1e9
    print("%s | %.4g | %.4g | %.3f | %.3f | %.3f" % (kn, fs, ws, tr, c, tr / c))
PY
