#!/bin/bash
# call U: rocprofv3 evidence for the two new kernels of the round -- kernel trace of the 8^4 one-launch CG, PMC traffic of the fp32 site-pair kernel
cd "$(dirname "$0")/../.."
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/r03_u; rm -rf $O; mkdir -p $O
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/small -o t -- python $R/scripts/r03/small_cg_probe.py > $O/small.log 2>&1)
f=$(find $O/small -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
for pass in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do n=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_$n -o p -- python $R/scripts/r03/xfuse_probe.py mixed_defer_x > $O/pmc_$n.log 2>&1)
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/r03_u/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pair32" in r["Kernel_Name"] or "cg32_update" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"]); acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
with open("gpurun_out/r03_u/pmc_pair32_summary.csv", "w") as o:
    o.write("kernel,counter,mean_per_launch,launches\n")
    for k, v in sorted(acc.items()): o.write('"%s",%s,%.6g,%d\n' % (k[0], k[1], v[0] / v[1], v[1]))
print(open("gpurun_out/r03_u/pmc_pair32_summary.csv").read())
PY
