#!/bin/bash
# kernel trace of one mixed-precision solve (32^3x64 Wilson, 1e-16): where the time outside the fp32 iterations goes
cd "$(dirname "$0")/.."
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/mixtrace; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o m -- python $R/scripts/mixed_probe.py 32,32,32,64 Wilson 1e-16 > $O/out.txt 2>&1)
tail -2 $O/out.txt
python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join("gpurun_out/mixtrace", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("lqcd::", "").split("(")[0][:50]) for r in csv.DictReader(open(f)))
# last mixed solve = from the last cvt_to_f32 of the gauge-less kind... take the window of the last 1400 kernels and print aggregate busy/idle
idx = [i for i, r in enumerate(rows) if "residual_kernel" in r[2]]
# a solve has outer+1 residual kernels; take the last 4 residual kernels as one solve window
lo, hi = idx[-4], idx[-1]
t0, t1 = rows[lo][0], rows[hi][1]
busy = sum(r[1] - r[0] for r in rows[lo:hi + 1])
print("window %.2f ms, kernels busy %.2f ms, idle %.2f ms, %d kernels" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, hi - lo + 1))
agg = {}
for r in rows[lo:hi + 1]:
    a = agg.setdefault(r[2], [0, 0]); a[0] += r[1] - r[0]; a[1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print("%-52s %8.2f ms %5d calls" % (k, v[0] / 1e6, v[1]))
gaps = sorted(((rows[i + 1][0] - rows[i][1]) / 1e3, rows[i][2], rows[i + 1][2]) for i in range(lo, hi))[-8:]
for g in gaps: print("gap %.1f us after %s before %s" % g)
PY
