"""Kernel sequence of a rocprofv3 --kernel-trace CSV between two occurrences of an anchor kernel: name, start (us from the anchor), duration, gap to the previous kernel.
usage: trace_sequence.py kernel_trace.csv anchor-substring [occurrence] [count]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2]
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cnt = int(sys.argv[4]) if len(sys.argv) > 4 else 60
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
i0 = idx[occ]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = None
for r in rows[i0:i0 + cnt]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("lqcd::", "").replace("void ", "")[:70]
    print("%-72s %9.1f %7.1f %7.1f" % (name, (s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev_end is None else (s - prev_end) / 1e3))
    prev_end = e
