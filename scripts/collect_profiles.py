#!/usr/bin/env python3
"""Condense gpurun_out/profile_r01 (rocprofv3 csv output) into the tracked files under profiles/."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/profile_r01"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
dst = "profiles"
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copyfile(f, os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
for name in ("bench_n1.json", "trace_bench.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copyfile(p, os.path.join(dst, f"{tag}_{name}"))
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[(row["Kernel_Name"], row["Counter_Name"])][0] += float(row["Counter_Value"])
        acc[(row["Kernel_Name"], row["Counter_Name"])][1] += 1
rows = [(k[0], k[1], v[0] / v[1], v[1]) for k, v in sorted(acc.items())]
with open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for r in rows:
        f.write('"%s",%s,%.6g,%d\n' % r)
m = {r[1]: r[2] for r in rows if "wilson_hopsplit<false" in r[0] or "wilson_interior<" in r[0] or "wilson_dirsplit<false" in r[0]}
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    # MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports 1/2 of the bytes of a 16-B/lane coalesced read -> doubled;
    # both counters are in KiB.
    traffic = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
    out = {"wilson_dslash_bytes_per_launch_32x32x32x64": traffic, "FETCH_SIZE_KiB": m["FETCH_SIZE"], "WRITE_SIZE_KiB": m["WRITE_SIZE"],
           "note": "HBM/fabric bytes per Wilson Dslash launch = (2*FETCH_SIZE + WRITE_SIZE) KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md",
           "algorithmic_bytes": 960 * 32 * 32 * 32 * 64,
           "kernel": [r[0] for r in rows if "wilson_dirsplit<false" in r[0] or "wilson_hopsplit<false" in r[0] or "wilson_interior<" in r[0]][0],
           "compulsory_bytes_moved": "768 B/site when the kernel is wilson_dirsplit<false, true, ...> (12-real links, third row rebuilt), 960 otherwise"}
    for k2 in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"):
        if k2 in m:
            out[k2] = m[k2]
    json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out))
print(open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv")).read()[:1500])
