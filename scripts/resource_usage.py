#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr saved to a file).
usage: resource_usage.py remarks.txt... [--filter name] [--allow name]...
Exit status 1 if a kernel uses scratch memory (a spill, or a dynamically indexed private array) and its name matches no --allow substring.
build.sh leaves the remarks of every translation unit in latticeqcd.jl_amd/csrc/build/liblqcd_hip/*.remarks."""
import re
import subprocess
import sys

args = sys.argv[1:]
files, flt, allow = [], "", []
while args:
    a = args.pop(0)
    if a == "--filter":
        flt = args.pop(0)
    elif a == "--allow":
        allow.append(args.pop(0))
    else:
        files.append(a)
rows, cur = [], None
for fn in files:
    for line in open(fn, errors="replace").read().splitlines():
        m = re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            continue
        body = m.group(1)
        if body.startswith("Function Name:"):
            cur = {"name": body.split(":", 1)[1].strip(), "file": fn}
            rows.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            cur[k.strip()] = v.strip()
names = [r["name"] for r in rows]
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True).stdout.splitlines()
    if len(dem) != len(names):
        dem = names
except Exception:
    dem = names
bad = []
for r, d in zip(rows, dem):
    d = re.sub(r"\(lqcd::.*", "", d).replace("void ", "").replace("lqcd::", "")
    if flt and flt not in d:
        continue
    scratch = r.get("ScratchSize [bytes/lane]", "0")
    if scratch not in ("0", "?") and not any(a in d for a in allow):
        bad.append((d, scratch))
    print(f"{d:90s} VGPR {r.get('VGPRs','?'):>4s} spill {r.get('VGPRs Spill','?'):>3s} SGPRspill {r.get('SGPRs Spill','?'):>3s} scratch {scratch:>4s} occ {r.get('Occupancy [waves/SIMD]','?')} LDS {r.get('LDS Size [bytes/block]','?')}")
if bad:
    print("\nkernels that use scratch memory:", file=sys.stderr)
    for d, sc in bad:
        print(f"  {d}: {sc} B/lane", file=sys.stderr)
    sys.exit(1)
