#!/usr/bin/env python3
"""Condense gpurun_out/profile_<tag> (scripts/gpu_profile_round.sh) into the tracked files under profiles/ and regenerate the
auto-generated table of profiles/README.md from them, so that every number quoted there is read from a committed file.
usage: collect_profiles.py [tag]"""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join("gpurun_out", "profile_" + tag)
dst = "profiles"
os.makedirs(dst, exist_ok=True)
V = 32 * 32 * 32 * 64

for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copyfile(f, os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
for name in ("bench_n1.json", "trace_bench.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copyfile(p, os.path.join(dst, f"{tag}_{name}"))

# per-kernel MEDIAN duration from the kernel trace of the profiled bench run.  The --stats average counts the no-op launches a burst enqueues
# behind the converging iteration (10 us each) and is therefore biased low; launches shorter than a fifth of the kernel's median are
# dropped here and both the median and the mean of the rest are reported.
med_rows = {}
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
    dur = defaultdict(list)
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    with open(os.path.join(dst, f"{tag}_bench_kernel_medians.csv"), "w") as out:
        out.write("kernel,launches,noop_launches_dropped,median_us,mean_us_without_noops,min_us,max_us\n")
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            v.sort()
            med_all = v[len(v) // 2]
            real = [x for x in v if x >= 0.2 * med_all]
            med = real[len(real) // 2]
            out.write('"%s",%d,%d,%.2f,%.2f,%.2f,%.2f\n' % (k, len(v), len(v) - len(real), med, sum(real) / len(real), real[0], real[-1]))
            med_rows[k] = (len(v), len(v) - len(real), med, sum(real) / len(real))

acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "wilson" in row["Kernel_Name"]:
            k = (row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
rows = [(k[0], k[1], v[0] / v[1], v[1]) for k, v in sorted(acc.items())]
if rows:
    with open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w") as f:
        f.write("kernel,counter,mean_per_launch,launches\n")
        for r in rows:
            f.write('"%s",%s,%.6g,%d\n' % r)

# ---- the generated table
lines = []
def stat_row(kernel_substr):
    p = os.path.join(dst, f"{tag}_bench_kernel_stats.csv")
    if not os.path.exists(p):
        return None
    for row in csv.DictReader(open(p)):
        if kernel_substr in row["Name"]:
            return row
    return None

def pm(kernel_substr, counter):
    for r in rows:
        if kernel_substr in r[0] and r[1] == counter:
            return r[2]
    return None

bench = trace = None
try:
    bench = json.loads(open(os.path.join(dst, f"{tag}_bench_n1.json")).read().strip().splitlines()[-1])
except Exception:
    pass
try:
    trace = json.loads(open(os.path.join(dst, f"{tag}_trace_bench.json")).read().strip().splitlines()[-1])
except Exception:
    pass
lines.append(f"<!-- BEGIN GENERATED {tag} (scripts/collect_profiles.py {tag}) -->")
lines.append(f"### {tag}: 32³×64 Wilson fp64 on one MI355X — every number below is read from the named file by `scripts/collect_profiles.py`")
lines.append("")
lines.append("| quantity | value | file |")
lines.append("|---|---|---|")
if bench:
    r = bench["roofline"]; g = bench.get("gauge_recon18_all_reals_read", {})
    lines.append(f"| CG iterations/s (`value`), ms per iteration | {bench['value']:.1f}, {bench['ms_per_step']:.4f} | `{tag}_bench_n1.json` |")
    lines.append(f"| default Dslash kernel `{r['kernel']}`: HIP-event mean / per-launch median | {bench['dslash_ms']:.4f} / {bench['dslash_ms_median_per_launch_events']:.4f} ms | `{tag}_bench_n1.json` |")
    lines.append(f"| `roofline.frac` (960 B/site algorithmic ÷ time ÷ 8 TB/s) / by the {r['compulsory_bytes_moved_per_site']} B/site it moves | {r['frac']:.3f} / {r['frac_by_bytes_moved']:.3f} | `{tag}_bench_n1.json` |")
    if r.get("traffic"):
        lines.append(f"| `roofline.traffic` measured in that run (live `--pmc` child passes) | {r['traffic'] / 1e9:.3f} GB per launch = {r['traffic_over_bytes_moved']:.3f} × the bytes moved | `{tag}_bench_n1.json` |")
    if g and "dslash_ms" in g:
        lines.append(f"| all 18 reals of every link read (`gauge_recon = 18`): Dslash, fraction of 8 TB/s on 960 B/site, CG | {g['dslash_ms']:.4f} ms, {g['frac_of_peak']:.3f}, {g['cg_iters_per_s']:.1f} iter/s | `{tag}_bench_n1.json` |")
        if g.get("traffic"):
            lines.append(f"| … its measured traffic | {g['traffic'] / 1e9:.3f} GB per launch = {g['traffic_over_bytes_moved']:.3f} × 960 B/site | `{tag}_bench_n1.json` |")
    t = bench.get("time_to_solution_1e-16")
    if t and "speedup" in t:
        lines.append(f"| time to r·r < 1e-16: fp64 CG / mixed-precision CG (true residual {t['mixed_true_rr']:.1e}) | {t['fp64_cg_ms']:.1f} ms ({t['fp64_iters']} it) / {t['mixed_cg_ms']:.1f} ms ({t['mixed_inner_iters']} inner, {t['mixed_outer_steps']} outer) = {t['speedup']:.2f} × | `{tag}_bench_n1.json` |")
    m = bench.get("after_md_trajectory")
    if m and "cg_iters_per_s" in m:
        lines.append(f"| after {m['link_updates']} link updates of MD (`md_reunitarize = {m['md_reunitarize']}`): max unitarity deviation, 12-real kernel active, Dslash, CG | {m['max_unitarity_deviation']:.1e}, {m['gauge_recon_active']}, {m['dslash_ms']:.4f} ms, {m['cg_iters_per_s']:.1f} iter/s | `{tag}_bench_n1.json` |")
    c = bench.get("cpu_baseline")
    if c:
        lines.append(f"| CPU baseline ({c['kind']}, {c['cores']} core) | {c['value']:.3f} iter/s, Dslash {c['dslash_gflops']:.2f} GFLOP/s | `{tag}_bench_n1.json` |")
        a = c.get("all_cores")
        if a:
            lines.append(f"| … the same oracle window on all {a['cores']} host cores (OpenMP; first-touch by one thread, NUMA-limited) | {a['value']:.3f} iter/s, Dslash {a['dslash_gflops']:.2f} GFLOP/s | `{tag}_bench_n1.json` |")
for sub, label in (("wilson_dirsplit_s<false, true", "12-real D, scalar-addressing instance (default since round 3)"), ("wilson_dirsplit_s<true, true", "12-real D† (CG update mode), scalar-addressing instance"),
                   ("wilson_dirsplit_pair32<false", "fp32 site-pair D (inner solver of the mixed-precision CG)"), ("wilson_dirsplit_pair32<true", "fp32 site-pair D† (update mode)"),
                   ("cg32_update_xp", "fp32 x, p update"),
                   ("wilson_dirsplit<false, true, false>", "12-real D"), ("wilson_dirsplit<true, true, false>", "12-real D† (CG update mode)"),
                   ("wilson_dirsplit<false, false, false>", "18-real D"), ("wilson_dirsplit_pipe<false, true", "12-real D, persistent form"), ("cg_update_even", "p update, even iterations (x deferred)"),
                   ("cg_update_odd", "x (two terms) and p update, odd iterations"), ("cg_update_xp", "x, p update"), ("reduce_final", "final reduction")):
    mk = [k for k in med_rows if sub in k]
    if mk:
        n, dropped, med, mean = med_rows[mk[0]]
        lines.append(f"| rocprofv3 `--kernel-trace` of the same command: {label} `{sub}` | median {med:.1f} µs, mean {mean:.1f} µs over {n - dropped} launches ({dropped} no-op launches of converged bursts dropped) | `{tag}_bench_kernel_medians.csv` |")
        continue
    s = stat_row(sub)
    if s:
        lines.append(f"| rocprofv3 `--kernel-trace --stats` of the same command: {label} `{sub}` | {float(s['AverageNs']) / 1e3:.1f} µs average over {s['Calls']} calls | `{tag}_bench_kernel_stats.csv` |")
if trace:
    lines.append(f"| `bench.py`'s HIP-event Dslash figure in that profiled run | {trace['dslash_ms']:.4f} ms | `{tag}_trace_bench.json` |")
for sub, label in (("wilson_dirsplit_s<false, true", "12-real kernel (scalar addressing)"), ("wilson_dirsplit<false, true, false>", "12-real kernel"), ("wilson_dirsplit<false, false, false>", "18-real kernel")):
    fs, ws = pm(sub, "FETCH_SIZE"), pm(sub, "WRITE_SIZE")
    if fs and ws:
        tr = (2 * fs + ws) * 1024
        moved = (768 if "<false, true" in sub else 960) * V
        hit, miss = pm(sub, "TCC_HIT_sum"), pm(sub, "TCC_MISS_sum")
        lines.append(f"| PMC passes (separate): {label}: FETCH_SIZE, WRITE_SIZE → (2·FETCH + WRITE) KiB | {fs:.4g} KiB, {ws:.4g} KiB → {tr / 1e9:.3f} GB = {tr / moved:.3f} × bytes moved"
                     + (f"; TCC hit {hit / (hit + miss):.3f}" if hit and miss else "") + f" | `{tag}_pmc_summary.csv` |")
lines.append(f"<!-- END GENERATED {tag} -->")
block = "\n".join(lines) + "\n"
readme = os.path.join(dst, "README.md")
text = open(readme).read() if os.path.exists(readme) else ""
pat = re.compile(r"<!-- BEGIN GENERATED %s .*?<!-- END GENERATED %s -->\n" % (tag, tag), re.S)
text = pat.sub(block, text) if pat.search(text) else block + "\n" + text
open(readme, "w").write(text)
print(block)
