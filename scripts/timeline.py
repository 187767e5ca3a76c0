#!/usr/bin/env python3
"""Per-iteration timeline from a rocprofv3 --kernel-trace CSV: finds the steady-state period (distance between consecutive
launches of an anchor kernel), then prints every kernel of one period with start offset, duration and the idle gap in
front of it (on the union of all streams).   usage: timeline.py <kernel_trace.csv> [anchor-substring] [period-index]"""
import csv
import sys


def short(name):
    name = name.replace("lqcd::", "")
    return name.split("(")[0][:46]


def main():
    path = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "cg_update_xp"
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(idx) < 4:
        print("anchor not found often enough:", anchor, len(idx))
        return
    periods = [rows[idx[i + 1]][0] - rows[idx[i]][0] for i in range(len(idx) - 1)]
    periods_sorted = sorted(periods)
    print("anchor %s: %d launches, period median %.1f us, min %.1f us" % (anchor, len(idx), periods_sorted[len(periods) // 2] / 1e3,
                                                                        periods_sorted[0] / 1e3))
    a, b = idx[which - 1], idx[which]
    t0 = rows[a + 1][0]
    busy_end = rows[a][1]
    busy = 0
    print("%-46s %6s %9s %9s %8s" % ("kernel", "queue", "start_us", "dur_us", "gap_us"))
    for r in rows[a + 1:b + 1]:
        gap = (r[0] - busy_end) / 1e3
        print("%-46s %6s %9.1f %9.1f %8.1f" % (short(r[2]), r[3][-4:], (r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, gap))
        if r[0] > busy_end:
            busy += r[1] - r[0]
        elif r[1] > busy_end:
            busy += r[1] - busy_end
        busy_end = max(busy_end, r[1])
    span = rows[b][1] - rows[a][1]
    print("period %.1f us, GPU busy (union) %.1f us, idle %.1f us" % (span / 1e3, busy / 1e3, (span - busy) / 1e3))


if __name__ == "__main__":
    main()
