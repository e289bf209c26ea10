#!/usr/bin/env python3
"""Per (kernel, grid size) average dispatch duration from a rocprofv3 --kernel-trace CSV: for runs like
tools/bench_rows.py that launch one kernel at several batch sizes, where the --stats table would mix them.

    python tools/trace_rows_summary.py <dir with *_kernel_trace.csv> > profiles/rNN/kernel_trace_rows.csv
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r"(\w+)(<[^>]*>)?\(", name)
    base = m.group(1) if m else name[:40]
    m2 = re.search(r"%s<[^>]*>" % re.escape(base), name)
    return m2.group(0) if m2 else base


def main():
    f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "at::native" in r["Kernel_Name"] or "elementwise" in k or "reduce" in k:
            continue                                         # torch's own kernels of the set-up passes
        grid = r.get("Grid_Size") or r.get("Grid_Size_X")
        wg = r.get("Workgroup_Size") or r.get("Workgroup_Size_X")
        acc[(k, int(grid), int(wg))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "grid_size_lanes", "workgroup_size", "dispatches", "avg_ns", "median_ns", "min_ns", "max_ns"])
    for (k, grid, wg), d in sorted(acc.items()):
        d.sort()
        w.writerow([k, grid, wg, len(d), "%.1f" % (sum(d) / len(d)), d[len(d) // 2], d[0], d[-1]])


if __name__ == "__main__":
    main()
