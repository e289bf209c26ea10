#!/usr/bin/env python3
"""Folds a rocprofv3 --pmc counter_collection.csv (SQ counters) into one row per kernel: mean
counter value per dispatch, plus VALU wave-instructions per board for the step kernels.

    python tools/pmc_sq_summary.py <dir with *counter_collection.csv> BOARDS > profiles/rNN/pmc_sq_summary.csv
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
    d, boards = sys.argv[1], int(sys.argv[2])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for v in acc.values() for c in v})
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "dispatches"] + counters + ["SQ_INSTS_VALU_per_board"])
    ours = ("step_kernel", "sample_actions_kernel", "observe_kernel", "check_win_kernel", "export_kernel",
            "import_kernel", "node_info_kernel", "expand_kernel", "rollout_kernel", "encode_kernel", "board_op_kernel",
            "step_fused_kernel", "step_random_fused_kernel", "floor")
    for k in sorted(acc):
        if not any(o in k for o in ours):
            continue                                   # torch's own reductions / copies of the recording pass
        v = acc[k]
        n = max(len(x) for x in v.values())
        row = [k, n] + ["%.1f" % (sum(v[c]) / len(v[c])) if c in v else "" for c in counters]
        per = ""
        if ("step_kernel" in k or "step_random_fused_kernel" in k) and "SQ_INSTS_VALU" in v:
            per = "%.1f" % (sum(v["SQ_INSTS_VALU"]) / len(v["SQ_INSTS_VALU"]) * 64.0 / boards)
        w.writerow(row + [per])


if __name__ == "__main__":
    main()
