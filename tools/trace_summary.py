#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace CSV of `bench.py`: the --stats table averages every
dispatch of the step kernel (recording pass + warm-up + timed region); this prints the same
kernel's average over the LAST K dispatches (= bench.py's timed region), which is the number
`roofline.launch_us` must agree with.

    python tools/trace_summary.py <dir with *_kernel_trace.csv> K > profiles/rNN/<name>.txt
"""
import csv
import glob
import os
import sys


def main():
    d, K = sys.argv[1], int(sys.argv[2])
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"]]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    import re; name = re.search(r"step_kernel<[^>]*>", rows[0]["Kernel_Name"]).group(0)
    print("kernel: %s" % name)
    print("VGPR_Count=%s SGPR_Count=%s Workgroup_Size=%s Grid_Size=%s" % (
        rows[-1].get("VGPR_Count"), rows[-1].get("SGPR_Count"), rows[-1].get("Workgroup_Size"),
        rows[-1].get("Grid_Size")))
    print("all %d dispatches : avg %.1f ns  min %d  max %d" % (len(dur), sum(dur) / len(dur), min(dur), max(dur)))
    t = dur[-K:]
    print("last %d (timed)   : avg %.1f ns  min %d  max %d" % (K, sum(t) / len(t), min(t), max(t)))
    st = [int(r["Start_Timestamp"]) for r in rows][-K:]
    en = [int(r["End_Timestamp"]) for r in rows][-K:]
    print("timed span / K     : %.1f ns" % ((en[-1] - st[0]) / K))


if __name__ == "__main__":
    main()
