#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace CSV of `bench.py`: the --stats table averages every
dispatch of the step kernel (recording pass + pilot region + R x (warm-up + timed region)); this
prints the same kernel's average over the TIMED dispatches only, which is the number bench.py's
`roofline.launch_us` / `ms_per_step` must agree with.

    python tools/trace_summary.py <dir with *_kernel_trace.csv> K W R [kernel-substring] > profiles/rNN/<name>.txt

K, W, R = bench.py's --steps, --warmup and the `regions` field of its JSON line: the last
R * (W + K) dispatches of the kernel are the R regions, the last K of each are timed.

Gap table.  For the timed dispatches it also prints what the queue looked like under the profiler:
duration of dispatch k, the idle gap end[k] -> begin[k+1], and the period begin[k] -> begin[k+1]
(= what an event pair around K back-to-back launches divides by K), averaged over all regions and listed
dispatch by dispatch for the median region.  UNPROFILED_US (optional 6th argument) = the un-profiled
bench.py line's us per launch, printed beside the profiled period.
"""
import csv
import glob
import os
import re
import sys


def main():
    d, K, W, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    pat = sys.argv[5] if len(sys.argv) > 5 else "step_kernel"
    unprofiled = float(sys.argv[6]) if len(sys.argv) > 6 else None
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    name = re.search(r"%s<[^>]*>" % re.escape(pat), rows[-1]["Kernel_Name"])
    print("kernel: %s" % (name.group(0) if name else rows[-1]["Kernel_Name"][:80]))
    print("VGPR_Count=%s SGPR_Count=%s LDS_Block_Size=%s Workgroup_Size=%s Grid_Size=%s" % (
        rows[-1].get("VGPR_Count"), rows[-1].get("SGPR_Count"), rows[-1].get("LDS_Block_Size"),
        rows[-1].get("Workgroup_Size") or rows[-1].get("Workgroup_Size_X"),
        rows[-1].get("Grid_Size") or rows[-1].get("Grid_Size_X")))
    print("all %d dispatches          : avg %.1f ns  min %d  max %d" % (len(dur), sum(dur) / len(dur), min(dur), max(dur)))
    tail = rows[-R * (W + K):]
    timed, spans = [], []
    for r in range(R):
        reg = tail[r * (W + K) + W:(r + 1) * (W + K)]
        timed += [int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in reg]
        spans.append((int(reg[-1]["End_Timestamp"]) - int(reg[0]["Start_Timestamp"])) / K)
    regs = []
    for r in range(R):
        reg = tail[r * (W + K) + W:(r + 1) * (W + K)]
        b = [int(x["Start_Timestamp"]) for x in reg]
        e = [int(x["End_Timestamp"]) for x in reg]
        regs.append((b, e))
    order = sorted(range(R), key=lambda r: spans[r])
    spans.sort()
    print("timed: %d regions x %d      : avg %.1f ns  min %d  max %d" % (R, K, sum(timed) / len(timed), min(timed), max(timed)))
    print("timed span / K per region  : median %.1f ns  min %.1f  max %.1f" % (spans[len(spans) // 2], spans[0], spans[-1]))
    gaps = [b[k + 1] - e[k] for b, e in regs for k in range(K - 1)]
    periods = [b[k + 1] - b[k] for b, e in regs for k in range(K - 1)]
    gaps.sort()
    print()
    print("queue under the profiler, timed dispatches of all %d regions:" % R)
    print("  duration      end[k]-begin[k]     : avg %8.1f ns" % (sum(timed) / len(timed)))
    print("  idle gap      begin[k+1]-end[k]   : avg %8.1f ns  median %d  p10 %d  p90 %d  (negative = the next dispatch "
          "started before this one ended)" % (sum(gaps) / len(gaps), gaps[len(gaps) // 2], gaps[len(gaps) // 10], gaps[len(gaps) * 9 // 10]))
    print("  period        begin[k+1]-begin[k] : avg %8.1f ns  (= avg duration + avg gap)" % (sum(periods) / len(periods)))
    if unprofiled is not None:
        print("  un-profiled bench.py line, HIP events / K                : %8.1f ns per launch" % (unprofiled * 1e3))
    b, e = regs[order[R // 2]]
    print()
    print("median region, dispatch by dispatch (ns):")
    print("   k  duration  gap_to_next  period")
    for k in range(K):
        print("  %2d  %8d  %11s  %6s" % (k, e[k] - b[k], "" if k == K - 1 else b[k + 1] - e[k], "" if k == K - 1 else b[k + 1] - b[k]))


if __name__ == "__main__":
    main()
