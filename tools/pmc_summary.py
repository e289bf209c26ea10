#!/usr/bin/env python3
"""Folds rocprofv3 --pmc counter_collection.csv files (one pass per counter) into
profiles/pmc_traffic.json, keyed by boards per launch.

    python tools/pmc_summary.py BOARDS fetch_dir write_dir [label] [state_bytes_per_board] [extra_bytes_per_board key_suffix]

extra_bytes_per_board / key_suffix: e.g. `30 gym` for the step kernel that also writes the observation.

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide
coalesced read (MI355X_MICROARCH.md §HBM), so it is doubled.  Only the step kernel's
dispatches are used."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# which dispatches to fold, e.g. PMC_KERNEL_FILTER="false, true>" for the observation-writing step kernel
KERNEL_FILTER = os.environ.get("PMC_KERNEL_FILTER", "step_kernel")


def mean_counter(d, name):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and KERNEL_FILTER in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit("no %s rows for step_kernel under %s" % (name, d))
    return sum(vals) / len(vals), len(vals)


STEP_SOURCES = ("qttt_state.h", "qttt_step_core.h", "qttt_observation.h", "qttt_step_kernels.h")


def step_sources_sha256():
    """Fingerprint of the sources the step kernels are made of: bench.py reports whether the committed traffic figure
    was collected on the build it is running (`roofline.traffic_measured_on_this_build`)."""
    import hashlib
    h = hashlib.sha256()
    for f in STEP_SOURCES:
        h.update(open(os.path.join(ROOT, "qtttgym_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main():
    boards, fetch_dir, write_dir = sys.argv[1], sys.argv[2], sys.argv[3]
    label = sys.argv[4] if len(sys.argv) > 4 else ""
    state_bytes = int(sys.argv[5]) if len(sys.argv) > 5 else 16
    extra = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    suffix = sys.argv[7] if len(sys.argv) > 7 else ""
    fetch_kib, nf = mean_counter(fetch_dir, "FETCH_SIZE")
    write_kib, nw = mean_counter(write_dir, "WRITE_SIZE")
    entry = {
        "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB": write_kib,
        "read_bytes": 2 * fetch_kib * 1024, "write_bytes": write_kib * 1024,
        "hbm_bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024,
        "dispatches": [nf, nw], "label": label, "state_bytes_per_board": state_bytes,
        "algorithmic_bytes_per_launch": (2 * state_bytes + 7 + extra) * int(boards),
        "note": "FETCH_SIZE doubled (gfx950 half-count of wide coalesced reads); separate --pmc passes",
        "step_sources_sha256": step_sources_sha256(),
    }
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    d = json.load(open(path)) if os.path.exists(path) else {}
    d["%s@%dB%s" % (boards, state_bytes, ("+" + suffix) if suffix else "")] = entry
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(entry))


if __name__ == "__main__":
    main()
