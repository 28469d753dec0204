#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace + PMC passes) into a per-kernel table."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def short(name):
    name = name.replace("ta::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:48]


def main(out):
    # kernel trace -> avg duration
    dur = defaultdict(list)
    for f in find(os.path.join(out, "trace"), "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"{'kernel':50s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s}")
    tot = sum(sum(v) for v in dur.values())
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:50s} {len(v):6d} {sum(v)/len(v)/1e3:10.2f} {sum(v)/1e6:10.3f}  {100*sum(v)/tot:5.1f}%")
    # PMC passes -> per-kernel average counter value
    counters = defaultdict(lambda: defaultdict(list))
    for sub in ("pmc_sq", "pmc_sq2", "pmc_grbm", "pmc_fetch", "pmc_write"):
        for f in find(os.path.join(out, sub), "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                counters[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for k in counters for c in counters[k]})
    print()
    for k in sorted(counters, key=lambda k: -sum(dur.get(k, [0]))):
        print(k)
        for c in names:
            v = counters[k].get(c)
            if v:
                print(f"    {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")


    # HBM traffic per launch of the dominant kernels, corrected as MI355X_MICROARCH.md
    # section HBM prescribes: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
    # half of the bytes of a 16 B/lane coalesced read, so it is doubled.
    import json
    traffic = {}
    for k in counters:
        f, w = counters[k].get("FETCH_SIZE"), counters[k].get("WRITE_SIZE")
        if f and w:
            traffic[k] = {"fetch_KiB_raw": sum(f) / len(f), "write_KiB": sum(w) / len(w),
                          "hbm_bytes_per_launch": (2.0 * sum(f) / len(f) + sum(w) / len(w)) * 1024.0}
            v = counters[k].get("SQ_INSTS_VALU")
            if v:  # wavefront-level VALU instructions per launch (issue-bound kernels)
                traffic[k]["valu_wave_insts_per_launch"] = sum(v) / len(v)
    # stamp: the sources the counters were measured on (bench.py drops them when its own differ)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import subprocess
    from bench import source_stamp
    try:
        commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True,
                                text=True).stdout.strip() or None
    except Exception:
        commit = None
    import hashlib
    traffic["_stamp"] = {"source_sha": source_stamp(), "commit": commit or os.environ.get("TA_COMMIT"),
                         "bench_sha16": hashlib.sha256(open(os.path.join(root, "bench.py"), "rb").read()).hexdigest()[:16]}
    with open(os.path.join(out, "pmc_traffic.json"), "w") as fp:
        json.dump(traffic, fp, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
