#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one counter set per run, as the pool requires) into one JSON per kernel.

    python tools/pmc_summary.py [--stamp ROWS_PER_LAUNCH "COMMAND"] gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq > profiles/<name>.json
--stamp adds what bench.py checks before it reports `roofline.traffic` from PMC_TRAFFIC.json (a copy of the newest summary):
the sha of the kernel sources the counters were collected on (`_csrc_sha`, bench.csrc_sha) and the rows per launch.
HBM-side bytes as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE / WRITE_SIZE are KiB counted at the L2's fabric side
(Infinity-Cache hits included); on gfx950 FETCH_SIZE reports half the bytes of coalesced reads (checked here on k_step_mid,
whose reads are exactly 8 B per cell), WRITE_SIZE is exact.  Values are averages per launch.
"""
import collections
import csv
import json
import os
import sys

args = sys.argv[1:]
stamp = None
if args and args[0] == "--stamp":
    stamp = (int(args[1]), args[2])
    args = args[3:]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for d in args:
    if os.path.isdir(d):
        hits = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")]
        d = hits[0] if hits else d
    path = os.path.join(d, "p_counter_collection.csv") if os.path.isdir(d) else d
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "dp::" not in name:
            continue
        per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for name, ctr in per.items():
    e = {"launches": max(len(v) for v in ctr.values())}
    for c, v in ctr.items():
        e[c] = sum(v) / len(v)
    if "FETCH_SIZE" in e:
        e["read_bytes_corrected"] = e["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in e:
        e["write_bytes"] = e["WRITE_SIZE"] * 1024
    if "read_bytes_corrected" in e and "write_bytes" in e:
        e["traffic_bytes"] = e["read_bytes_corrected"] + e["write_bytes"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("SQ_BUSY_CYCLES"):
        # SQ_BUSY_CYCLES is summed over the 32 shader engines; 1024 SIMDs share the MFMA-busy total
        e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["SQ_BUSY_CYCLES"] / 32 * 1024)
    out[name.split("(")[0].replace("void ", "")] = e
if stamp:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["_csrc_sha"] = bench.csrc_sha()
    out["_rows_per_launch"] = stamp[0]
    out["_command"] = stamp[1]
json.dump(out, sys.stdout, indent=1, sort_keys=True)
print()
