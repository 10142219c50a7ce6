#!/usr/bin/env python3
"""Whole-step counters from rocprofv3 --pmc passes over `bench.py --no-graphs` (one counter_collection.csv per pass).
usage: step_pmc_summary.py nsteps [--after-first-adamw] file.csv [file.csv ...]   -> MFMA pipe utilisation and HBM-side traffic per step."""
import collections
import csv
import sys

nsteps = float(sys.argv[1])
files = sys.argv[2:]
# --after-first-adamw: count only the dispatches that follow the first optimizer kernel (a run of `--warmup 1 --steps 1`: the second step alone,
# without the one-time initialisation and warm-up launches)
after = "--after-first-adamw" in files
files = [f for f in files if not f.startswith("--")]
tot = collections.Counter()
dur = collections.Counter()
for f in files:
    seen = set()
    rows = list(csv.DictReader(open(f)))
    if after:
        first = min((int(r["Dispatch_Id"]) for r in rows if "adamw_kernel" in r["Kernel_Name"]), default=-1)
        rows = [r for r in rows if int(r["Dispatch_Id"]) > first]
    for r in rows:
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], r["Counter_Name"])
        if key not in seen:
            seen.add(key)
            dur[r["Counter_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k in sorted(tot):
    print("%-28s total %.4g   kernel time under the profiler %.1f ms per step" % (k, tot[k], dur[k] / nsteps / 1e6))
if "SQ_VALU_MFMA_BUSY_CYCLES" in tot and "GRBM_GUI_ACTIVE" in tot:
    # busy cycles are summed over 1024 SIMDs, GUI_ACTIVE over 8 XCDs
    print("MFMA pipe busy: %.1f %% of the cycles of all kernels" % (100.0 * (tot["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (tot["GRBM_GUI_ACTIVE"] / 8.0)))
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    rd, wr = tot["FETCH_SIZE"] * 2 * 1024 / nsteps, tot["WRITE_SIZE"] * 1024 / nsteps        # KiB; gfx950: FETCH_SIZE counts 64-byte units as 32
    print("HBM-side traffic per step: %.1f GB read + %.1f GB written" % (rd / 1e9, wr / 1e9))
