#!/usr/bin/env python3
"""Summary of tools/r6_family_pmc.sh: per profiled program and kernel, mean FETCH_SIZE / WRITE_SIZE per launch -> bytes (FETCH_SIZE x 2 x 1024:
the guide's gfx950 correction; WRITE_SIZE x 1024), the launch time under the profiler, GB/s, and the algorithmic bytes where the driver printed
them (`ALG` lines) or they follow from the shape.  Writes <dir>/../r6_family_pmc.json.  usage: family_pmc_summary.py <dir>"""
import collections
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
tags = sorted({os.path.basename(p)[:-len("_FETCH_SIZE.log")] for p in glob.glob(os.path.join(d, "*_FETCH_SIZE.log"))})
SKIP = ("at::native", "rocclr", "distribution", "elementwise", "fill", "Fill")
out = {}
print("%-18s %-44s %8s %10s %10s %10s %8s %9s" % ("program", "kernel", "launches", "read MB", "written MB", "alg MB", "ratio", "GB/s"))
for tag in tags:
    alg = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        log = os.path.join(d, "%s_%s.log" % (tag, c))
        if os.path.exists(log):
            for line in open(log, errors="replace"):
                m = re.match(r"ALG (\S+) (\S+) (\d+)", line)
                if m:
                    alg[m.group(2)] = int(m.group(3))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for path in glob.glob(os.path.join(d, "%s_%s" % (tag, c), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                k = r["Kernel_Name"]
                if any(s in k for s in SKIP) or r["Counter_Name"] != c:
                    continue
                short = re.sub(r"^void ", "", k).replace("(anonymous namespace)::", "")
                short = re.sub(r"\(.*", "", short)                                   # drop the argument list, keep the template arguments
                short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", short)[:44]
                acc[short][c].append(float(r["Counter_Value"]))
                if c == "FETCH_SIZE":
                    acc[short]["ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, v in sorted(acc.items()):
        if not v["FETCH_SIZE"] or not v["WRITE_SIZE"]:
            continue
        n = len(v["FETCH_SIZE"])
        # the first launch of a program includes cold caches: take the median
        med = lambda a: sorted(a)[len(a) // 2]      # noqa: E731
        rd, wr, ns = med(v["FETCH_SIZE"]) * 2048.0, med(v["WRITE_SIZE"]) * 1024.0, med(v["ns"])
        a = next((b for s, b in alg.items() if s in k), None)
        out.setdefault(tag, {})[k] = {"launches_profiled": n, "read_bytes": rd, "write_bytes": wr, "ns_under_profiler": ns, "algorithmic_bytes": a,
                                      "traffic_over_algorithmic": (rd + wr) / a if a else None, "gbps": (rd + wr) / ns}
        print("%-18s %-44s %8d %10.1f %10.1f %10s %8s %9.0f" % (tag, k, n, rd / 1e6, wr / 1e6, "%.1f" % (a / 1e6) if a else "-",
                                                                 "%.2f" % ((rd + wr) / a) if a else "-", (rd + wr) / ns))
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(d)), (sys.argv[2] if len(sys.argv) > 2 else "r6_family_pmc") + ".json"), "w"), indent=1)
