#!/usr/bin/env python3
"""Per-kernel mean of every counter in rocprofv3 counter_collection CSVs.  usage: pmc_summary.py <kernel substring> file.csv [file.csv ...]"""
import csv, sys, collections
sub = sys.argv[1]
acc = collections.defaultdict(list)
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in sorted(acc):
    v = acc[k]
    print("%-32s n=%3d mean %.4g" % (k, len(v), sum(v) / len(v)))
