#!/usr/bin/env python3
"""Top kernels by total time from a rocprofv3 *_kernel_stats.csv.  usage: prof_top.py file.csv nsteps [n]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
if n <= 0:        # 0 = count the steps: one adamw_kernel launch per optimizer step
    n = float(sum(int(r["Calls"]) for r in rows if "adamw_kernel" in r["Name"]) or 1)
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total %.2f ms/step" % (tot / n / 1e6))
for r in sorted(rows, key=lambda r: -int(r['TotalDurationNs']))[:top]:
    nm = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
    nm = re.sub(r'^void ', '', nm)
    print("%7.2f ms %5.1f%%  calls %5d  avg %8.1f us  %s" % (int(r['TotalDurationNs']) / n / 1e6, 100.0 * int(r['TotalDurationNs']) / tot,
          int(r['Calls']) / n, float(r['AverageNs']) / 1e3, nm[:110]))
