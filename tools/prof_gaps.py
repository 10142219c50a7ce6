#!/usr/bin/env python3
"""Idle time between kernels from a rocprofv3 kernel trace: usage prof_gaps.py <kernel_trace.csv> [steps]
The device is idle between two kernels when the next one starts after EVERY earlier kernel has ended.  Prints the span, the busy union,
the idle total, and the idle time grouped by the kernel that FOLLOWS the gap (the one whose launch was late)."""
import collections, csv, sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort()
# the timed region = the last `steps` optimizer launches backwards: take everything after the (steps+1)-th last adamw kernel
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
if steps <= 0:              # no optimizer in this workload: the last 60 % of the trace (past the warm-up)
    rows = rows[int(len(rows) * 0.4):]
    steps = 1
elif len(ad) > steps:
    rows = rows[ad[-steps - 1] + 1:ad[-1] + 1]
span = rows[-1][1] - rows[0][0]
busy_end = rows[0][0]
idle = 0
by = collections.Counter()
cnt = collections.Counter()
hist = collections.Counter()
for s, e, n in rows:
    if s > busy_end:
        g = s - busy_end
        idle += g
        nm = n[5:] if n.startswith("void ") else n
        key = nm.replace("(anonymous namespace)::", "").split("(")[0][-60:] or nm[:60]
        by[key] += g
        cnt[key] += 1
        hist[min(int(g / 1000), 50)] += 1
    busy_end = max(busy_end, e)
print("kernels %d  span %.2f ms  idle %.2f ms (%.1f %%)  per step: span %.2f ms idle %.2f ms" % (len(rows), span / 1e6, idle / 1e6, 100.0 * idle / span, span / 1e6 / steps, idle / 1e6 / steps))
print("gap histogram (us: count):", sorted(hist.items()))
for k, v in by.most_common(25):
    print("%9.1f us/step  %5.1f gaps/step  avg %6.1f us  before %s" % (v / 1e3 / steps, cnt[k] / steps, v / 1e3 / cnt[k], k))
