#!/usr/bin/env python3
"""Timeline of one bench step from a rocprofv3 --kernel-trace CSV: start offset, duration and stream of every kernel between two
consecutive k_bow launches, and the idle time on the critical path.  usage: step_timeline.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
bows = [i for i, r in enumerate(rows) if "k_bow" in r["Kernel_Name"]]
a, b = bows[-2], bows[-1]
step = rows[a + 1:b + 1]
t0 = int(rows[a]["End_Timestamp"])
busy_until = t0
idle = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    if s > busy_until:
        idle += s - busy_until
    busy_until = max(busy_until, e)
    print("%-42s start %8.1f us  dur %7.1f us  queue %s" % (name, (s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?")))
print("step %.1f us, no kernel running for %.1f us" % ((int(step[-1]["End_Timestamp"]) - t0) / 1e3, idle / 1e3))
