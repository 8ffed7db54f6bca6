#!/usr/bin/env python3
"""Print the kernel_stats.csv of a rocprofv3 --stats output directory: tools/kstats.py <dir> [name filter]"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(f)):
    if flt in r["Name"]:
        print("%-50s calls %5s avg %7.1f us total %7.2f ms" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
