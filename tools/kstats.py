#!/usr/bin/env python3
"""Print the library's kernels from a rocprofv3 --kernel-trace --stats CSV: kstats.py <kernel_stats.csv>"""
import csv
import sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "k_" in n and "at::" not in n:
        n = n.replace("(anonymous namespace)::", "").split("(")[0]
        print("%-56s calls %5s avg %9.1f us  min %8.1f max %8.1f  total %9.2f ms" % (
            n[:56], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
            float(r["TotalDurationNs"]) / 1e6))
