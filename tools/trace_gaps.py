#!/usr/bin/env python3
"""Kernel time against wall span of a run of launches: trace_gaps.py <kernel_trace.csv> <name regex> - for the LAST
contiguous run of dispatches whose kernels match the regex (or the tiny layout kernel in between): launches, sum of the
kernels' durations, first-start -> last-end span, and the idle time between them."""
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
pat = re.compile(sys.argv[2])
runs, cur = [], []
for r in rows:
    if pat.search(r['Kernel_Name']):
        cur.append(r)
    elif cur:
        runs.append(cur)
        cur = []
if cur:
    runs.append(cur)
runs = [r for r in runs if len(r) >= 50]
for run in runs[-3:]:
    dur = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in run)
    span = int(run[-1]['End_Timestamp']) - int(run[0]['Start_Timestamp'])
    by = {}
    for r in run:
        k = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', ''))[:40]
        by.setdefault(k, [0, 0])
        by[k][0] += 1
        by[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    print('%d launches: kernels %.3f ms, span %.3f ms, idle %.3f ms | %s' % (
        len(run), dur / 1e6, span / 1e6, (span - dur) / 1e6, {k: (v[0], round(v[1] / 1e6, 3)) for k, v in by.items()}))
