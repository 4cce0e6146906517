#!/usr/bin/env python3
"""Per-kernel-family, per-grid-size averages of the counters of one rocprofv3 --pmc pass.
usage: summarize_counters.py <counter_collection.csv> [family regex, default k_mmq|k_qdq|k_minmax]"""
import csv
import re
import sys
from collections import defaultdict

pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else r'(k_mmq_[a-z]+|k_qdq|k_minmax[a-z_]*|k_pipe[a-z_]*)<')
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
dur = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    m = pat.search(r['Kernel_Name'])
    if not m:
        continue
    key = (m.group(1), int(r['Grid_Size']) // int(r['Workgroup_Size']), r['VGPR_Count'])
    a = acc[key][r['Counter_Name']]
    a[0] += 1
    a[1] += float(r['Counter_Value'])
    d = dur[key]
    d[0] += 1
    d[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for key in sorted(acc):
    print('%s  workgroups=%d  vgprs=%s  avg duration under the counters %.1f us' % (key[0], key[1], key[2], dur[key][1] / dur[key][0]))
    for c in sorted(acc[key]):
        n, v = acc[key][c]
        print('    %-40s %16.1f per launch (%d launches)' % (c, v / n, n))
