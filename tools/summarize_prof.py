#!/usr/bin/env python3
"""Aggregate a rocprofv3 --kernel-trace --stats CSV by kernel family (template variants merged).
usage: summarize_prof.py <kernel_stats.csv> [<out.md>]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
fam = {}
for r in rows:
    m = re.search(r'::(k_[a-z_]+)<', r['Name']) or re.search(r'::(k_[a-z_]+)\(', r['Name'])
    name = m.group(1) if m else ('torch/other: ' + r['Name'][:60])
    f = fam.setdefault(name, dict(calls=0, total=0))
    f['calls'] += int(r['Calls'])
    f['total'] += int(r['TotalDurationNs'])
lines = ['| kernel family | calls | total ms | avg us |', '|---|---|---|---|']
for name, f in sorted(fam.items(), key=lambda kv: -kv[1]['total']):
    if name.startswith('k_'):
        lines.append('| %s | %d | %.3f | %.2f |' % (name, f['calls'], f['total'] / 1e6, f['total'] / f['calls'] / 1e3))
other = sum(f['total'] for n, f in fam.items() if not n.startswith('k_'))
lines.append('| (torch kernels: workload generation, copies) | - | %.3f | - |' % (other / 1e6))
out = '\n'.join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], 'a').write(out + '\n')
