#!/usr/bin/env python3
"""Reads the rocprofv3 JSON of `rocprofv3 --kernel-trace --pmc <TCC counter> --output-format json -- python tools/stack_probe.py`
and prints, per buffer pair, the launch time next to how evenly the counter spreads over the 16 L2 channel instances (summed
over the 8 XCDs) and over the 128 (XCD, instance) units: stack_probe_report.py <results.json> [...]"""
import json
import sys

import numpy as np


def main():
    for path in sys.argv[1:]:
        r = json.load(open(path))['rocprofiler-sdk-tool'][0]
        names = {k['kernel_id']: k.get('formatted_kernel_name', k.get('kernel_name', '')) for k in r['kernel_symbols']}
        cname = r['counters'][0]['name'] if r.get('counters') else '?'
        rows = []
        for rec in r['callback_records']['counter_collection']:
            di = rec['dispatch_data']['dispatch_info']
            if 'k_mmq_flat' not in names.get(di['kernel_id'], '') or di['grid_size']['x'] < 256 * 10000:
                continue
            v = np.array([x['value'] for x in rec['records']], dtype=np.float64)
            dt = (rec['dispatch_data']['end_timestamp'] - rec['dispatch_data']['start_timestamp']) / 1e3
            rows.append((dt, v))
        print('%s: counter %s, %d launches of k_mmq_flat on [512,256,56,56], %d values per launch' % (path.split('/')[-2], cname, len(rows), len(rows[0][1]) if rows else 0))
        reps = 4
        print('%-5s %9s %14s %12s %12s %12s %12s' % ('pair', 'us', 'total', 'max/mean 16', 'cv 16', 'max/mean 128', 'cv 128'))
        for p in range(len(rows) // reps):
            grp = rows[p * reps + 1:(p + 1) * reps]                     # the first launch on a pair is its warm-up
            us = np.mean([g[0] for g in grp])
            v = np.mean([g[1] for g in grp], axis=0)
            n = len(v)
            inst = v.reshape(-1, 16).sum(0) if n % 16 == 0 else v      # records come XCD-major: [xcc][instance]
            print('%-5d %9.1f %14.0f %12.4f %12.4f %12.4f %12.4f' % (p, us, v.sum(), inst.max() / inst.mean(), inst.std() / inst.mean(),
                                                                    v.max() / v.mean(), v.std() / v.mean()))


if __name__ == '__main__':
    main()
