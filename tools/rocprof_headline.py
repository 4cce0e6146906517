#!/usr/bin/env python3
"""profiles/rNN_rocprof_headline.json and rNN_pmc_traffic.json from one box's rocprofv3 outputs:
    rocprof_headline.py <box id> <kernel_stats b512.csv> [<kernel_stats b64.csv>] [--pmc <fetch.csv> <write.csv>] --out <dir> [--round r05]
Average launch duration per kernel family of the bench command (what bench.py's roofline.frac_rocprof reads) and, with
--pmc, the HBM bytes per launch (FETCH_SIZE doubled: gfx950, MI355X_MICROARCH.md)."""
import csv, json, os, re, sys


def fam(name):
    m = re.search(r'(k_mmq_flat|k_mmq_group|k_mmq_whole|k_fused_flat|k_fused_group|k_stats_flat|k_moments|k_absdev|k_mt_qdq|k_qdq|k_minmax)\b', name)
    return m.group(1) if m else None


def stats(path):
    acc = {}
    for r in csv.DictReader(open(path)):
        f = fam(r['Name'])
        if f:
            a = acc.setdefault(f, [0, 0.0])
            a[0] += int(r['Calls']); a[1] += float(r['TotalDurationNs'])
    return {k: v[1] / v[0] / 1e3 for k, v in acc.items() if v[0]}


def pmc(path):
    acc = {}
    for r in csv.DictReader(open(path)):
        f = fam(r['Kernel_Name'])
        if f:
            a = acc.setdefault(f, [0, 0.0])
            a[0] += 1; a[1] += float(r['Counter_Value'])
    return acc


a = sys.argv[1:]
out = a[a.index('--out') + 1]
RND = a[a.index('--round') + 1] if '--round' in a else 'r04'
box = a[0]
rec = {'box': box, 'source': 'rocprofv3 --kernel-trace --stats of `python bench.py --batch B --steps 5 --warmup 2 --no-cpu-baseline '
       '--no-other-configs` (tools/runs/%s_profile.sh), same box and call as profiles/%s_bench_line_b512.json' % (RND.replace('0', ''), RND), 'avg_launch_us': {}}
for k, v in stats(a[1]).items():
    rec['avg_launch_us']['%s@b512' % k] = round(v, 2)
if len(a) > 2 and a[2].endswith('.csv') and a[2] != '--pmc':
    for k, v in stats(a[2]).items():
        rec['avg_launch_us']['%s@b64' % k] = round(v, 2)
json.dump(rec, open(os.path.join(out, '%s_rocprof_headline.json' % RND), 'w'), indent=1)
if '--pmc' in a:
    i = a.index('--pmc')
    fe, wr = pmc(a[i + 1]), pmc(a[i + 2])
    t = {'box': box, 'source': 'rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes of `python bench.py --batch 512 --steps 2 '
         '--warmup 1 --no-cpu-baseline --no-other-configs` (tools/runs/%s_profile.sh)' % RND.replace('0', ''), 'bytes_per_launch': {}}
    for k in fe:
        if k in wr:
            t['bytes_per_launch']['%s@b512' % k] = round(fe[k][1] * 1024 * 2 / fe[k][0] + wr[k][1] * 1024 / wr[k][0])
    json.dump(t, open(os.path.join(out, '%s_pmc_traffic.json' % RND), 'w'), indent=1)
print(json.dumps(rec))
