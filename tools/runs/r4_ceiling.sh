#!/bin/bash
# round 4: the ceiling report of one box (tools/ceiling_report.py) next to its bench line
O=$PWD/gpurun_out/r4_ceiling; mkdir -p $O
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err
B=$(python -c "import json; print(json.load(open('$O/bench.json'))['box'])")
timeout 900 python tools/ceiling_report.py > $O/ceiling_$B.md 2> $O/ceiling.err
cp $O/bench.json $O/bench_$B.json
python - <<PY
import json
d = json.load(open('$O/bench.json'))
print(d['box'], '%.3f ms/step' % d['ms_per_step'], '%.1f G elem/s' % (d['value'] / 1e9), 'frac %.3f' % d['roofline']['frac'], {k: round(v['frac'], 3) for k, v in d['roofline_other_kernels'].items()}, d['verified'], d['group_status'])
PY
cat $O/ceiling_$B.md
