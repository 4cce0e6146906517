#!/bin/bash
# A/B on ONE box: bench.py's step with the flat tiles on / off (CNNQ_GRP_FLAT), interleaved, three rounds each
O=$PWD/gpurun_out/r3_ab; mkdir -p $O
for r in 1 2 3; do
  for f in 1 0; do
    CNNQ_GRP_FLAT=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/b_${f}_$r.json 2>/dev/null
    python - <<PY
import json
d=json.load(open('$O/b_${f}_$r.json'))
print('flat=$f round $r: %.3f ms/step  %.1f G elem/s  kernel frac %.3f  verified %s' % (d['ms_per_step'], d['value']/1e9, d['roofline']['frac'], d['verified']))
PY
  done
done
