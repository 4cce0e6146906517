#!/bin/bash
# round 4: the slot meeting at the batch-64 shard, and with / without the LDS rows on the packed single launch
for r in 1 2 3; do for m in 0 1; do
  CNNQ_MEET_SLOTS=$m python bench.py --batch 64 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b64 SLOTS=$m round $r: %.4f ms  frac %.3f  verified %s status %d' % (d['ms_per_step'], d['roofline']['frac'], d['verified'], d['group_status']))"
done; done
for cfg in "1 0" "1 -1" "1 0" "1 -1"; do set -- $cfg; echo "SLOTS=$1 KL=$2"; CNNQ_MEET_SLOTS=$1 CNNQ_FLAT_KL=$2 python tools/bench_pack_single.py 2>&1 | tail -1; done
