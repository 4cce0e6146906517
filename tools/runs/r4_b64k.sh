#!/bin/bash
# round 4: tile height sweep at the batch-64 shard (what an 8-GPU run gives every rank)
for r in 1 2; do for k in 0 16 8; do
  CNNQ_GRP_K=$k python bench.py --batch 64 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b64 K=$k round $r: %.4f ms  frac %.3f  %s verified %s' % (d['ms_per_step'], d['roofline']['frac'], {k: round(v['frac'],3) for k,v in d['roofline_other_kernels'].items()}, d['verified']))"
done; done
for t in 512 768 1536 2048; do
  CNNQ_GRP_WGS=$t python bench.py --batch 64 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b64 target WGs=$t: %.4f ms  frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
done
