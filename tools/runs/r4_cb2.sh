#!/bin/bash
# round 4 (final library): dispatch block size of the flat tiles again, now with the slot meeting
for r in 1 2; do for cb in 4 1 2 8; do
  CNNQ_GRP_CB=$cb python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b512 CB=$cb round $r: %.3f ms  frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
done; done
