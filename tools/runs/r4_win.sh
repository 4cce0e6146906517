#!/bin/bash
# round 4: the slot meeting with poll windows of 4 / 2 slots per lane (registers: occupancy of the K = 16 / 8 kernels restored)
O=$PWD/gpurun_out/r4_win; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
F='s/\| A=.*Gs= *([0-9]+) wgs= *([0-9]+) \| chain +([0-9.]+) us.*group +([0-9.]+) us +([0-9]+) GB.*mismatches=([0-9]+).*/| Gs \1 wgs \2 group \4 us \5 GB\/s(8B) mismatches \6/'
for m in 0 1 0 1; do echo "SLOTS=$m"; CNNQ_MEET_SLOTS=$m python tools/bench_group.py --rounds 1 --reps 20 --shapes 1024x14,512x14,2048x7,256x14,512x7 2>&1 | grep "^C=" | sed -E "$F"; done
for r in 1 2; do for m in 0 1; do
  CNNQ_MEET_SLOTS=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b512 SLOTS=$m round $r: %.3f ms  frac %.3f group %.3f verified %s status %d' % (d['ms_per_step'], d['roofline']['frac'], list(d['roofline_other_kernels'].values())[0]['frac'], d['verified'], d['group_status']))"
done; done
for r in 1 2 3; do for m in 0 1; do
  CNNQ_MEET_SLOTS=$m python bench.py --batch 64 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b64 SLOTS=$m round $r: %.4f ms  frac %.3f  verified %s status %d' % (d['ms_per_step'], d['roofline']['frac'], d['verified'], d['group_status']))"
done; done
