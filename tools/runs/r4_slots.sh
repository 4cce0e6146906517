#!/bin/bash
# round 4: the slot meeting of k_mmq_flat (CNNQ_MEET_SLOTS=1): parity of the suites that reach the flat tiles, then A/B
O=$PWD/gpurun_out/r4_slots; mkdir -p $O
CNNQ_MEET_SLOTS=1 timeout 1500 python -m pytest tests/test_group_gpu.py tests/test_single_outputs_gpu.py tests/test_full_size_gpu.py tests/test_concurrent_gpu.py tests/test_xrank_gpu.py tests/test_graph_gpu.py -q > $O/pytest_slots.log 2>&1; tail -3 $O/pytest_slots.log
for r in 1 2 3; do for m in 0 1; do
  CNNQ_MEET_SLOTS=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('SLOTS=$m round $r: %.3f ms  frac %.3f  verified %s status %d' % (d['ms_per_step'], d['roofline']['frac'], d['verified'], d['group_status']))"
done; done
for m in 0 1 0 1; do echo "SLOTS=$m"; CNNQ_MEET_SLOTS=$m python tools/bench_pack_single.py 2>&1 | tail -1; done
F='s/\| A=.*Gs= *([0-9]+) wgs= *([0-9]+) \| chain +([0-9.]+) us.*group +([0-9.]+) us +([0-9]+) GB.*mismatches=([0-9]+).*/| Gs \1 wgs \2 group \4 us \5 GB\/s(8B) mismatches \6/'
for m in 0 1; do echo "SLOTS=$m"; CNNQ_MEET_SLOTS=$m python tools/bench_group.py --rounds 1 --reps 10 --shapes 64x112,256x56,128x56,512x28,64x56,256x28,128x28 2>&1 | grep "^C=" | sed -E "$F"; done
