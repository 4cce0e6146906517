#!/bin/bash
# round 4: the slot meeting in k_mmq_group too (default on): the single-launch suites, A/B at b512 / b64, the small layers one by one
O=$PWD/gpurun_out/r4_slots; mkdir -p $O
timeout 1500 python -m pytest tests/test_group_gpu.py tests/test_single_outputs_gpu.py tests/test_full_size_gpu.py tests/test_concurrent_gpu.py tests/test_xrank_gpu.py tests/test_graph_gpu.py tests/test_fuzz_gpu.py tests/test_parity_holes_gpu.py -q > $O/pytest_slots_group.log 2>&1; tail -2 $O/pytest_slots_group.log
for r in 1 2 3; do for m in 0 1; do
  CNNQ_MEET_SLOTS=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b512 SLOTS=$m round $r: %.3f ms  frac %.3f group %.3f verified %s status %d' % (d['ms_per_step'], d['roofline']['frac'], list(d['roofline_other_kernels'].values())[0]['frac'], d['verified'], d['group_status']))"
done; done
for r in 1 2 3; do for m in 0 1; do
  CNNQ_MEET_SLOTS=$m python bench.py --batch 64 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b64 SLOTS=$m round $r: %.4f ms  frac %.3f  verified %s status %d' % (d['ms_per_step'], d['roofline']['frac'], d['verified'], d['group_status']))"
done; done
F='s/\| A=.*Gs= *([0-9]+) wgs= *([0-9]+) \| chain +([0-9.]+) us.*group +([0-9.]+) us +([0-9]+) GB.*mismatches=([0-9]+).*/| Gs \1 wgs \2 group \4 us \5 GB\/s(8B) mismatches \6/'
for m in 0 1 0 1; do echo "SLOTS=$m"; CNNQ_MEET_SLOTS=$m python tools/bench_group.py --rounds 1 --reps 20 --shapes 1024x14,512x14,2048x7,256x14,512x7 2>&1 | grep "^C=" | sed -E "$F"; done
