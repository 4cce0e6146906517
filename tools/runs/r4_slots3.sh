#!/bin/bash
# round 4: the slot meeting polled by wave 0 alone: suites with it forced on, A/B at b512 / b64 / packed
O=$PWD/gpurun_out/r4_slots; mkdir -p $O
CNNQ_MEET_SLOTS=1 timeout 1500 python -m pytest tests/test_group_gpu.py tests/test_single_outputs_gpu.py tests/test_full_size_gpu.py tests/test_concurrent_gpu.py tests/test_xrank_gpu.py tests/test_graph_gpu.py -q > $O/pytest_slots.log 2>&1; tail -2 $O/pytest_slots.log
for r in 1 2 3; do for m in 0 1; do
  CNNQ_MEET_SLOTS=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b512 SLOTS=$m round $r: %.3f ms  frac %.3f  verified %s status %d' % (d['ms_per_step'], d['roofline']['frac'], d['verified'], d['group_status']))"
done; done
for r in 1 2 3; do for m in 0 1; do
  CNNQ_MEET_SLOTS=$m python bench.py --batch 64 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b64 SLOTS=$m round $r: %.4f ms  frac %.3f  verified %s status %d' % (d['ms_per_step'], d['roofline']['frac'], d['verified'], d['group_status']))"
done; done
for m in 0 -1 0 -1; do echo "SLOTS=$m"; CNNQ_MEET_SLOTS=$m python tools/bench_pack_single.py 2>&1 | tail -1; done
