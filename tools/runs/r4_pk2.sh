#!/bin/bash
# round 4: the fast quotient two elements per instruction (v_pk_*_f32) in k_mmq_flat: parity suites, then A/B against the previous build
O=$PWD/gpurun_out/r4_pk2; mkdir -p $O
timeout 1500 python -m pytest tests/test_fastdiv_gpu.py tests/test_group_gpu.py tests/test_single_outputs_gpu.py tests/test_full_size_gpu.py -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for r in 1 2 3; do for lib in tools/alt/libcnnq_prev.so ""; do
  CNNQ_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b512 ${lib:-new} round $r: %.3f ms  frac %.3f verified %s' % (d['ms_per_step'], d['roofline']['frac'], d['verified']))"
done; done
for r in 1 2; do for lib in tools/alt/libcnnq_prev.so ""; do
  CNNQ_HIP_LIB=$lib python bench.py --batch 64 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b64 ${lib:-new} round $r: %.4f ms  frac %.3f verified %s' % (d['ms_per_step'], d['roofline']['frac'], d['verified']))"
done; done
for lib in tools/alt/libcnnq_prev.so "" tools/alt/libcnnq_prev.so ""; do echo "${lib:-new}"; CNNQ_HIP_LIB=$lib python tools/bench_pack_single.py 2>&1 | tail -1; done
