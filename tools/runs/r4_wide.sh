#!/bin/bash
# round 4: k_mmq_group blocks that are whole multiples of 128 bytes (two column steps; CNNQ_GRP_WIDE): suites, small layers A/B, steps A/B
O=$PWD/gpurun_out/r4_wide; mkdir -p $O
timeout 1500 python -m pytest tests/test_group_gpu.py tests/test_single_outputs_gpu.py tests/test_full_size_gpu.py tests/test_xrank_gpu.py tests/test_fuzz_gpu.py tests/test_concurrent_gpu.py -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
F='s/\| A=.*Gs= *([0-9]+) wgs= *([0-9]+) \| chain +([0-9.]+) us.*group +([0-9.]+) us +([0-9]+) GB.*mismatches=([0-9]+).*/| Gs \1 wgs \2 group \4 us \5 GB\/s(8B) mismatches \6/'
for cfg in "0 0" "1 0" "1 16" "1 32" "0 0" "1 0"; do set -- $cfg; echo "WIDE=$1 K=$2"; CNNQ_GRP_WIDE=$1 CNNQ_GRP_K=$2 python tools/bench_group.py --rounds 1 --reps 20 --shapes 1024x14,512x14,2048x7,256x14,512x7 2>&1 | grep "^C=" | sed -E "$F"; done
for r in 1 2 3; do for m in 0 1; do
  CNNQ_GRP_WIDE=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b512 WIDE=$m round $r: %.3f ms  frac %.3f group %.3f verified %s status %d' % (d['ms_per_step'], d['roofline']['frac'], list(d['roofline_other_kernels'].values())[0]['frac'], d['verified'], d['group_status']))"
done; done
for r in 1 2; do for m in 0 1; do
  CNNQ_GRP_WIDE=$m python bench.py --batch 64 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b64 WIDE=$m round $r: %.4f ms  frac %.3f  verified %s status %d' % (d['ms_per_step'], d['roofline']['frac'], d['verified'], d['group_status']))"
done; done
