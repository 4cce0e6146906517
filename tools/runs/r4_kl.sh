#!/bin/bash
# round 4: eight more steps per flat tile in LDS (CNNQ_FLAT_KL=8): parity of the flat-tile suites, then A/B of the step and of the packed form
O=$PWD/gpurun_out/r4_kl; mkdir -p $O
CNNQ_FLAT_KL=8 timeout 1200 python -m pytest tests/test_group_gpu.py tests/test_single_outputs_gpu.py tests/test_fastdiv_gpu.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for r in 1 2 3; do for kl in 0 8; do
  CNNQ_FLAT_KL=$kl python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('KL=$kl round $r: %.3f ms  frac %.3f  group %.3f  verified %s status %d' % (d['ms_per_step'], d['roofline']['frac'], list(d['roofline_other_kernels'].values())[0]['frac'], d['verified'], d['group_status']))"
done; done
for kl in 0 8 0 8; do echo "KL=$kl"; CNNQ_FLAT_KL=$kl python tools/bench_pack_single.py 2>&1 | tail -1; done
for kl in 0 8; do echo "KL=$kl"; CNNQ_FLAT_KL=$kl python tools/bench_group.py --rounds 1 --reps 10 --shapes 64x112,256x56,128x56,512x28,64x56,256x28,128x28 2>&1 | grep "^C=" | sed -E 's/\| A=.*Gs= *([0-9]+) wgs= *([0-9]+) \| chain +([0-9.]+) us.*group +([0-9.]+) us +([0-9]+) GB.*mismatches=([0-9]+).*/| Gs \1 wgs \2 group \4 us \5 GB\/s(8B) mismatches \6/'; done
