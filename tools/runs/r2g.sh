#!/bin/bash
O=gpurun_out/r2g; mkdir -p $O
timeout 900 python -m pytest tests/test_group_gpu.py tests/test_resident_gpu.py -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 12 $O/pytest.log
timeout 300 python tools/bench_group.py --batch 64 --shapes 64x112,256x56,128x56,512x28,64x56 > $O/group_b64.log 2>&1
timeout 600 python tools/bench_group.py --batch 512 --reps 6 --rounds 2 > $O/group_b512.log 2>&1
CNNQ_HIP_LIB=$PWD/tools/alt/libcnnq_noacq.so timeout 600 python tools/bench_group.py --batch 512 --reps 6 --rounds 2 > $O/group_b512_noacq.log 2>&1; CNNQ_HIP_LIB=$PWD/tools/alt/libcnnq_noacq.so timeout 900 python -m pytest tests/test_group_gpu.py -q > $O/pytest_noacq.log 2>&1; tail -n 3 $O/pytest_noacq.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench_b512.json 2> $O/bench_b512.err
timeout 600 python bench.py --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_b64.json 2> $O/bench_b64.err
cut -c1-250 $O/group_b64.log $O/group_b512.log; tail -n 3 $O/group_b512_noacq.log
for f in bench_b512 bench_b64; do python -c "
import json,sys
d=json.load(open('$O/$f.json')); print('$f', d['ms_per_step'], d['value']/1e9, d['path_frac_hbm_peak'], d['verified'], d['roofline']['kernel'][:12], d['roofline']['frac'])"; tail -n 2 $O/$f.err; done
