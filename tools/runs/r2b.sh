#!/bin/bash
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_resident_gpu.py -q > $O/pytest_resident.log 2>&1; echo "pytest rc=$?" >> $O/pytest_resident.log
tail -25 $O/pytest_resident.log
timeout 300 python tools/debug_resident.py 64 > $O/debug_b64_new.log 2>&1
CNNQ_HIP_LIB=$PWD/tools/alt/libcnnq_v0.so timeout 300 python tools/debug_resident.py 64 > $O/debug_b64_v0.log 2>&1
timeout 300 python tools/bench_resident.py --batch 64 > $O/layers_b64_auto.log 2>&1
CNNQ_RES_K=32 timeout 300 python tools/bench_resident.py --batch 64 > $O/layers_b64_K32.log 2>&1
timeout 300 python tools/bench_resident.py --batch 512 --reps 10 > $O/layers_b512_auto.log 2>&1
grep -c "y_mismatch=0 " $O/debug_b64_new.log; grep -vc "y_mismatch=0 " $O/debug_b64_new.log
