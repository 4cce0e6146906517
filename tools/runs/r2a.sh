#!/bin/bash
# round 2, GPU call A: correctness of the resident kernel, per-layer sweep, whole-workload bench at b64 / b512
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_resident_gpu.py -x -q > $O/pytest_resident.log 2>&1; echo "pytest rc=$?" >> $O/pytest_resident.log
tail -5 $O/pytest_resident.log
for K in 0 8 16 32; do
  CNNQ_RES_K=$K timeout 300 python tools/bench_resident.py --batch 64 > $O/layers_b64_K$K.log 2>&1
done
CNNQ_RES_WGS=2048 timeout 300 python tools/bench_resident.py --batch 64 > $O/layers_b64_T2048.log 2>&1
CNNQ_RES_WGS=512 timeout 300 python tools/bench_resident.py --batch 64 > $O/layers_b64_T512.log 2>&1
timeout 300 python tools/bench_resident.py --batch 512 --reps 10 > $O/layers_b512_auto.log 2>&1
CNNQ_RES_K=16 timeout 300 python tools/bench_resident.py --batch 512 --reps 10 > $O/layers_b512_K16.log 2>&1
CNNQ_HIP_LIB=$PWD/tools/alt/libcnnq_k32w2.so timeout 300 python tools/bench_resident.py --batch 512 --reps 10 > $O/layers_b512_w2.log 2>&1
CNNQ_HIP_LIB=$PWD/tools/alt/libcnnq_k32w2.so CNNQ_RES_K=32 timeout 300 python tools/bench_resident.py --batch 64 > $O/layers_b64_K32_w2.log 2>&1
for R in 0 1; do
  CNNQ_RESIDENT=$R timeout 300 python bench.py --batch 64 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_b64_res$R.json 2> $O/bench_b64_res$R.err
  CNNQ_RESIDENT=$R timeout 300 python bench.py --batch 512 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_b512_res$R.json 2> $O/bench_b512_res$R.err
done
tail -3 $O/layers_b64_K0.log $O/layers_b512_auto.log
cat $O/bench_b64_res1.json | cut -c1-400
