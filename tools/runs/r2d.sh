#!/bin/bash
O=gpurun_out/r2d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_resident_gpu.py -q > $O/pytest_resident.log 2>&1; echo "pytest rc=$?" >> $O/pytest_resident.log
tail -n 15 $O/pytest_resident.log
timeout 300 python tools/bench_resident.py --batch 64 > $O/layers_b64_auto.log 2>&1
CNNQ_RES_T=1024 timeout 300 python tools/bench_resident.py --batch 64 --shapes 512x28,256x28,128x28,1024x14,256x14 > $O/layers_b64_T1024.log 2>&1
CNNQ_RES_T=256 timeout 300 python tools/bench_resident.py --batch 64 --shapes 1024x14,256x14,2048x7,512x7 > $O/layers_b64_T256.log 2>&1
timeout 300 python tools/bench_resident.py --batch 8 > $O/layers_b8_auto.log 2>&1
for R in 0 1; do
  CNNQ_RESIDENT=$R timeout 300 python bench.py --batch 64 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_b64_res$R.json 2> $O/bench_b64_res$R.err
done
cat $O/layers_b64_auto.log | cut -c1-240
