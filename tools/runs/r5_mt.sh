#!/bin/bash
# round 5: the single-launch mid-tread kernels and the big-channel tiles - parity, then chain vs single launch on VGG-16 b512
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_midtread_single_gpu.py tests/test_aciq_single_gpu.py -x -q -m gpu > gpurun_out/r5/mt_tests.log 2>&1
tail -25 gpurun_out/r5/mt_tests.log
timeout 600 python tools/bench_aciq.py --vgg --layers > gpurun_out/r5/mt_bench.log 2>&1
grep -v amdgpu.ids gpurun_out/r5/mt_bench.log
