#!/bin/bash
# round 5: the single-launch ACIQ kernels - parity, then chain vs single launch at b512 (whole forward and per layer)
mkdir -p gpurun_out/r5
cd "$GRAFT_REPO_ROOT"
python -c "import torch;print(torch.cuda.get_device_name(0))" > gpurun_out/r5/aciq_box.txt 2>&1
timeout 900 python -m pytest tests/test_aciq_single_gpu.py -x -q -m gpu > gpurun_out/r5/aciq_tests.log 2>&1
tail -15 gpurun_out/r5/aciq_tests.log
timeout 600 python tools/bench_aciq.py --layers > gpurun_out/r5/aciq_bench.log 2>&1
cat gpurun_out/r5/aciq_bench.log
timeout 600 python -m pytest tests/test_full_size_gpu.py tests/test_hip_parity.py -x -q -m gpu > gpurun_out/r5/aciq_tests2.log 2>&1
tail -5 gpurun_out/r5/aciq_tests2.log
