#!/bin/bash
O=$PWD/gpurun_out/r3_misc; mkdir -p $O
python -m pytest tests/test_bench_contract_gpu.py tests/test_concurrent_gpu.py tests/test_midtread_hist_gpu.py -x -q -s > $O/pytest.log 2>&1; grep -v "amdgpu.ids" $O/pytest.log | tail -25
