#!/bin/bash
O=gpurun_out/r2n; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; tail -n 4 $O/pytest_all.log
CFGS=3,4 timeout 600 python tools/bench_configs.py > $O/configs.log 2>&1; grep -v amdgpu $O/configs.log
