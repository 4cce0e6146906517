#!/bin/bash
O=gpurun_out/r2l; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_holes_gpu.py tests/test_hip_parity.py -q > $O/pytest.log 2>&1; tail -n 12 $O/pytest.log
