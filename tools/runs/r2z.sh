#!/bin/bash
O=gpurun_out/r2z; mkdir -p $O
timeout 600 python -m pytest tests/test_midtread_hist_gpu.py -q -x > $O/pytest1.log 2>&1; tail -n 25 $O/pytest1.log
timeout 600 python -m pytest tests -m gpu -q -x -k "mid or tread or entropy or cfg5 or vgg or mt or fuzz" > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log
CFGS=5 timeout 300 python tools/bench_configs.py 2>&1 | tail -n 2
