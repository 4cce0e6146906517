#!/bin/bash
# round 5: the single-read statistics kernel on row-piece tiles (k_stats_group): tests, then config 4 chain vs single per layer
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_stats_single_gpu.py -q -m gpu -x 2>&1 | tail -15
timeout 600 python tools/bench_stats4.py 2>&1 | tee gpurun_out/r5/stats4_group.log | tail -30
