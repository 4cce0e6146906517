#!/bin/bash
# round 6: config 4 (the seven statistics) at batch 512 after k_stats_flat stopped waiting for its second meeting: tests, the
# whole-forward time, per-layer times against the chain
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_stats_single_gpu.py tests/test_sharded_single_gpu.py -q -m gpu -x 2>&1 | tail -3
python tools/bench_stats4.py 2>&1 | tail -40
