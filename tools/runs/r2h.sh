#!/bin/bash
O=gpurun_out/r2h; mkdir -p $O
timeout 900 python -m pytest tests/test_group_gpu.py -q -x > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
timeout 300 python tools/bench_group.py --batch 512 --reps 6 --rounds 1 > $O/b512.log 2>&1; tail -n 2 $O/b512.log
timeout 300 python tools/bench_group.py --batch 64 --shapes 64x112,256x56,128x56,64x56 > $O/b64.log 2>&1; tail -n 2 $O/b64.log
