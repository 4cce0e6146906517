#!/bin/bash
O=gpurun_out/r2j; mkdir -p $O
CNNQ_HIP_LIB=$PWD/tools/alt/libcnnq_cheap.so timeout 300 python tools/bench_group.py --batch 512 --reps 6 --rounds 1 > $O/b512_cheap.log 2>&1; cut -c1-215 $O/b512_cheap.log
