#!/bin/bash
O=gpurun_out/r2f; mkdir -p $O
timeout 300 python tools/bench_group.py --batch 64 > $O/group_b64.log 2>&1
CNNQ_GRP_K=16 timeout 300 python tools/bench_group.py --batch 64 --shapes 64x112,256x56,128x56,64x56 > $O/group_b64_K16.log 2>&1
timeout 600 python tools/bench_group.py --batch 512 --reps 6 --rounds 2 > $O/group_b512.log 2>&1
CNNQ_GRP_K=16 timeout 600 python tools/bench_group.py --batch 512 --reps 6 --rounds 2 --shapes 64x112,256x56,512x28,1024x14 > $O/group_b512_K16.log 2>&1
cut -c1-250 $O/group_b64.log $O/group_b512.log
