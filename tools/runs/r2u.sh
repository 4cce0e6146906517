#!/bin/bash
O=gpurun_out/r2u; mkdir -p $O
timeout 300 python tools/rccl_latency.py > $O/rccl_latency.log 2>&1; grep "C=" $O/rccl_latency.log
CNNQ_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 tools/p2p_latency.py > $O/p2p_latency.log 2>&1; grep "per all-gather\|healthy\|unavailable" $O/p2p_latency.log
