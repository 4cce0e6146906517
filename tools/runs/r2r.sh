#!/bin/bash
O=gpurun_out/r2r; mkdir -p $O
CNNQ_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --batch 64 --steps 5 --warmup 2 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
tail -n 3 $O/bench_2rank_gloo.err; cut -c1-900 $O/bench_2rank_gloo.json
timeout 600 python bench.py --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --force-exchange > $O/bench_forced.json 2> $O/bench_forced.err; cut -c1-300 $O/bench_forced.json
