#!/bin/bash
O=gpurun_out/r2t; mkdir -p $O
timeout 600 python -m pytest tests/test_distributed_gpu.py -q > $O/pytest_dist.log 2>&1; tail -n 2 $O/pytest_dist.log
for B in 64 512; do
timeout 600 python bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --force-exchange > $O/bench_forced_b$B.json 2> $O/bench_forced_b$B.err
python -c "
import json; d=json.load(open('$O/bench_forced_b$B.json')); print('forced b$B', d['ms_per_step'], d['verified'], {k:round(v['time_per_step_ms'],3) for k,v in [(d['roofline']['kernel'][:8],d['roofline'])]+list(d['roofline_other_kernels'].items())})"
CNNQ_RESIDENT=0 timeout 600 python bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench_chain_b$B.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench_chain_b$B.json')); print('chain  b$B', d['ms_per_step'])"
done
CNNQ_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --batch 64 --steps 5 --warmup 2 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; python -c "
import json; d=json.load(open('$O/bench_2rank_gloo.json')); print('2-rank gloo b64', d['ms_per_step'], d['verified'])"
