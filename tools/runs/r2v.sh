#!/bin/bash
O=gpurun_out/r2v; mkdir -p $O
for B in 64 512; do
timeout 600 python bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --force-exchange > $O/bench_forced_b$B.json 2> $O/bench_forced_b$B.err
CNNQ_DIRECT_RCCL=0 timeout 600 python bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --force-exchange > $O/bench_forced_torch_b$B.json 2> $O/bench_forced_torch_b$B.err
python -c "
import json
for f in ('bench_forced_b$B','bench_forced_torch_b$B'):
    d=json.load(open('$O/%s.json'%f)); print(f, round(d['ms_per_step'],3), d['verified'], d['config']['exchange'])"
tail -n 3 $O/bench_forced_b$B.err
done
