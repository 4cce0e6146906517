#!/bin/bash
O=gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_resident_gpu.py -q -x > $O/pytest_resident.log 2>&1; echo "pytest rc=$?" >> $O/pytest_resident.log
tail -n 4 $O/pytest_resident.log
timeout 600 python bench.py --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_b64.json 2> $O/bench_b64.err
timeout 600 python bench.py --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --graph > $O/bench_b64_graph.json 2> $O/bench_b64_graph.err
timeout 600 python bench.py --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --force-exchange > $O/bench_b64_forced.json 2> $O/bench_b64_forced.err
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_b512_full.json 2> $O/bench_b512_full.err
timeout 300 python tools/host_overhead.py > $O/host_overhead.log 2>&1
for f in bench_b64 bench_b64_graph bench_b64_forced; do python -c "
import json,sys
d=json.load(open('$O/$f.json')); print('$f', d['ms_per_step'], d['value']/1e9, d['path_frac_hbm_peak'], d['verified'], d['config']['exchange'], d['roofline']['kernel'][:12], d['roofline']['frac'])"; done
tail -3 $O/*.err; cat $O/host_overhead.log
