#!/bin/bash
O=gpurun_out/r2w; mkdir -p $O
CNNQ_GROUP_WS_CACHED=1 timeout 900 python -m pytest tests/test_group_gpu.py tests/test_parity_holes_gpu.py -q > $O/pytest_cached.log 2>&1; tail -n 3 $O/pytest_cached.log
CNNQ_GROUP_WS_CACHED=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench_cached.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_cached.json')); print('cached ws b512', d['ms_per_step'], d['verified'])"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench_fg.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_fg.json')); print('fine-grained ws b512', d['ms_per_step'], d['verified'])"
