#!/bin/bash
# round 4: LDS rows as the packed output's default: the packed suites by default, the flat-tile suites with the knob forced on
O=$PWD/gpurun_out/r4_kl; mkdir -p $O
timeout 1200 python -m pytest tests/test_single_outputs_gpu.py tests/test_parity_holes_gpu.py tests/test_group_gpu.py -q > $O/pytest_default.log 2>&1; tail -3 $O/pytest_default.log
CNNQ_FLAT_KL=8 timeout 1500 python -m pytest tests/test_group_gpu.py tests/test_single_outputs_gpu.py tests/test_xrank_gpu.py tests/test_full_size_gpu.py tests/test_concurrent_gpu.py -q --deselect tests/test_group_gpu.py::test_group_plans_cover_one_and_two_level_arrival > $O/pytest_kl8.log 2>&1; tail -3 $O/pytest_kl8.log
for kl in 0 -1 0 -1; do echo "KL=$kl"; CNNQ_FLAT_KL=$kl python tools/bench_pack_single.py 2>&1 | tail -1; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > $O/bench.json; python - <<'PY'
import json; d=json.load(open('gpurun_out/r4_kl/bench.json')); print(d['ms_per_step'], d['roofline']['frac'], d['box'])
for k,v in d['other_configs'].items(): print(k, v['roofline']['frac'], v['verified'])
PY
