#!/bin/bash
O=gpurun_out/r2s; mkdir -p $O
timeout 600 python -m pytest tests/test_resident_gpu.py tests/test_group_gpu.py -q -x > $O/pytest.log 2>&1; tail -n 2 $O/pytest.log
timeout 300 python tools/bench_resident.py --batch 64 > $O/res_b64.log 2>&1; cut -c1-60,108-215 $O/res_b64.log
timeout 300 python bench.py --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_b64.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_b64.json')); print(d['ms_per_step'], d['path_frac_hbm_peak'], d['verified'], {k:(v['launches_per_step'], round(v['time_per_step_ms'],3)) for k,v in [('dom',d['roofline'])]+list(d['roofline_other_kernels'].items())})"
