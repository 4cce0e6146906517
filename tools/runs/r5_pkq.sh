#!/bin/bash
# round 5: the packed quotient in every A = 1 single-launch instance - parity suites, then the steps and configs 3 / 5
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_group_gpu.py tests/test_resident_gpu.py tests/test_aciq_single_gpu.py tests/test_hip_parity.py tests/test_fastdiv_gpu.py tests/test_single_outputs_gpu.py tests/test_fuzz_gpu.py -q -m gpu -x 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b512', d['box'], '%.3f ms' % d['ms_per_step'], 'flat %.3f' % d['roofline']['frac'], {k: round(v['frac'], 3) for k, v in d['roofline_other_kernels'].items()})"
python bench.py --batch 64 --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b64', d['box'], '%.3f ms' % d['ms_per_step'], {k: round(v['frac'], 3) for k, v in d['roofline_other_kernels'].items()})"
python tools/bench_aciq.py --only single 2>&1 | grep "config 3"
