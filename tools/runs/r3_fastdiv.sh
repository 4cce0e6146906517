#!/bin/bash
# the divide-free exact quotient: edge-case parity, the suites of the three single-launch kernels, then an A/B of the
# b512 and b64 steps with the hardware divide forced (CNNQ_IEEE_DIVIDE=1) and not, interleaved on one box
O=$PWD/gpurun_out/r3_fastdiv; mkdir -p $O
timeout 600 python -m pytest tests/test_fastdiv_gpu.py tests/test_group_gpu.py tests/test_resident_gpu.py tests/test_single_outputs_gpu.py -x -q > $O/pytest.log 2>&1; echo rc=$?; grep -v amdgpu.ids $O/pytest.log | tail -5
for r in 1 2; do
  for d in 0 1; do
    CNNQ_IEEE_DIVIDE=$d timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/b512_d${d}_$r.json 2> $O/err.log
    python -c "import json;d=json.load(open('$O/b512_d${d}_$r.json'));r=d['roofline'];print('b512 ieee_divide=$d round $r: %.3f ms  %.1f G elem/s  %s frac %.4f  verified %s' % (d['ms_per_step'], d['value']/1e9, r['kernel'][:12], r['frac'], d['verified']), {k:round(v['frac'],3) for k,v in d['roofline_other_kernels'].items()})"
  done
done
for d in 0 1; do
  CNNQ_IEEE_DIVIDE=$d timeout 200 python bench.py --batch 64 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs > $O/b64_d${d}.json 2>> $O/err.log
  python -c "import json;d=json.load(open('$O/b64_d${d}.json'));r=d['roofline'];print('b64 ieee_divide=$d: %.3f ms  %.1f G elem/s  %s frac %.4f' % (d['ms_per_step'], d['value']/1e9, r['kernel'][:12], r['frac']), {k:round(v['frac'],3) for k,v in d['roofline_other_kernels'].items()})"
done
