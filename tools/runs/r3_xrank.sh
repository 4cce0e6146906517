#!/bin/bash
# the in-launch cross-rank exchange (CNNQ_XRANK=1): the 1-rank and 2-ranks-on-one-GPU tests, then the forced-exchange
# b64 / b512 steps through the collective path and through the in-launch exchange on one box
O=$PWD/gpurun_out/r3_xrank; mkdir -p $O
timeout 600 python -m pytest tests/test_xrank_gpu.py tests/test_distributed_gpu.py -x -q > $O/pytest.log 2>&1; echo rc=$?; grep -v amdgpu.ids $O/pytest.log | tail -15
for xr in 0 1; do for b in 64 512; do
  CNNQ_XRANK=$xr timeout 300 python bench.py --force-exchange --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/b${b}_x$xr.json 2> $O/b${b}_x$xr.err
  python -c "import json;d=json.load(open('$O/b${b}_x$xr.json'));print('forced exchange, batch $b, CNNQ_XRANK=$xr: %.3f ms  %.1f G elem/s  verified %s  rccl_ranks %s  |' % (d['ms_per_step'], d['value']/1e9, d['verified'], d.get('rccl_ranks')), d['config']['exchange'][:60])" || tail -5 $O/b${b}_x$xr.err
done; done
