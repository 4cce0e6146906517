#!/bin/bash
# round 4 (final library): the forced 1-rank exchange - b64 / b512 steps through the collective path (CNNQ_XRANK=0) and through the
# in-launch exchange (CNNQ_XRANK=1) on one box, and the plain single-GPU steps next to them
O=$PWD/gpurun_out/r4_xrank; mkdir -p $O
for b in 64 512; do
  timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/b${b}_plain.json 2> $O/b${b}_plain.err
  python -c "import json;d=json.load(open('$O/b${b}_plain.json'));print('single GPU, no exchange, batch $b: %.3f ms  %.1f G elem/s  verified %s' % (d['ms_per_step'], d['value']/1e9, d['verified']))"
  for xr in 0 1; do
  CNNQ_XRANK=$xr timeout 300 python bench.py --force-exchange --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/b${b}_x$xr.json 2> $O/b${b}_x$xr.err
  python -c "import json;d=json.load(open('$O/b${b}_x$xr.json'));print('forced exchange, batch $b, CNNQ_XRANK=$xr: %.3f ms  %.1f G elem/s  verified %s  rccl_ranks %s  xrank %s |' % (d['ms_per_step'], d['value']/1e9, d['verified'], d.get('rccl_ranks'), d.get('xrank')), d['config']['exchange'][:60])" || tail -5 $O/b${b}_x$xr.err
done; done
