#!/bin/bash
# round 5: the 64-sample shard (the per-rank step of the 8-GPU run) eager and as a HIP graph, without and with the exchange forced
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5; mkdir -p $O
for v in "" "--graph" "--force-exchange" "--force-exchange --graph" "--force-exchange --xr" "--force-exchange --graph --xr"; do
  n=$(echo "b64$v" | tr -d ' ' | tr -s '-' '_')
  X=0; case "$v" in *--xr*) X=1; v=${v/ --xr/};; esac        # --xr: the in-launch exchange (CNNQ_XRANK=1) instead of the collective
  CNNQ_XRANK=$X timeout 300 python bench.py --batch 64 --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs $v > $O/r05_bench_line_$n.json 2> $O/$n.err
  python -c "
import json
d=json.loads(open('$O/r05_bench_line_$n.json').read().strip().splitlines()[-1])
print('%-34s' % '$v', d['box'], '%.3f ms' % d['ms_per_step'], 'sustained %.3f' % d['sustained']['ms_per_step'], d['verified'], d['group_status'], d['config']['exchange'][:60])
"
done
