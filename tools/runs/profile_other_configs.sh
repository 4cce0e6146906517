#!/bin/bash
# kernel-level breakdown of configs 3 / 4 / 5 (tools/bench_configs.py) under rocprofv3
O=$PWD/gpurun_out/profile_other_configs; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in 3 4 5; do
  CFGS=$c timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c$c -o kt -- python $R/tools/bench_configs.py > $O/c$c.log 2> $O/c$c.err
  tail -n 4 $O/c$c.log
  f=$(find $O/kt_c$c -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_c$c.csv
  python $R/tools/summarize_prof.py $f | head -20
done
find $O -name "*kernel_trace.csv" -delete
du -sh $O
