#!/bin/bash
# round 4: kernel-level breakdown of configs 1 / 3 / 4 with the final library (rocprofv3 --kernel-trace --stats of tools/bench_configs.py)
O=$PWD/gpurun_out/r4_cfg_prof; rm -rf $O; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in 1 3 4; do
  CFGS=$c timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c$c -o kt -- python $R/tools/bench_configs.py > $O/c$c.log 2> $O/c$c.err
  tail -n 4 $O/c$c.log
  f=$(find $O/kt_c$c -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_c$c.csv
  python $R/tools/kstats.py $f | head -12
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
