#!/bin/bash
# round 6: where the shard legs of configs 3 / 4 / 5 spend their time: wall clock and host time per call for the three routes, and
# rocprofv3 kernel statistics of the in-launch route
O=$PWD/gpurun_out/r6/shard_prof; rm -rf $O; mkdir -p $O
R=$PWD
for c in 3 4 5; do
  python tools/bench_shard_cfg.py --config $c --plain 2>/dev/null | tail -1
  python tools/bench_shard_cfg.py --config $c --xrank 1 2>/dev/null | grep config
  python tools/bench_shard_cfg.py --config $c --xrank 0 2>/dev/null | grep config
done | tee $O/walls.txt
cd /tmp && export TMPDIR=/tmp
for c in 3 4; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$c -o kt -- python $R/tools/bench_shard_cfg.py --config $c --xrank 1 > $O/c$c.log 2> $O/c$c.err
  f=$(find $O/kt_$c -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_c$c.csv
  python $R/tools/kstats.py $f | head -16
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
