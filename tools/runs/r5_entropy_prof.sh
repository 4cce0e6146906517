#!/bin/bash
# round 5: kernel statistics of config 2 with the entropy (which part of the 0.7 ms is the entropy launch, which the counting)
O=$PWD/gpurun_out/r5_ent; rm -rf $O; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/bench_entropy.py > $O/log.txt 2> $O/err.txt
cd $R; python tools/kstats.py $(find $O/kt -name "*kernel_stats.csv" | head -1) | head -30
find $O -name "*kernel_trace.csv" -delete
