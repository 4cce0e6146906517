#!/bin/bash
# round 5: rocprofv3 kernel statistics of config 3 through the single launch (and through the chain), one box
O=$PWD/gpurun_out/r5/aciq_prof; rm -rf $O; mkdir -p $O
R=$PWD
python -c "import torch;print(torch.cuda.get_device_properties(0))" > $O/box.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for m in single chain; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$m -o kt -- python $R/tools/bench_aciq.py --only $m > $O/$m.log 2> $O/$m.err
  tail -n 3 $O/$m.log
  f=$(find $O/kt_$m -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_$m.csv
  python $R/tools/kstats.py $f | head -14
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
