#!/bin/bash
# HBM traffic of the kernels of configs 3 / 4 / 5 (tools/bench_configs.py): two rocprofv3 PMC passes per configuration
O=$PWD/gpurun_out/profile_other_configs_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in 3 4 5; do
  for k in FETCH_SIZE WRITE_SIZE; do
    CFGS=$c timeout 600 rocprofv3 --kernel-trace --pmc $k --output-format csv -d $O/pmc_${k}_c$c -o pmc -- python $R/tools/bench_configs.py > $O/c${c}_$k.log 2> $O/c${c}_$k.err
  done
  echo "## config $c"
  python $R/tools/summarize_pmc.py $(find $O/pmc_FETCH_SIZE_c$c -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE_c$c -name "*counter_collection.csv" | head -1) | tee $O/pmc_c$c.md
done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
du -sh $O
