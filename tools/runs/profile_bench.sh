#!/bin/bash
# round 2 profiles: rocprofv3 kernel trace + stats and the two PMC passes, of the driver's bench command (b512)
# and of the batch-64 shard; summaries go to gpurun_out/profile_bench/ (copied into profiles/ by hand)
O=$PWD/gpurun_out/profile_bench; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 512 64; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b$B -o kt -- python $R/bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $O/kt_b$B.json 2> $O/kt_b$B.err
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_b$B -o pmc -- python $R/bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $O/pmc_${c}_b$B.json 2> $O/pmc_${c}_b$B.err
  done
done
cd $R
find $O -name "*kernel_stats.csv" | head; find $O -name "*counter_collection.csv" | head
for B in 512 64; do
  python tools/summarize_prof.py $(find $O/kt_b$B -name "*kernel_stats.csv" | head -1) > $O/kernel_stats_b$B.md
  python tools/summarize_pmc.py $(find $O/pmc_FETCH_SIZE_b$B -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE_b$B -name "*counter_collection.csv" | head -1) > $O/pmc_b$B.md
  cat $O/kernel_stats_b$B.md $O/pmc_b$B.md
done
cp $(find $O/kt_b512 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_b512.csv
cp $(find $O/kt_b64 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_b64.csv
# keep the merged output small: drop the raw traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
du -sh $O
