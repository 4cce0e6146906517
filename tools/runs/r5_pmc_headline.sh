#!/bin/bash
# round 5: the two PMC traffic passes of the headline alone (FETCH_SIZE, WRITE_SIZE), then the harness forwards
O=$PWD/gpurun_out/r5_pmc; rm -rf $O; mkdir -p $O; R=$PWD
B=$(python -c "import bench, torch; print(bench.box_id(0))")
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --batch 512 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $O/line.json 2> $O/kt.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python $R/bench.py --batch 512 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --sustained-secs 0 > $O/pmc_$c.json 2> $O/pmc_$c.err
done
cd $R
python tools/rocprof_headline.py $B $(find $O/kt -name "*kernel_stats.csv" | head -1) --pmc $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) --out $O --round r05
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/r05_pmc_traffic.json
bash tools/runs/r5_harness.sh
