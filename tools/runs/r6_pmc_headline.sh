#!/bin/bash
# round 6: the two PMC traffic passes of the headline alone (FETCH_SIZE, WRITE_SIZE; without the 2 s sustained loop, which
# outlives the time limit under counter collection), an A/B of the non-temporal-load threshold of the statistics passes
# (CNNQ_NT_BYTES: the one environment variable the library reads), and config 3 per layer
O=$PWD/gpurun_out/r6_pmc; rm -rf $O; mkdir -p $O; R=$PWD
B=$(python -c "import bench, torch; print(bench.box_id(0))")
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --batch 512 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --sustained-secs 0 > $O/line.json 2> $O/kt.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python $R/bench.py --batch 512 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --sustained-secs 0 > $O/pmc_$c.json 2> $O/pmc_$c.err
done
cd $R
python tools/rocprof_headline.py $B $(find $O/kt -name "*kernel_stats.csv" | head -1) --pmc $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) --out $O --round r06
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/r06_pmc_traffic.json
for nt in default 0; do
  echo "== CNNQ_NT_BYTES=$nt"
  if [ $nt = default ]; then E=""; else E="CNNQ_NT_BYTES=$nt"; fi
  env $E python tools/bench_aciq.py --only single 2>&1 | grep "config 3"
  env $E ONLY=single python tools/bench_stats4.py 2>&1 | grep "config 4"
  env $E python tools/bench_aciq.py --vgg --only single 2>&1 | grep "config 5"
done
python tools/bench_aciq.py --layers 2>&1 | tail -16
