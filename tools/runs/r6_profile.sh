#!/bin/bash
# round 6 profiles, ALL from one box and one call: the bench line (plain, with other_configs and the CPU baseline), rocprofv3
# kernel trace + stats of the same command (b512 and the b64 shard), the two PMC traffic passes of the headline, and the kernel
# statistics + PMC traffic of the single-launch forms of configs 3 / 4 / 5.  Everything lands in gpurun_out/r6_profile/.
O=$PWD/gpurun_out/r6_profile; rm -rf $O; mkdir -p $O
R=$PWD
timeout 900 python bench.py > $O/r06_bench_line_b512.json 2> $O/bench.err
B=$(python -c "import json; print(json.load(open('$O/r06_bench_line_b512.json'))['box'])")
echo $B > $O/box.txt
cd /tmp && export TMPDIR=/tmp
for BT in 512 64; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b$BT -o kt -- python $R/bench.py --batch $BT --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $O/r06_bench_line_b${BT}_under_rocprof.json 2> $O/kt_b$BT.err
  f=$(find $O/kt_b$BT -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r06_bench_b${BT}_kernel_stats.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python $R/bench.py --batch 512 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --sustained-secs 0 > $O/pmc_$c.json 2> $O/pmc_$c.err
done
# configs 3 / 5 / 4 through their single launches
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -o kt -- python $R/tools/bench_aciq.py --only single > $O/c3.log 2> $O/c3.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c5 -o kt -- python $R/tools/bench_aciq.py --vgg --only single > $O/c5.log 2> $O/c5.err
ONLY=single timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4 -o kt -- python $R/tools/bench_stats4.py > $O/c4.log 2> $O/c4.err
for c in 3 4 5; do f=$(find $O/kt_c$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r06_config${c}_kernel_stats.csv; done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc3_$c -o pmc -- python $R/tools/bench_aciq.py --only single > $O/pmc3_$c.log 2> $O/pmc3_$c.err
  ONLY=single timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc4_$c -o pmc -- python $R/tools/bench_stats4.py > $O/pmc4_$c.log 2> $O/pmc4_$c.err
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc5_$c -o pmc -- python $R/tools/bench_aciq.py --vgg --only single > $O/pmc5_$c.log 2> $O/pmc5_$c.err
done
cd $R
python tools/rocprof_headline.py $B $O/r06_bench_b512_kernel_stats.csv $O/r06_bench_b64_kernel_stats.csv --pmc $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) --out $O --round r06
mkdir -p $O/c3 $O/c4 $O/c5
python tools/rocprof_headline.py $B $O/r06_config5_kernel_stats.csv --pmc $(find $O/pmc5_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc5_WRITE_SIZE -name "*counter_collection.csv" | head -1) --out $O/c5 --round r06 > /dev/null
cp $O/c5/r06_pmc_traffic.json $O/r06_pmc_traffic_config5.json; cp $O/c5/r06_rocprof_headline.json $O/r06_rocprof_config5.json
python tools/rocprof_headline.py $B $O/r06_config3_kernel_stats.csv --pmc $(find $O/pmc3_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc3_WRITE_SIZE -name "*counter_collection.csv" | head -1) --out $O/c3 --round r06 > /dev/null
python tools/rocprof_headline.py $B $O/r06_config4_kernel_stats.csv --pmc $(find $O/pmc4_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc4_WRITE_SIZE -name "*counter_collection.csv" | head -1) --out $O/c4 --round r06 > /dev/null
cp $O/c3/r06_pmc_traffic.json $O/r06_pmc_traffic_config3.json; cp $O/c3/r06_rocprof_headline.json $O/r06_rocprof_config3.json
cp $O/c4/r06_pmc_traffic.json $O/r06_pmc_traffic_config4.json; cp $O/c4/r06_rocprof_headline.json $O/r06_rocprof_config4.json
# the 64-sample shard: configs 2 - 5 through the forced 1-rank exchange (in-launch, collective) and without exchange
for x in 1 0; do
  CNNQ_XRANK=$x timeout 300 python bench.py --batch 64 --force-exchange --no-cpu-baseline --steps 40 --warmup 5 > $O/r06_bench_line_b64_force_exchange_xr$x.json 2> $O/b64_xr$x.err
done
timeout 300 python bench.py --batch 64 --no-cpu-baseline --steps 40 --warmup 5 > $O/r06_bench_line_b64.json 2> $O/b64.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
for c in 3 4 5; do echo "== config $c"; tail -n 2 $O/c$c.log; python tools/kstats.py $O/r06_config${c}_kernel_stats.csv | head -12; done
python tools/kstats.py $O/r06_bench_b512_kernel_stats.csv | head; cat $O/r06_rocprof_headline.json; cat $O/r06_pmc_traffic.json; cat $O/r06_pmc_traffic_config3.json $O/r06_pmc_traffic_config4.json
python -c "
import json
for f in ('r06_bench_line_b512.json', 'r06_bench_line_b512_under_rocprof.json', 'r06_bench_line_b64_under_rocprof.json'):
    d = json.load(open('$O/' + f)); r = d['roofline']
    print(f, d['box'], '%.3f ms' % d['ms_per_step'], 'sustained %.3f' % d['sustained']['ms_per_step'], '%.1f G elem/s' % (d['value'] / 1e9), 'frac live %.3f' % r['frac'], 'avg launch %.1f us' % (r['avg_launch_ms'] * 1e3), {k: round(v['frac'], 3) for k, v in d['roofline_other_kernels'].items()}, d['verified'], d['group_status'])
d = json.load(open('$O/r06_bench_line_b512.json'))
print({k: (round(v['roofline']['frac'], 3), round(v['ms'], 3), v['verified']) for k, v in d.get('other_configs', {}).items()})
print(d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('cores'))
for f in ('r06_bench_line_b64_force_exchange_xr1.json', 'r06_bench_line_b64_force_exchange_xr0.json', 'r06_bench_line_b64.json'):
    d = json.load(open('$O/' + f))
    print(f, '%.3f ms' % d['ms_per_step'], {k: (round(v['ms'], 3), v['verified'], round(v['roofline']['bytes_moved_per_element'], 2)) for k, v in d['other_configs'].items() if k in ('config3', 'config4', 'config5')})
"
