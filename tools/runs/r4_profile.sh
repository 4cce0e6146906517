#!/bin/bash
# round 4 profiles, ALL from one box and one call: the bench line (plain), rocprofv3 kernel trace + stats of the same
# command (b512 and the b64 shard), the two PMC traffic passes, the ceiling report (tools/ceiling_report.py) and the
# phase census of the flat kernel.  Everything lands in gpurun_out/r4_profile/ with the box id in the file names' header.
O=$PWD/gpurun_out/r4_profile; rm -rf $O; mkdir -p $O
R=$PWD
timeout 600 python bench.py > $O/r04_bench_line_b512.json 2> $O/bench.err
B=$(python -c "import json; print(json.load(open('$O/r04_bench_line_b512.json'))['box'])")
echo $B > $O/box.txt
cd /tmp && export TMPDIR=/tmp
for BT in 512 64; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b$BT -o kt -- python $R/bench.py --batch $BT --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $O/r04_bench_line_b${BT}_under_rocprof.json 2> $O/kt_b$BT.err
  f=$(find $O/kt_b$BT -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r04_bench_b${BT}_kernel_stats.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python $R/bench.py --batch 512 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $O/pmc_$c.json 2> $O/pmc_$c.err
done
cd $R
python tools/rocprof_headline.py $B $O/r04_bench_b512_kernel_stats.csv $O/r04_bench_b64_kernel_stats.csv --pmc $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) --out $O
timeout 900 python tools/ceiling_report.py > $O/r04_ceiling_$B.md 2> $O/ceiling.err
CNNQ_HIP_LIB=$R/tools/alt/libcnnq_trace0.so timeout 200 python tools/trace_group.py --shapes 256x56,64x112,1024x14,256x14 --save $O/tr > $O/r04_phase_timeline.log 2>&1
python tools/trace_census.py $O/tr/*.npz > $O/r04_phase_census.md 2>&1
rm -rf $O/tr
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
python tools/kstats.py $O/r04_bench_b512_kernel_stats.csv | head; cat $O/r04_rocprof_headline.json; cat $O/r04_pmc_traffic.json
python -c "
import json
for f in ('r04_bench_line_b512.json', 'r04_bench_line_b512_under_rocprof.json'):
    d = json.load(open('$O/' + f)); r = d['roofline']
    print(f, d['box'], '%.3f ms' % d['ms_per_step'], '%.1f G elem/s' % (d['value'] / 1e9), 'frac live %.3f' % r['frac'], 'avg launch %.1f us' % (r['avg_launch_ms'] * 1e3), {k: round(v['frac'], 3) for k, v in d['roofline_other_kernels'].items()}, d['verified'], d['group_status'])
d = json.load(open('$O/r04_bench_line_b512.json'))
print({k: (round(v['roofline']['frac'], 3), round(v['ms'], 3), v['verified']) for k, v in d.get('other_configs', {}).items()})
print(d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('cores'))
"
