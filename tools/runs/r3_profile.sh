#!/bin/bash
# round 3 profiles, ALL from one box: rocprofv3 kernel trace + stats of the driver's bench command (b512) and of the
# batch-64 shard, the two PMC traffic passes (FETCH_SIZE, WRITE_SIZE) at b512, SQ / TCC / TCP counter passes of the
# group kernels on four layer shapes, and the phase timeline of k_mmq_flat.  Summaries -> gpurun_out/r3_profile/.
O=$PWD/gpurun_out/r3_profile; mkdir -p $O
R=$PWD
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_line_b512_plain.json 2>/dev/null
CNNQ_HIP_LIB=$R/tools/libcnnq_trace.so timeout 200 python tools/trace_group.py --shapes 256x56,64x112,512x28,1024x14,256x14 > $O/trace_flat.log 2>&1
cd /tmp && export TMPDIR=/tmp
for B in 512 64; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b$B -o kt -- python $R/bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $O/bench_line_b${B}_under_rocprof.json 2> $O/kt_b$B.err
  f=$(find $O/kt_b$B -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_b${B}_kernel_stats.csv && python $R/tools/summarize_prof.py $f > $O/kernel_stats_b$B.md
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python $R/bench.py --batch 512 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $O/pmc_$c.json 2> $O/pmc_$c.err
done
python $R/tools/summarize_pmc.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) > $O/pmc_traffic_b512.md 2>&1
SH="256x56,64x112,512x28,1024x14"
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM" \
           "TCC_BUSY_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/cnt$i -o pmc -- python $R/tools/bench_group.py --shapes $SH --rounds 1 --reps 2 > $O/cnt$i.log 2>&1
  f=$(find $O/cnt$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/summarize_counters.py $f > $O/counters$i.md 2>&1
done
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +3M -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; cat $O/kernel_stats_b512.md | head -20; cat $O/pmc_traffic_b512.md; python -c "
import json;d=json.load(open('$O/bench_line_b512_plain.json'));print('plain', d['ms_per_step'], d['value']/1e9, d['roofline']['frac'], d['roofline']['avg_launch_ms'])
d=json.load(open('$O/bench_line_b512_under_rocprof.json'));print('rocprof', d['ms_per_step'], d['value']/1e9, d['roofline']['frac'])"
