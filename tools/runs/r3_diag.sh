#!/bin/bash
# round 3, first diagnostic call: (1) workgroup-structure microbenchmark, (2) phase timeline of k_mmq_group,
# (3) SQ / TCC counters of k_mmq_group on four layer shapes.  Outputs under gpurun_out/r3_diag/.
O=$PWD/gpurun_out/r3_diag; mkdir -p $O
R=$PWD
python tools/ubench_pipe.py > $O/ubench_pipe.log 2>&1
python tools/ubench_tile.py > $O/ubench_tile.log 2>&1
CNNQ_HIP_LIB=$R/tools/libcnnq_trace.so python tools/trace_group.py > $O/trace.log 2>&1
cd /tmp && export TMPDIR=/tmp
SH="256x56,64x112,512x28,1024x14"
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACCUM_PREV" \
           "TCC_BUSY_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc$i -o pmc -- python $R/tools/bench_group.py --shapes $SH --rounds 1 --reps 2 > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/summarize_counters.py $f > $O/pmc$i.md 2>&1
  find $O/pmc$i -name "*.csv" -size +5M -delete
done
cd $R
cat $O/ubench_pipe.log $O/trace.log $O/pmc*.md | tail -250
