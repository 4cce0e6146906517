#!/bin/bash
# round 3, second diagnostic call: the corrected structure microbenchmark (back-to-back loads, delay after the fold) and
# the start-phase stagger of k_mmq_group (CNNQ_GRP_STAGGER) on all layers, plus its timeline on two shapes.
O=$PWD/gpurun_out/r3_diag2; mkdir -p $O
R=$PWD
python tools/ubench_pipe.py 3 > $O/ubench_pipe.log 2>&1
python tools/bench_group.py --rounds 1 --reps 6 > $O/group_base.log 2>&1
for sg in 2,28 2,20 3,18 4,14 4,10 6,10; do
  CNNQ_GRP_STAGGER=$sg python tools/bench_group.py --rounds 1 --reps 6 > $O/group_stag_$sg.log 2>&1
done
CNNQ_GRP_STAGGER=4,14 CNNQ_HIP_LIB=$R/tools/libcnnq_trace.so python tools/trace_group.py --shapes 256x56,64x112,512x28 > $O/trace_stag.log 2>&1
cat $O/ubench_pipe.log
grep -h "per forward group\|^C=" $O/group_base.log | cut -c1-60,95-140
for f in $O/group_stag_*.log; do echo $f; grep -h "per forward group\|^C=" $f | cut -c1-60,95-140; done
