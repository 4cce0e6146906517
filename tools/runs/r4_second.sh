#!/bin/bash
# round 4, second call: ablations of the small layers (k_mmq_group), write-only ablation of k_mmq_flat, the K sweep of the
# small layers, phase timelines (with the drain stamp) of the normal and the meeting-less build, the in-place probe.
O=$PWD/gpurun_out/r4_second; mkdir -p $O
R=$PWD
SMALL=1024x14,512x14,2048x7,256x14,512x7
for a in 0 1 2 3; do
  L=$R/tools/alt/libcnnq_abl$a.so; [ $a = 0 ] && L=$R/cnn_quantization_amd/libcnnq_hip.so
  CNNQ_HIP_LIB=$L timeout 300 python tools/bench_group.py --rounds 1 --reps 8 --shapes $SMALL > $O/small_abl_$a.log 2>&1
done
for a in 4 6; do
  CNNQ_HIP_LIB=$R/tools/alt/libcnnq_abl$a.so timeout 300 python tools/bench_group.py --rounds 1 --reps 8 --shapes 64x112,256x56,512x28 > $O/flat_abl_$a.log 2>&1
done
for k in 4 8 16 32; do
  CNNQ_GRP_K=$k timeout 300 python tools/bench_group.py --rounds 1 --reps 8 --shapes $SMALL > $O/small_K$k.log 2>&1
done
for k in 8 16; do
  CNNQ_GRP_K=$k timeout 300 python tools/bench_group.py --rounds 1 --reps 8 --shapes 64x112,256x56,128x56,512x28,64x56,256x28,128x28 > $O/flat_K$k.log 2>&1
done
CNNQ_HIP_LIB=$R/tools/alt/libcnnq_trace0.so timeout 300 python tools/trace_group.py --shapes 256x56,64x112,1024x14,256x14,2048x7,512x7 --save $O/tr0 > $O/trace0.log 2>&1
CNNQ_HIP_LIB=$R/tools/alt/libcnnq_trace2.so timeout 300 python tools/trace_group.py --shapes 256x56,64x112,1024x14,256x14,2048x7 --save $O/tr2 > $O/trace2.log 2>&1
timeout 200 python tools/alias_probe.py > $O/alias.log 2>&1
for a in 0 1 2 3; do echo "== small layers FLAT_ABL=$a"; grep "^C=" $O/small_abl_$a.log | cut -c1-75,118-160; done
for a in 4 6; do echo "== flat layers FLAT_ABL=$a"; grep "^C=" $O/flat_abl_$a.log | cut -c1-75,118-160; done
for k in 4 8 16 32; do echo "== small layers K=$k"; grep "^C=" $O/small_K$k.log | cut -c1-75,100-175; done
for k in 8 16; do echo "== flat layers K=$k"; grep "^C=" $O/flat_K$k.log | cut -c1-75,100-175; done
cat $O/alias.log
cat $O/trace0.log | grep -v "^   t(us)\|^   [0-9]" | head -120
