#!/bin/bash
# round 4, first call: (1) the one-shot Infinity-Cache two-pass microbenchmark, (2) the chunked chain of the product
# kernels in a graph, (3) the ablation of k_mmq_flat (stores off / meeting off / both), (4) the box's bench line.
O=$PWD/gpurun_out/r4_first; mkdir -p $O
R=$PWD
rocm-smi --showserial --showbus > $O/box.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err
timeout 600 python tools/ubench_mall3.py > $O/mall3.log 2>&1
timeout 600 python tools/chunk_chain.py > $O/chunk_chain.log 2>&1
for a in 0 1 2 3; do
  L=$R/tools/alt/libcnnq_abl$a.so; [ $a = 0 ] && L=$R/cnn_quantization_amd/libcnnq_hip.so
  CNNQ_HIP_LIB=$L timeout 300 python tools/bench_group.py --rounds 1 --reps 8 --shapes 64x112,256x56,128x56,512x28,64x56 > $O/abl_$a.log 2>&1
done
tail -c 600 $O/bench.json
cat $O/mall3.log
cat $O/chunk_chain.log
for a in 0 1 2 3; do echo "== FLAT_ABL=$a"; grep "^C=" $O/abl_$a.log | cut -c1-75,118-160; done
