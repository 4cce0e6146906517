#!/bin/bash
# round 6: config 4's single launch with smaller tiles (development build: tools/build_alt.sh knobs -DCNNQ_DEV_KNOBS)
for k in 32 16 8; do
  echo "== CNNQ_GRP_K=$k"
  CNNQ_HIP_LIB=$PWD/tools/alt/libcnnq_knobs.so CNNQ_GRP_K=$k ONLY=single python tools/bench_stats4.py 2>&1 | grep "config 4"
  CNNQ_HIP_LIB=$PWD/tools/alt/libcnnq_knobs.so CNNQ_GRP_K=$k python tools/bench_stats_layers.py 2>&1 | tail -14
done
