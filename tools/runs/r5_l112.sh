#!/bin/bash
# round 5: the 112x112 layer of the headline (196 members per channel) under tile / dispatch knobs (development build)
cd "$GRAFT_REPO_ROOT"
for s in "" "CNNQ_FLAT_KL=8" "CNNQ_FLAT_KL=8 CNNQ_GRP_CB=1" "CNNQ_GRP_K=16" ""; do
  echo "== ${s:-default}"; env $s CNNQ_HIP_LIB=tools/alt/libcnnq_knobs.so python tools/bench_shard.py --batch 512 --reps 12 --tag x 2>&1 | grep -E "112x112|C= 256  56x56|per forward"
done
