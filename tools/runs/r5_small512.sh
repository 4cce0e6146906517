#!/bin/bash
# round 5: the small layers of the b512 step under the tile-height / workgroup-count knobs (development build)
cd "$GRAFT_REPO_ROOT"
for s in "" "CNNQ_GRP_K=16" "CNNQ_GRP_K=32" "CNNQ_GRP_WGS=512" "CNNQ_GRP_WGS=256" "CNNQ_GRP_WGS=2048"; do
  echo "== ${s:-default}"; env $s CNNQ_HIP_LIB=tools/alt/libcnnq_knobs.so python tools/bench_shard.py --batch 512 --reps 24 --tag x 2>&1 | grep -E "14x14|7x7 " | cut -c1-120
done
