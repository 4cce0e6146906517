#!/bin/bash
# round 5: the 64-sample shard (the 8-GPU operating point) layer by layer under the planner's knobs (development build)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5; rm -f gpurun_out/r5/shard_sweep.jsonl
export CNNQ_HIP_LIB=$PWD/tools/alt/libcnnq_knobs.so
run() { tag=$1; shift; env "$@" timeout 300 python tools/bench_shard.py --tag $tag --json 2>&1 | grep -v amdgpu.ids; }
run default A=1 > gpurun_out/r5/shard_default.log; cat gpurun_out/r5/shard_default.log
for k in 8 16 32; do run K$k CNNQ_GRP_K=$k | tail -1; done
for t in 256 512 1024; do run T$t CNNQ_RES_T=$t | tail -1; done
for m in 0 96 384 768 100000; do run MINW$m CNNQ_RES_MIN_WGS=$m | tail -1; done
for u in 64 128 256; do run U$u CNNQ_RES_UNITS=$u | tail -1; done
for w in 256 512 2048; do run WGS$w CNNQ_GRP_WGS=$w | tail -1; done
for c in 1 2 8; do run CB$c CNNQ_GRP_CB=$c | tail -1; done
run MINCPC49 CNNQ_FLAT_MINCPC=49 | tail -1
run default2 A=1 | tail -1
