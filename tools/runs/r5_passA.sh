#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo product; python tools/bench_passA.py 2>&1 | grep -v amdgpu
for w in 2048 8192 16384; do echo "PLAN_WGS=$w"; CNNQ_HIP_LIB=tools/alt/libcnnq_knobs.so CNNQ_PLAN_WGS=$w python tools/bench_passA.py 2>&1 | grep -v amdgpu | grep -E "112x|56x56 x 4|28x28 x 5|1024|forward"; done
