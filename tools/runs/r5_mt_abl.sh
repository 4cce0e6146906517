#!/bin/bash
# round 5: what the code histogram of config 5 is made of (ablation builds: no counting / no flush / no LDS atomic)
cd "$GRAFT_REPO_ROOT"
for lib in "" tools/alt/libcnnq_mta1.so tools/alt/libcnnq_mta2.so tools/alt/libcnnq_mta4.so; do echo "lib=${lib:-product}"; CNNQ_HIP_LIB=$lib python tools/bench_modes_vgg.py 2>&1 | grep -v amdgpu | sed 's/cfg2 *[0-9.]* us  cfg3 *[0-9.]* us//'; done
