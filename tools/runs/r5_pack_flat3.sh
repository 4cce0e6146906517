#!/bin/bash
# round 5: what bounds k_pack_flat - ablation builds (no stores / trivial codes / both) beside the product
cd "$GRAFT_REPO_ROOT"
for lib in "" tools/alt/libcnnq_pfa1.so tools/alt/libcnnq_pfa2.so tools/alt/libcnnq_pfa3.so; do echo "lib=${lib:-product}"; CNNQ_HIP_LIB=$lib PACK_FORM=3 timeout 600 python tools/bench_packed.py 2>&1 | grep -v amdgpu.ids | sed 's/unpack.*//' | grep -E "hw=112|C= 256 hw= 56|C=1024|C= 512 hw= 28|back|total"; done
