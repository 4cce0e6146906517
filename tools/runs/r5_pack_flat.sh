#!/bin/bash
# round 5: k_pack_flat (store direction, one-shot workgroups): the forms test, then per layer lean (form 0) vs flat (form 3)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_parity_holes_gpu.py -q -m gpu -x -k "packed" 2>&1 | tail -8
for f in 0 3 0 3; do echo "PACK_FORM=$f"; PACK_FORM=$f timeout 600 python tools/bench_packed.py 2>&1 | grep -v amdgpu.ids | sed 's/unpack.*//' ; done | tee gpurun_out/r5/pack_flat.log
