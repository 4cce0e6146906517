#!/bin/bash
# round 5: what the entropy of the codes (-me) costs config 2, whole forward and per layer (tools/bench_entropy.py)
cd "$GRAFT_REPO_ROOT"
timeout 600 python tools/bench_entropy.py 2>&1 | grep -v amdgpu
