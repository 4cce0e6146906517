#!/bin/bash
# round 5: what the entropy of the codes (-me) costs config 2: the tests that touch it, whole forward and per layer
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -q -m gpu -x -k "entropy or single_outputs or hist" 2>&1 | tail -3
timeout 600 python tools/bench_entropy.py 2>&1 | grep -v amdgpu
