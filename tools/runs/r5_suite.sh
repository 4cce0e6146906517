#!/bin/bash
# round 5: the whole -m gpu suite (log under gpurun_out/r5), slowest tests listed
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
timeout 2400 python -m pytest tests -q -m gpu --durations=15 --maxfail=6 > gpurun_out/r5/suite.log 2>&1
tail -40 gpurun_out/r5/suite.log
