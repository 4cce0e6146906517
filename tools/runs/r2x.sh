#!/bin/bash
O=gpurun_out/r2x; mkdir -p $O
timeout 300 python tools/bench_stats_group.py > $O/sg.log 2>&1; tail -n 30 $O/sg.log
FULL=0 timeout 300 python tools/bench_stats_group.py > $O/sg0.log 2>&1; tail -n 30 $O/sg0.log
