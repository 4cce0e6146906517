#!/bin/bash
O=gpurun_out/r2z; mkdir -p $O
for r in 1 2; do
timeout 300 python tools/bench_stats_layers.py > $O/sl_new$r.log 2>&1
CNNQ_PLAN_MINWGS=1024 timeout 300 python tools/bench_stats_layers.py > $O/sl_min1k$r.log 2>&1
CNNQ_PLAN_MINWGS=1600 timeout 300 python tools/bench_stats_layers.py > $O/sl_min16$r.log 2>&1
paste <(cut -c1-28 $O/sl_new$r.log) <(cut -c19-24 $O/sl_min1k$r.log) <(cut -c19-24 $O/sl_min16$r.log)
done
