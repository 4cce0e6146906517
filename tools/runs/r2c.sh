#!/bin/bash
O=gpurun_out/r2c; mkdir -p $O
for m in cached uncached memset; do
  timeout 200 python tools/exp_sync.py 64 $m > $O/exp_$m.log 2>&1
  CNNQ_HIP_LIB=$PWD/tools/alt/libcnnq_rmw.so timeout 200 python tools/exp_sync.py 64 $m > $O/exp_rmw_$m.log 2>&1
done
tail -13 $O/exp_uncached.log $O/exp_rmw_cached.log
