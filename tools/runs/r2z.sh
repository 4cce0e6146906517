#!/bin/bash
O=gpurun_out/r2z; mkdir -p $O
python tools/host_profile.py 2>&1 | grep "host\|function calls" 
SHAPE=2,8,4,4 python tools/host_profile.py 2>&1 | grep "host\|function calls" 
python tools/host_overhead.py 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; tail -n 4 $O/pytest_all.log
