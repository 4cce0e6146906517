#!/bin/bash
O=gpurun_out/r2i; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_distributed_gpu.py > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -n 15 $O/pytest_all.log
timeout 600 python -m pytest tests/test_distributed_gpu.py -q > $O/pytest_dist.log 2>&1; tail -n 5 $O/pytest_dist.log
