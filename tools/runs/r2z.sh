#!/bin/bash
O=gpurun_out/r2z; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "pack" > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
timeout 300 python tools/bench_packed.py 2>&1 | tail -13
