#!/bin/bash
O=$PWD/gpurun_out/r3_single; mkdir -p $O
python -m pytest tests/test_single_outputs_gpu.py -x -q > $O/pytest_single.log 2>&1; tail -8 $O/pytest_single.log
python -m pytest tests -m gpu -x -q --deselect tests/test_single_outputs_gpu.py > $O/pytest_rest.log 2>&1; tail -8 $O/pytest_rest.log
