#!/bin/bash
# round 4: the contract / concurrency / packed tests after the bench.py clean-up, and the packed single launch's time
O=$PWD/gpurun_out/r4_misc; mkdir -p $O
timeout 1500 python -m pytest tests/test_bench_contract_gpu.py tests/test_concurrent_gpu.py tests/test_single_outputs_gpu.py tests/test_graph_gpu.py -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
cat gpurun_out/concurrent_status.json
python tools/bench_pack_single.py 2>&1 | tail -1
