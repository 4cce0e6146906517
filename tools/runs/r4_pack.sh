#!/bin/bash
# round 4: packed nibbles out of the single launch through the LDS strip (16 / 8 byte stores): tests, then the per-forward time
O=$PWD/gpurun_out/r4_pack; mkdir -p $O
timeout 900 python -m pytest tests/test_single_outputs_gpu.py -x -q -k "packed" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for v in "" "CNNQ_PK_NARROW=1" "CNNQ_GRP_K=16" "CNNQ_GRP_K=8"; do
  echo "== $v"; env $v timeout 300 python tools/bench_pack_single.py 2>&1 | tail -1
done
echo "== again default"; timeout 300 python tools/bench_pack_single.py 2>&1 | tail -1
