#!/bin/bash
# round 3: flat tiles (k_mmq_flat) - parity tests, then per-layer timing against the row-piece tiling on the same box
O=$PWD/gpurun_out/r3_flat; mkdir -p $O
python -m pytest tests/test_group_gpu.py tests/test_resident_gpu.py -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
CNNQ_GRP_FLAT=0 python tools/bench_group.py --rounds 2 --reps 6 > $O/rows.log 2>&1
python tools/bench_group.py --rounds 2 --reps 6 > $O/flat.log 2>&1
CNNQ_GRP_K=16 python tools/bench_group.py --rounds 1 --reps 6 > $O/flat_k16.log 2>&1
CNNQ_GRP_STAGGER=2,20 python tools/bench_group.py --rounds 1 --reps 6 > $O/flat_stag.log 2>&1
CNNQ_HIP_LIB=$PWD/tools/libcnnq_trace.so python tools/trace_group.py --shapes 256x56,512x28 > $O/trace.log 2>&1
for f in rows flat flat_k16 flat_stag; do echo "== $f"; grep -h "^C=\|per forward group" $O/$f.log | sed 's/| chain.*| group/| group/' | cut -c1-130; done
