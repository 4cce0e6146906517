#!/bin/bash
O=$PWD/gpurun_out/r3_ptf; mkdir -p $O
timeout 150 python -m pytest tests/test_pt_fused_gpu.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -12
timeout 200 python -m pytest tests/test_full_size_gpu.py::test_config1_per_tensor_full_size tests/test_hip_parity.py -x -q > $O/pytest2.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/pytest2.log | tail -5
timeout 100 python - <<'PY'
import torch, time, sys
sys.path.insert(0, '.')
import bench
from cnn_quantization_amd import ops
dev = torch.device('cuda')
xs = [bench.laplace_activation((32, 64, 112, 112), 1 + i, dev) for i in range(16)]
for name, kw in (('fused', {}), ('chain', {'chain': True})):
    t = bench.timed_best(lambda: [ops.minmax_qdq_per_tensor(x, 8, avg_over_batch=True, **kw) for x in xs])
    print('%s: %.1f us per [32,64,112,112] tensor = %.0f G elem/s' % (name, t / 16 * 1e6, xs[0].numel() * 16 / t / 1e9))
PY
