#!/bin/bash
O=$PWD/gpurun_out/r3_pack; mkdir -p $O
timeout 300 python -m pytest tests/test_parity_holes_gpu.py -x -q > $O/pytest.log 2>&1; echo rc=$?; grep -v amdgpu.ids $O/pytest.log | tail -6
timeout 300 python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
import bench
from cnn_quantization_amd import ops, _lib as Lb
dev = torch.device('cuda')
tot = {1: 0., 2: 0.}
elems = 0
for (C, hw, half, count) in bench.RESNET50_CONV_OUTPUTS:
    x = bench.laplace_activation((512, C, hw, hw), 100 + C + hw, dev)
    _, parts = ops.act_qdq_per_channel(x, 4, positive=half, clip='laplace', bit_alloc=True, want_parts=True)
    qp, bits = parts['qp'], parts['diag'][Lb.DIAG_BITS].contiguous()
    buf = torch.empty(ops.packed_capacity(x.shape), dtype=torch.uint8, device=dev)
    row = []
    for form in (1, 2):
        if form == 2 and (hw * hw) % 4:
            row.append(float('nan')); continue
        t = bench.timed_best(lambda: ops.quantize_packed(x, qp, bits, out=buf, form=form), reps=5)
        row.append(t * 1e6)
    t2 = row[1] if row[1] == row[1] else row[0]
    tot[1] += row[0] * count; tot[2] += t2 * count; elems += x.numel() * count
    print('C=%4d %3dx%-3d x%2d  general %7.1f us   lean %7.1f us   (%.0f -> %.0f GB/s of 4.5 B/elem)' % (
        C, hw, hw, count, row[0], row[1], x.numel() * 4.5 / row[0] / 1e3, x.numel() * 4.5 / t2 / 1e3), flush=True)
    del x, buf
print('sum over the 53 layers (one by one): general %.2f ms, lean (general for 7x7) %.2f ms' % (tot[1] / 1e3, tot[2] / 1e3))
PY
