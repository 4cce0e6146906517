#!/bin/bash
O=gpurun_out/r2q; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_holes_gpu.py -q -k "packed" > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
timeout 600 python - > $O/packed.log 2>&1 <<'PY'
import torch, time, bench
from cnn_quantization_amd import ops, _lib as Lb
dev = torch.device('cuda')
layers, seed = [], 100
for (C, hw, half, count) in bench.RESNET50_CONV_OUTPUTS:
    for _ in range(count):
        layers.append((bench.laplace_activation((512, C, hw, hw), seed, dev), half)); seed += 1
elems = sum(x.numel() for x, _ in layers)
pk = []
for (x, half) in layers:
    _, parts = ops.act_qdq_per_channel(x, 4, positive=half, clip='laplace', bit_alloc=True, want_parts=True)
    pk.append((x, parts['qp'], parts['diag'][Lb.DIAG_BITS].contiguous()))
stored = [ops.quantize_packed(x, qp, bits) for x, qp, bits in pk]
nbytes = sum(p.numel() for p, _ in stored)
bufs = [torch.empty(ops.packed_capacity(x.shape), dtype=torch.uint8, device=dev) for x, _, _ in pk]
t = bench.timed_best(lambda: [ops.quantize_packed(x, qp, bits, out=b) for (x, qp, bits), b in zip(pk, bufs)])
print('quantize_packed: %.2f ms  %.1f G elem/s  %.3f B/elem written  %.0f GB/s' % (t * 1e3, elems / t / 1e9, nbytes / elems, elems * (4 + nbytes / elems) / t / 1e9))
ys = [torch.empty_like(x) for x, _, _ in pk[:1]]
t = bench.timed_best(lambda: [ops.dequantize_packed(p, tuple(x.shape), qp, bits, ro) for (x, qp, bits), (p, ro) in zip(pk, stored)])
print('dequantize_packed: %.2f ms  %.1f G elem/s  %.0f GB/s' % (t * 1e3, elems / t / 1e9, elems * (4 + nbytes / elems) / t / 1e9))
PY
grep -v amdgpu $O/packed.log
