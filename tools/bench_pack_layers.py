#!/usr/bin/env python3
"""Per-layer time of the packed single launch (ops.minmax_quantize_pack4) on the ResNet-50 b512 shapes, rotating buffers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnn_quantization_amd import ops
dev = torch.device('cuda')
N = int(os.environ.get('BATCH', '512'))
tot = 0.
for (C, hw, half, count) in bench.RESNET50_CONV_OUTPUTS:
    n = N * C * hw * hw
    nbuf = max(2, min(8, (700 << 20) // (4 * n) + 1))
    xs = [bench.laplace_activation((N, C, hw, hw), 50 + i, dev) for i in range(nbuf)]
    bufs = [torch.empty(n // 2, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    def run(i):
        ops.minmax_quantize_pack4(xs[i], 4, half, out=bufs[i])
    for i in range(nbuf): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for r in range(reps): run(r % nbuf)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    tot += t * count
    print('C=%4d %3dx%-3d x%2d: %7.1f us  %.2f TB/s (4.5 B/elem)  read rate %.2f TB/s' % (C, hw, hw, count, t * 1e6, n * 4.5 / t / 1e12, n * 4 / t / 1e12), flush=True)
    del xs, bufs
print('per forward %.3f ms' % (tot * 1e3))
