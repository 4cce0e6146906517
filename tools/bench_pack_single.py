#!/usr/bin/env python3
"""Per-forward time of config 2 with the packed 4-bit codes as the stored result, straight from x in the single-launch
kernels (ops.minmax_quantize_pack4), over the ResNet-50 b512 layer set.  Development aid (CNNQ_GRP_K sweeps)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from cnn_quantization_amd import ops  # noqa: E402

dev = torch.device('cuda')
layers = bench.build_workload(int(os.environ.get('BATCH', '512')), dev, seed=1)
bufs = [torch.empty(L['x'].numel() // 2, dtype=torch.uint8, device=dev) for L in layers]


def step():
    for L, b in zip(layers, bufs):
        ops.minmax_quantize_pack4(L['x'], 4, L['half'], out=b)


t = bench.timed_best(step, reps=5)
n = sum(L['x'].numel() for L in layers)
print('packed single launch: %.3f ms per forward, %.1f G elem/s, %.3f of 8 TB/s on 4.5 B/elem' % (t * 1e3, n / t / 1e9, n * 4.5 / t / 8e12))
