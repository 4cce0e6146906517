#!/usr/bin/env python3
"""Round 6: a map of a fresh process's device memory - G GB of candidate OUTPUTS of one shape allocated back to back right after
the input, each timed under the single launch of config 2, in allocation order.  Development aid (profiles/r06_placement.md)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from cnn_quantization_amd import ops  # noqa: E402

dev = torch.device('cuda')
shape = tuple(int(v) for v in os.environ.get('SHAPE', '512,256,56,56').split(','))
G = float(os.environ.get('GB', '200'))
x = bench.laplace_activation(shape, 7, dev)
K = int(G * 2 ** 30 / (x.numel() * 4))
ys, ts = [], []
for k in range(K):
    y = torch.empty_like(x)
    ys.append(y)
    ops.act_qdq_per_channel(x, 4, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ops.act_qdq_per_channel(x, 4, out=y)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / 3)
print(list(shape), '%d outputs of %.2f GB, us per launch in allocation order:' % (K, x.numel() * 4 / 2 ** 30))
for i in range(0, K, 16):
    print('  %5.1f GB: ' % (i * x.numel() * 4 / 2 ** 30) + ' '.join('%4.0f' % t for t in ts[i:i + 16]))
s = sorted(ts)
print('min %.0f  p10 %.0f  median %.0f  max %.0f; within 3 %% of the minimum: %d of %d' % (s[0], s[K // 10], s[K // 2], s[-1], sum(t <= s[0] * 1.03 for t in ts), K))
