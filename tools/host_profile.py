#!/usr/bin/env python3
"""Where the host time of one hot call goes (cProfile over 20000 calls of ops.act_qdq_per_channel on a tiny tensor)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cnn_quantization_amd import ops  # noqa: E402

shape = tuple(int(v) for v in os.environ.get('SHAPE', '8,64,56,56').split(','))
x = torch.randn(shape, device='cuda')
y = torch.empty_like(x)
fn = lambda: ops.act_qdq_per_channel(x, 4, out=y)
for _ in range(200):
    fn()
torch.cuda.synchronize()
n = 20000
t0 = time.perf_counter()
for _ in range(n):
    fn()
t1 = time.perf_counter()
torch.cuda.synchronize()
print('host %.2f us per call' % ((t1 - t0) / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    fn()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
