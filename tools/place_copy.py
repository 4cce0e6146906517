#!/usr/bin/env python3
"""Round 6: is a slow output placement slow for ANY kernel?  K candidate outputs of one shape: the single launch of config 2, a
plain device copy (y.copy_(x): linear streaming), a fill (y.fill_: writes only), and the chain's Q/DQ pass on each.  Development aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from cnn_quantization_amd import ops  # noqa: E402

dev = torch.device('cuda')
shape = tuple(int(v) for v in os.environ.get('SHAPE', '512,512,28,28').split(','))
K = int(os.environ.get('K', '12'))
x = bench.laplace_activation(shape, 7, dev)


def timed(f, reps=3):
    f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


ys = [torch.empty_like(x) for _ in range(K)]
print(list(shape), 'us: single launch | copy_ | fill_ | read-only sum of y | the three-launch chain of config 2 (CNNQ_RESIDENT=0)')
for y in ys:
    a = timed(lambda: ops.act_qdq_per_channel(x, 4, out=y))
    b = timed(lambda: y.copy_(x))
    c = timed(lambda: y.fill_(1.0))
    d = timed(lambda: ops.pc_moments(y, shape[0], shape[1], shape[2] * shape[3], False))
    ops._RESIDENT = False
    e = timed(lambda: ops.act_qdq_per_channel(x, 4, out=y))
    ops.reload_switches()
    print('%s  %7.1f | %7.1f | %7.1f | %7.1f | %7.1f' % (hex(y.data_ptr()), a, b, c, d, e), flush=True)
