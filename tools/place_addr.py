#!/usr/bin/env python3
"""Round 6: what distinguishes a fast OUTPUT placement?  One input, many candidate outputs of one shape: separate allocations
(address printed), and views at several offsets into ONE larger allocation.  Development aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from cnn_quantization_amd import ops  # noqa: E402

dev = torch.device('cuda')
shape = tuple(int(v) for v in os.environ.get('SHAPE', '512,256,56,56').split(','))
K = int(os.environ.get('K', '12'))
x = bench.laplace_activation(shape, 7, dev)
n = x.numel()


def t_of(y, reps=3):
    ops.act_qdq_per_channel(x, 4, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.act_qdq_per_channel(x, 4, out=y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print('x at', hex(x.data_ptr()), 'bytes', n * 4)
ys = [torch.empty_like(x) for _ in range(K)]
for y in ys:
    print('separate allocation  y %s  (mod 1 GB %4d MB, mod 64 MB %2d MB)  %.1f us' % (hex(y.data_ptr()), (y.data_ptr() >> 20) & 1023, (y.data_ptr() >> 20) & 63, t_of(y)), flush=True)
del ys
torch.cuda.empty_cache()
big = torch.empty(n + (1 << 28), dtype=torch.float32, device=dev)        # 1 GB of slack
for off_mb in (0, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1, 3):
    off = off_mb << 18                                                    # floats
    y = big[off:off + n].view(shape)
    print('one allocation %s + %4d MB  %.1f us' % (hex(big.data_ptr()), off_mb, t_of(y)), flush=True)
