#!/usr/bin/env python3
"""Round 6: is an output FAST when its physical memory comes from two different regions?  Outputs built with the virtual-memory API
(tools/_vmm.py): one piece; two halves created one after the other; two halves created with a spacer allocation between them; 2 P
pieces alternating between two pools that were created far apart.  Development aid (profiles/r06_placement.md)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from cnn_quantization_amd import ops  # noqa: E402
import _vmm as V  # noqa: E402

dev = torch.device('cuda')
shape = tuple(int(v) for v in os.environ.get('SHAPE', '512,256,56,56').split(','))
x = bench.laplace_activation(shape, 7, dev)
nbytes = x.numel() * 4
keep = []


def t_of(y, reps=3):
    ops.act_qdq_per_channel(x, 4, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.act_qdq_per_channel(x, 4, out=y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def tensor_of(pieces):
    m = V.Mapped(shape, pieces)
    keep.append(m)
    return torch.as_tensor(m, device=dev)


spacer_gb = int(os.environ.get('SPACER_GB', '24'))
print(list(shape), 'granularity', V.GR >> 20, 'MB', flush=True)
print('torch.empty x5        ', ' '.join('%.0f' % t_of(y) for y in [torch.empty_like(x) for _ in range(5)]), flush=True)
half = V.rnd(nbytes // 2)
print('one piece x4          ', ' '.join('%.0f' % t_of(tensor_of([(V.create(V.rnd(nbytes)), V.rnd(nbytes))])) for _ in range(4)), flush=True)
res = []
for _ in range(4):
    a, b = V.create(half), V.create(half)
    res.append(t_of(tensor_of([(a, half), (b, half)])))
print('two halves, adjacent  ', ' '.join('%.0f' % t for t in res), flush=True)
res = []
for _ in range(4):
    a = V.create(half)
    spacer = torch.empty(spacer_gb << 28, dtype=torch.float32, device=dev)
    b = V.create(half)
    del spacer
    torch.cuda.empty_cache()
    res.append(t_of(tensor_of([(a, half), (b, half)])))
print('two halves, %d GB apart' % spacer_gb, ' '.join('%.0f' % t for t in res), flush=True)
for P in (2, 8, 32):
    piece = V.rnd(nbytes // (2 * P))
    poolA = [V.create(piece) for _ in range(P)]
    spacer = torch.empty(spacer_gb << 28, dtype=torch.float32, device=dev)
    poolB = [V.create(piece) for _ in range(P)]
    del spacer
    torch.cuda.empty_cache()
    pieces = []
    for i in range(P):
        pieces += [(poolA[i], piece), (poolB[i], piece)]
    print('%2d pieces alternating between two pools' % (2 * P), '%.0f' % t_of(tensor_of(pieces)), flush=True)
