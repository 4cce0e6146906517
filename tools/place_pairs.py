#!/usr/bin/env python3
"""Round 6: which pairs of physical regions make a fast output?  H half-size physical chunks created S GB apart (the spacers are
held while the chunks are created), every pair (i, j) mapped back to back as one output and timed under the single launch of
config 2; (i, i+) = two chunks created next to each other.  Development aid (profiles/r06_placement.md)."""
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
H = int(os.environ.get('H', '12'))
S = int(os.environ.get('S', '16'))
x = bench.laplace_activation(shape, 7, dev)
nbytes = x.numel() * 4
half = V.rnd(nbytes // 2)
chunks, twins, spacers = [], [], []
for i in range(H):
    chunks.append(V.create(half))
    twins.append(V.create(half))                   # created right behind chunk i: the same region
    spacers.append(torch.empty(S << 28, dtype=torch.float32, device=dev))
del spacers
torch.cuda.empty_cache()


def t_pair(a, b, reps=3):
    m = V.Mapped(shape, [(a, half), (b, half)])
    y = torch.as_tensor(m, device=dev)
    ops.act_qdq_per_channel(x, 4, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.act_qdq_per_channel(x, 4, out=y)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e3 / reps
    del y
    m.close()
    return t


print(list(shape), '%d half chunks of %.2f GB, %d GB apart; rows: first half, columns: second half; diagonal: chunk i + its twin' % (H, half / 2 ** 30, S), flush=True)
for i in range(H):
    row = []
    for j in range(H):
        row.append(t_pair(chunks[i], twins[i] if i == j else chunks[j]))
    print('  ' + ' '.join('%4.0f' % t for t in row), flush=True)
