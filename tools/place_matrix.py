#!/usr/bin/env python3
"""Round 6: how much does the PLACEMENT of a tensor pair decide the time of the single launch on it?  K candidate allocations
for x and K for y of one layer shape (each its own hipMalloc: its own physical pages), every (x_i, y_j) combination timed with
HIP events (4 launches after a warm-up), the matrix printed.  Development aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from cnn_quantization_amd import ops  # noqa: E402

dev = torch.device('cuda')
K = int(os.environ.get('K', '5'))
shapes = [(512, 256, 56, 56), (512, 64, 112, 112), (512, 512, 28, 28)]
torch.cuda.memory.set_per_process_memory_fraction(1.0)
for shape in shapes:
    src = bench.laplace_activation(shape, 7, dev)
    # the caching allocator would hand the same block back: every candidate is kept alive until the matrix is done
    xs = [src.clone() for _ in range(K)]
    ys = [torch.empty_like(src) for _ in range(K)]
    half = False
    def t_of(x, y):
        ops.act_qdq_per_channel(x, 4, positive=half, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            ops.act_qdq_per_channel(x, 4, positive=half, out=y)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / 4
    print(list(shape), 'us per launch, rows x_i, columns y_j; addresses x', [hex(x.data_ptr()) for x in xs], 'y', [hex(y.data_ptr()) for y in ys], flush=True)
    m = [[t_of(x, y) for y in ys] for x in xs]
    for row in m:
        print('  ' + ' '.join('%7.1f' % v for v in row), flush=True)
    flat = sorted(v for row in m for v in row)
    print('  min %.1f  median %.1f  max %.1f  (best / median %.3f)' % (flat[0], flat[len(flat) // 2], flat[-1], flat[0] / flat[len(flat) // 2]), flush=True)
    # repeatability: the best and the worst pair again
    bi = min(((m[i][j], i, j) for i in range(K) for j in range(K)))
    wi = max(((m[i][j], i, j) for i in range(K) for j in range(K)))
    print('  again: best (%d,%d) %.1f -> %.1f   worst (%d,%d) %.1f -> %.1f' % (bi[1], bi[2], bi[0], t_of(xs[bi[1]], ys[bi[2]]), wi[1], wi[2], wi[0], t_of(xs[wi[1]], ys[wi[2]])), flush=True)
    del xs, ys, src
    torch.cuda.empty_cache()
