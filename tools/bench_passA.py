#!/usr/bin/env python3
"""Pass A alone (cnnq_pc_moments: min / max / sum / sum of squares per channel, one read of x) per ResNet-50 layer shape at
batch 512: us per call and TB/s of the 4 bytes per element, over rotating distinct buffers.  Development aid (kernel sweeps
with a -DCNNQ_DEV_KNOBS build: CNNQ_PLAN_WGS, CNNQ_PLAN_MINWGS)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from cnn_quantization_amd import ops  # noqa: E402

dev = torch.device('cuda')
batch = int(os.environ.get('BATCH', '512'))
tot = 0.
for (C, hw, half, count) in bench.RESNET50_CONV_OUTPUTS:
    n = batch * C * hw * hw
    nbuf = max(2, min(8, (1500 << 20) // (4 * n) + 1))
    xs = [bench.laplace_activation((batch, C, hw, hw), 100 + i, dev) for i in range(nbuf)]
    for x in xs:
        ops.pc_moments(x, batch, C, hw * hw)
    torch.cuda.synchronize()
    best = None
    reps = max(8, nbuf * 2)
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(reps):
            ops.pc_moments(xs[r % nbuf], batch, C, hw * hw)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / reps
        best = t if best is None else min(best, t)
    print('C=%4d %3dx%-3d x%2d  %7.1f us  %5.2f TB/s' % (C, hw, hw, count, best * 1e6, n * 4 / best / 1e12), flush=True)
    tot += best * count
    del xs
print('pass A per forward: %.3f ms' % (tot * 1e3))
