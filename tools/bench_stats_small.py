#!/usr/bin/env python3
"""The seven statistics per layer shape at BATCH (default 64, the shard of an 8-GPU run): the three-launch chain against the single
launch FORCED on every shape (cnnq_pc_stats_single with flags bit 3: also row-piece tiles, k_stats_group) - which shapes the
single launch should be routed to at small batches.  Development aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from cnn_quantization_amd import ops  # noqa: E402

dev = torch.device('cuda')
batch = int(os.environ.get('BATCH', '64'))
tot = {'chain': 0., 'single': 0., 'best': 0.}
for (C, hw, half, count) in bench.RESNET50_CONV_OUTPUTS:
    xs = [bench.laplace_activation((batch, C, hw, hw), 7 + i, dev) for i in range(4)]
    N, HW = batch, hw * hw
    res = {}
    for name in ('chain', 'single'):
        def f(x):
            if name == 'chain':
                ops._ACIQ_SINGLE = False
                r = ops.pc_stats(x, N, C, HW, need_b=True, need_kurt=True, need_relu=True)
                ops._ACIQ_SINGLE = True
                return r
            return ops.pc_stats_single(x, N, C, HW, True, True, True, flags=8)
        if f(xs[0]) is None:
            res[name] = float('nan')
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            for x in xs:
                f(x)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 1e3 / 20
    best = min(v for v in res.values() if v == v)
    tot['chain'] += res['chain'] * count
    tot['single'] += (res['single'] if res['single'] == res['single'] else res['chain']) * count
    tot['best'] += best * count
    print('[%d,%4d,%3d,%3d] x%-2d chain %7.1f us   single %7.1f us' % (batch, C, hw, hw, count, res['chain'], res['single']), flush=True)
print('per forward: chain %.3f ms, single everywhere it exists %.3f ms, best of both %.3f ms' % (tot['chain'] / 1e3, tot['single'] / 1e3, tot['best'] / 1e3))
