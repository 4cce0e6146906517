#!/usr/bin/env python3
"""A reference-generated `-sm collect` case at a shape that HAS a single-launch plan (the cases of collect.npz,
[4,6,5,7], are too small for one - k_stats_flat was checked against fp64 torch and the chain only): the reference's own
StatisticManagerPerChannel.save_tensor_stats (statistic_manager_perchannel.py:45-79), imported with the set-up of make_golden.py,
on two batches of [16,3,28,28] per-channel Laplace activations (flat tiles: two members per channel), batch_avg False and True.

    python tests/golden/make_golden_collect_flat.py      # rewrites tests/golden/collect_flat.npz (build container only)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G  # noqa: E402  (imports the reference, the stub module, the spies; generates nothing at import)

torch = G.torch
smpc = G.smpc


def main():
    g = torch.Generator().manual_seed(909)
    d = {}
    shape = (16, 3, 28, 28)
    xs = [G.laplace_nchw(g, shape) for _ in range(2)]
    xs[1][:, 1] = xs[1][:, 1].clamp(min=0)          # a post-ReLU channel (std_pos == std)
    for k, x in enumerate(xs):
        d['x%d' % k] = x
    for bi, batch_avg in enumerate((False, True)):
        smpc.Singleton._instances.pop(smpc.StatisticManagerPerChannel, None)
        sm = smpc.StatisticManagerPerChannel('golden_flat_%d' % bi, load_stats=False, batch_avg=batch_avg,
                                             stats=['max', 'min', 'std', 'mean', 'kurtosis', 'b', 'std_pos'])
        for x in xs:
            sm.save_tensor_stats(x, 'activation', 'conv0_activation')
        for s in sm.stats_names:
            d['b%d_%s' % (bi, s)] = sm.stats['conv0_activation'][s]          # [2 batches, C]
    smpc.Singleton._instances.pop(smpc.StatisticManagerPerChannel, None)
    G.save('collect_flat', d)


if __name__ == '__main__':
    torch.set_num_threads(1)
    main()
