#!/usr/bin/env python3
"""Config 4 (-sm collect: the seven per-channel statistics) over the ResNet-50 conv outputs at batch 512: the three-launch
chain against the single launch that reads x once (cnnq_pc_stats_single), whole forward and per layer shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnn_quantization_amd import ops
dev = torch.device('cuda')
batch = int(os.environ.get('BATCH', '512'))
layers, seed = [], 100
for (C, hw, half, count) in bench.RESNET50_CONV_OUTPUTS:
    for _ in range(count):
        layers.append(bench.laplace_activation((batch, C, hw, hw), seed, dev))
        seed += 1
elems = sum(x.numel() for x in layers)
def fwd():
    for x in layers:
        ops.pc_stats(x, x.shape[0], x.shape[1], x.shape[2] * x.shape[3], need_b=True, need_kurt=True, need_relu=True)
ONLY = os.environ.get('ONLY')
for single in {'chain': (False,), 'single': (True,), None: (False, True, False, True)}[ONLY]:
    ops._ACIQ_SINGLE = single
    t = bench.timed_best(fwd)
    print('config 4 b%d %-13s %.3f ms per forward  (%.1f G elem/s, %.2f of 8 TB/s on the 8 B accounting)' % (
        batch, 'single launch' if single else 'chain', t * 1e3, elems / t / 1e9, elems * 8 / t / 8e12), flush=True)
print('status word', ops.group_status(layers[0]))
if ONLY:
    sys.exit(0)
seen = set()
for x in layers:
    if tuple(x.shape) in seen:
        continue
    seen.add(tuple(x.shape))
    same = [xx for xx in layers if xx.shape == x.shape]
    times = {}
    for single in (False, True):
        ops._ACIQ_SINGLE = single
        f = lambda xx: ops.pc_stats(xx, xx.shape[0], xx.shape[1], xx.shape[2] * xx.shape[3], need_b=True, need_kurt=True, need_relu=True)
        for xx in same: f(xx)
        torch.cuda.synchronize()
        reps = max(2, 12 // len(same))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for xx in same: f(xx)
        e1.record(); torch.cuda.synchronize()
        times[single] = e0.elapsed_time(e1) * 1e3 / (reps * len(same))
    print('%-22s chain %8.1f us  single %8.1f us  x%.3f' % (list(x.shape), times[False], times[True], times[False] / times[True]), flush=True)
ops.reload_switches()
