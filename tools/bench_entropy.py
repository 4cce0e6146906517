#!/usr/bin/env python3
"""Config 2 with and without the entropy of the codes (-me) over the ResNet-50 conv outputs at batch 512: what the code
histogram inside the single launch costs, whole forward and per layer shape.  Development aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from cnn_quantization_amd import ops  # noqa: E402

dev = torch.device('cuda')
batch = int(os.environ.get('BATCH', '512'))
layers, seed = [], 100
for (C, hw, half, count) in bench.RESNET50_CONV_OUTPUTS:
    for _ in range(count):
        layers.append((bench.laplace_activation((batch, C, hw, hw), seed, dev), half))
        seed += 1
ys = [torch.empty_like(x) for x, _ in layers]
elems = sum(x.numel() for x, _ in layers)
def fwd(ent, batched):
    if ent and batched:
        with ops.entropy_batch():          # round 6: ONE entropy launch for the 53 tensors, at the end of the forward
            for (x, half), y in zip(layers, ys):
                ops.act_qdq_per_channel(x, 4, positive=half, want_entropy=True, out=y)
    else:
        for (x, half), y in zip(layers, ys):
            ops.act_qdq_per_channel(x, 4, positive=half, want_entropy=ent, out=y)


for ent, batched in ((False, False), (True, False), (True, True), (False, False), (True, False), (True, True)):
    t = bench.timed_best(lambda: fwd(ent, batched))
    print('config 2 b%d %-28s %.3f ms per forward (%.1f G elem/s)' % (batch, ('with entropy, one launch at the end' if batched else 'with entropy, per tensor') if ent else 'plain',
                                                                     t * 1e3, elems / t / 1e9), flush=True)
seen = set()
for (x, half), y in zip(layers, ys):
    if tuple(x.shape) in seen:
        continue
    seen.add(tuple(x.shape))
    ts = {}
    for ent in (False, True, 'batched'):
        f = lambda: ops.act_qdq_per_channel(x, 4, positive=half, want_entropy=bool(ent), out=y)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if ent == 'batched':
            with ops.entropy_batch():
                for _ in range(6):
                    f()
        else:
            for _ in range(6):
                f()
        e1.record(); torch.cuda.synchronize()
        ts[ent] = e0.elapsed_time(e1) * 1e3 / 6
    print('%-22s plain %8.1f us  entropy %8.1f us  x%.3f   batched %8.1f us  x%.3f' % (list(x.shape), ts[False], ts[True], ts[True] / ts[False],
                                                                                       ts['batched'], ts['batched'] / ts[False]), flush=True)
