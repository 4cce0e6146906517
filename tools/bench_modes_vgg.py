#!/usr/bin/env python3
"""The single-launch forms side by side on the VGG-16 conv outputs at batch 512, HIP-event time per layer shape: config 2 (one
launch), config 3 (pass A + merge + bit allocation + the fused launch, MODE 0), config 5 without and with the code histogram
(MODE 1).  What the mid-tread arithmetic and the histogram cost the same structure.  Development aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from cnn_quantization_amd import ops  # noqa: E402

dev = torch.device('cuda')
batch = int(os.environ.get('BATCH', '512'))
tot = {}
for (C, hw, count) in bench.VGG16_CONV_OUTPUTS:
    n = batch * C * hw * hw
    nbuf = 2 if n * 4 > (1 << 30) else 4
    xs = [bench.laplace_activation((batch, C, hw, hw), 100 + i, dev) for i in range(nbuf)]
    ys = [torch.empty_like(x) for x in xs]
    modes = {
        'cfg2': lambda x, y: ops.act_qdq_per_channel(x, 4, positive=True, out=y),
        'cfg3': lambda x, y: ops.act_qdq_per_channel(x, 4, positive=True, clip='laplace', bit_alloc=True, out=y),
        'cfg5-nohist': lambda x, y: ops.mid_tread_qdq(x, 4, clip=True, sym=False, want_entropy=False),
        'cfg5': lambda x, y: ops.mid_tread_qdq(x, 4, clip=True, sym=False, want_entropy=True),
    }
    line = '[%d,%d,%d,%d] x%d' % (batch, C, hw, hw, count)
    for name, f in modes.items():
        for x, y in zip(xs, ys):
            f(x, y)
        torch.cuda.synchronize()
        reps = max(4, nbuf * 2)
        best = None
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(reps):
                f(xs[r % nbuf], ys[r % nbuf])
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e3 / reps
            best = t if best is None else min(best, t)
        line += '  %s %8.1f us' % (name, best)
        tot[name] = tot.get(name, 0.) + best * count
    print(line, flush=True)
    del xs, ys
print('per forward: ' + '  '.join('%s %.3f ms' % (k, v / 1e3) for k, v in tot.items()))
