#!/usr/bin/env python3
"""Per-layer time of the per-channel statistics (cnnq_pc_stats: pass A, pass B, merge) on the ResNet-50 conv outputs
at BATCH (default 512); FULL=1 (default) = all seven statistics (config 4), FULL=0 = min/max/mean/std/b (config 3).
Development aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import RESNET50_CONV_OUTPUTS, laplace_activation  # noqa: E402
from cnn_quantization_amd import ops  # noqa: E402


def timed(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    best = 1e9
    for _ in range(reps):
        ev[0].record()
        fn()
        ev[1].record()
        torch.cuda.synchronize()
        best = min(best, ev[0].elapsed_time(ev[1]))
    return best


def main():
    batch = int(os.environ.get('BATCH', '512'))
    full = os.environ.get('FULL', '1') == '1'
    dev = torch.device('cuda')
    tot = ideal = 0.
    for (C, hw, _half, rep) in RESNET50_CONV_OUTPUTS:
        x = laplace_activation((batch, C, hw, hw), 3, dev)
        kw = dict(need_b=True, need_kurt=full, need_relu=full)
        t = timed(lambda: ops.pc_stats(x, batch, C, hw * hw, **kw))
        gb = x.numel() * 4 / 1e9
        print('C=%4d hw=%3d x%-2d  %.3f ms  %.2f TB/s (8 B/elem)' % (C, hw, rep, t, 2 * gb / t), flush=True)
        tot += t * rep
        ideal += 2 * gb / 8.0 * rep
        del x
    print('total %.3f ms = %.1f %% of 8 TB/s' % (tot, ideal / tot * 100))


if __name__ == '__main__':
    main()
