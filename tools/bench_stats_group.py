#!/usr/bin/env python3
"""Per-layer time of the statistics of one tensor: single-read kernel (cnnq_pc_stats_group) vs the two-pass chain
(cnnq_pc_stats), ResNet-50 conv outputs at BATCH (default 512).  Development aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import RESNET50_CONV_OUTPUTS, laplace_activation  # noqa: E402
from cnn_quantization_amd import ops  # noqa: E402


def timed(fn, reps=5):
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
    tot = [0., 0.]
    seen = {}
    for (C, hw, _half, rep) in RESNET50_CONV_OUTPUTS:
        x = laplace_activation((batch, C, hw, hw), 3, dev)
        kw = dict(need_b=True, need_kurt=full, need_relu=full)
        os.environ['CNNQ_STATS_GROUP'] = '0'
        tc = timed(lambda: ops.pc_stats(x, batch, C, hw * hw, **kw))
        os.environ['CNNQ_STATS_GROUP'] = '1'
        tg = timed(lambda: ops.pc_stats(x, batch, C, hw * hw, **kw))
        sup = ops.pc_stats_group(x, batch, C, hw * hw, **kw) is not None
        gb = x.numel() * 4 / 1e9
        print('C=%4d hw=%3d x%d  chain %.3f ms (%.2f TB/s @8B)  single %.3f ms (%.2f TB/s @4B) %s' % (
            C, hw, rep, tc, 2 * gb / tc, tg, gb / tg, '' if sup else '[chain]'), flush=True)
        tot[0] += tc * rep
        tot[1] += tg * rep
        del x
    print('total chain %.3f ms, default route %.3f ms, status %d' % (tot[0], tot[1], 0))


if __name__ == '__main__':
    main()
