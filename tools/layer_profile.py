#!/usr/bin/env python3
"""Per-shape timing of the config-2 hot path (development aid; run on the GPU box): the fused
call cnnq_pc_minmax_qdq for several channel-group sizes, on rotating buffers (> 1.5 GB) so nothing
is served from the Infinity Cache except what the path itself leaves there."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import RESNET50_CONV_OUTPUTS  # noqa: E402
from cnn_quantization_amd import ops  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    batch = int(os.environ.get('BATCH', '512'))
    chunks = [0]
    tot = {c: 0. for c in chunks}
    el = 0
    print('%-24s %7s | %s' % ('shape', 'Melem', ' '.join('%9s' % ('ck%dMB' % c) for c in chunks)))
    for (C, hw, half, count) in RESNET50_CONV_OUTPUTS:
        N, HW = batch, hw * hw
        n = N * C * HW
        nbuf = max(2, int(1.6e9 // (n * 4)))
        xs = [torch.randn(N, C, hw, hw, device='cuda') for _ in range(nbuf)]
        ys = [torch.empty_like(xs[0]) for _ in range(min(nbuf, 6))]
        k = [0]
        res = []
        for ck in chunks:
            def f():
                k[0] += 1
                ops.minmax_qdq_fused(xs[k[0] % nbuf], N, C, HW, 4, half, out=ys[k[0] % len(ys)])
            t = timeit(f, 12)
            tot[ck] += t * count
            res.append('%5.0f/%4.0f' % (t * 1e6, n * 12 / t / 1e9))
        el += n * count
        print('%-24s %7.1f | %s   (us / GBps-12B)' % ('[%d,%d,%d,%d]x%d' % (N, C, hw, hw, count), n / 1e6, ' '.join(res)))
        del xs, ys
        torch.cuda.empty_cache()
    print('weighted GB/s (12 B/elem): ' + ' '.join('ck%d=%.0f' % (c, el * 12 / tot[c] / 1e9) for c in chunks))
    print('ms per forward:            ' + ' '.join('ck%d=%.2f' % (c, tot[c] * 1e3) for c in chunks))


if __name__ == '__main__':
    main()
