#!/usr/bin/env python3
"""-sm use + -bca on the ResNet-50 b512 layer set: static per-channel parameters, Q/DQ + activation bias
correction; two-step (quantize, re-read both, update in place: 24 B/elem) vs fused (12 B/elem)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import RESNET50_CONV_OUTPUTS, laplace_activation  # noqa: E402
from cnn_quantization_amd import ops  # noqa: E402


def main():
    dev = torch.device('cuda')
    layers, seed = [], 300
    for (C, hw, half, count) in RESNET50_CONV_OUTPUTS:
        for _ in range(count):
            x = laplace_activation((512, C, hw, hw), seed, dev)
            seed += 1
            N, HW = 512, hw * hw
            stats, _ = ops.pc_stats(x, N, C, HW)
            qp, _ = ops.pc_params(stats, 4, half, 'no', False)
            layers.append((x, torch.empty_like(x), N, C, HW, qp, half))
    elems = sum(l[0].numel() for l in layers)

    def two_step():
        for x, y, N, C, HW, qp, half in layers:
            ops.act_bias_correction_(x, ops.pc_qdq(x, N, C, HW, qp, out=y), half)

    def fused():
        for x, y, N, C, HW, qp, half in layers:
            ops.qdq_bias_corrected(x, N, C, HW, qp, half, out=y)

    for name, fn, by in (('two-step', two_step, 24), ('fused', fused, 12)):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print('%-9s %.2f ms/forward  %.1f G elem/s  (%d B/elem -> %.0f GB/s)' % (name, best * 1e3, elems / best / 1e9, by,
                                                                                  elems * by / best / 1e9))


if __name__ == '__main__':
    main()
