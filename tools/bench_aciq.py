#!/usr/bin/env python3
"""Config 3 (ACIQ laplace + bit allocation, dynamic statistics) over the ResNet-50 conv outputs - or, with --vgg, config 5
(mid-tread + bin allocation + entropy) over the VGG-16 conv outputs: the chain against the single-launch form
(cnnq_pc_aciq_qdq_single / cnnq_pc_midtread_qdq_single), whole forward and per layer shape, on one GPU.

    python tools/bench_aciq.py [--batch 512] [--layers] [--vgg]

Prints ms per forward for both (best of 3, wall clock around a synchronised region as bench_other.py does it) and, with
--layers, HIP-event time per layer shape over rotating distinct buffers."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--layers', action='store_true')
    ap.add_argument('--no-ba', action='store_true')
    ap.add_argument('--vgg', action='store_true', help='config 5 on the VGG-16 conv outputs')
    ap.add_argument('--only', choices=['chain', 'single'], help='time one form only (profiling)')
    a = ap.parse_args()
    from cnn_quantization_amd import ops
    dev = torch.device('cuda')
    ba = not a.no_ba
    layers, seed = [], 100
    shapes = [(C, hw, True, n) for (C, hw, n) in bench.VGG16_CONV_OUTPUTS] if a.vgg else bench.RESNET50_CONV_OUTPUTS
    for (C, hw, half, count) in shapes:
        for _ in range(count):
            layers.append((bench.laplace_activation((a.batch, C, hw, hw), seed, dev), half))
            seed += 1
    elems = sum(x.numel() for x, _ in layers)
    ys = [None if a.vgg else torch.empty_like(x) for x, _ in layers]

    def one(x, half, y):
        if a.vgg:
            ops.mid_tread_qdq(x, 4, clip=True, sym=False, want_entropy=True)
        else:
            ops.act_qdq_per_channel(x, 4, positive=half, clip='laplace', bit_alloc=ba, out=y)

    def fwd():
        for (x, half), y in zip(layers, ys):
            one(x, half, y)

    res = {}
    for single in {'chain': (False,), 'single': (True,), None: (False, True, False, True)}[a.only]:
        ops._ACIQ_SINGLE = single
        t = bench.timed_best(fwd)
        res.setdefault(single, []).append(t)
        print('config %d b%d %-13s %.3f ms per forward  (%.1f G elem/s, %.2f of 8 TB/s on the 16 B accounting, %.2f on %d B moved)' % (
            5 if a.vgg else 3, a.batch, 'single launch' if single else 'chain', t * 1e3, elems / t / 1e9, elems * 16 / t / 8e12,
            elems * (12 if single else 16) / t / 8e12, 12 if single else 16), flush=True)
    print('status word', ops.group_status(layers[0][0]))
    if a.layers:
        seen = set()
        print('%-22s %10s %10s %8s' % ('layer', 'chain us', 'single us', 'ratio'))
        for (x, half) in layers:
            key = (tuple(x.shape), half)
            if key in seen:
                continue
            seen.add(key)
            same = [(xx, yy) for (xx, hh), yy in zip(layers, ys) if tuple(xx.shape) == key[0] and hh == half]
            times = {}
            for single in (False, True):
                ops._ACIQ_SINGLE = single
                for xx, yy in same:
                    one(xx, half, yy)
                torch.cuda.synchronize()
                reps = max(2, 12 // len(same))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    for xx, yy in same:
                        one(xx, half, yy)
                e1.record()
                torch.cuda.synchronize()
                times[single] = e0.elapsed_time(e1) * 1e3 / (reps * len(same))
            print('%-22s %10.1f %10.1f %8.3f' % (str(list(x.shape)) + ('+' if half else ''), times[False], times[True],
                                                 times[False] / times[True]), flush=True)
    ops.reload_switches()


if __name__ == '__main__':
    main()
