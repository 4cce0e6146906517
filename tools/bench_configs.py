#!/usr/bin/env python3
"""Throughput of the other BASELINE configs on one MI355X (development aid, not the driver's bench):
  cfg1  per-tensor int8 GEMMLOWP Q/DQ with dynamic min/max on [32,64,112,112]          (12 B/elem)
  cfg3  ResNet-50 b512, per-channel int4 + ACIQ laplace + bit allocation (dynamic)      (16 B/elem)
  cfg4  ResNet-50 b512, -sm collect: the seven per-channel statistics                   ( 8 B/elem)
  cfg5  VGG-16 b512, mid-tread W4A4 per channel + ACIQ + bin allocation + entropy        (16 B/elem)
Batch can be reduced with BATCH=... for quick runs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import RESNET50_CONV_OUTPUTS, laplace_activation  # noqa: E402
from cnn_quantization_amd import ops  # noqa: E402

VGG16_CONV_OUTPUTS = [(64, 224, 2), (128, 112, 2), (256, 56, 3), (512, 28, 3), (512, 14, 3)]


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    batch = int(os.environ.get('BATCH', '512'))
    which = os.environ.get('CFGS', '1,3,4,5').split(',')
    dev = torch.device('cuda')
    if '1' in which:
        xs = [laplace_activation((32, 64, 112, 112), 1 + i, dev) for i in range(16)]
        n = xs[0].numel()

        def cfg1():
            for x in xs:
                ops.minmax_qdq_per_tensor(x, 8, avg_over_batch=True)
        t = timed(cfg1)
        print('cfg1 per-tensor int8 [32,64,112,112] x16: %.1f us/tensor  %.1f G elem/s  %.0f GB/s (12 B/elem)' % (
            t / 16 * 1e6, n * 16 / t / 1e9, n * 16 * 12 / t / 1e9))
        del xs
    layers = []
    if any(c in which for c in ('3', '4')):
        seed = 100
        for (C, hw, half, count) in RESNET50_CONV_OUTPUTS:
            for _ in range(count):
                layers.append((laplace_activation((batch, C, hw, hw), seed, dev), half))
                seed += 1
        elems = sum(x.numel() for x, _ in layers)
    if '3' in which:
        def cfg3():
            for x, half in layers:
                ops.act_qdq_per_channel(x, 4, positive=half, clip='laplace', bit_alloc=True)
        t = timed(cfg3)
        print('cfg3 ResNet-50 b%d ACIQ+bit-alloc: %.2f ms/forward  %.1f G elem/s  %.0f GB/s (16 B/elem) = %.1f%% of 8 TB/s' % (
            batch, t * 1e3, elems / t / 1e9, elems * 16 / t / 1e9, elems * 16 / t / 8e12 * 100))
        for name, kw in (('pass A moments', dict()), ('pass A+B', dict(need_b=True))):
            tt = timed(lambda: [ops.pc_stats(x, x.shape[0], x.shape[1], x.shape[2] * x.shape[3], **kw) for x, _ in layers])
            by = 4 if not kw else 8
            print('      %s: %.2f ms  %.0f GB/s (%d B/elem)' % (name, tt * 1e3, elems * by / tt / 1e9, by))
    if '4' in which:
        def cfg4():
            for x, _ in layers:
                ops.pc_stats(x, x.shape[0], x.shape[1], x.shape[2] * x.shape[3], need_b=True, need_kurt=True,
                             need_relu=True)
        t = timed(cfg4)
        print('cfg4 ResNet-50 b%d collect 7 stats: %.2f ms/forward  %.1f G elem/s  %.0f GB/s (8 B/elem) = %.1f%% of 8 TB/s' % (
            batch, t * 1e3, elems / t / 1e9, elems * 8 / t / 1e9, elems * 8 / t / 8e12 * 100))
    del layers
    torch.cuda.empty_cache()
    if '5' in which:
        vb = int(os.environ.get('VGG_BATCH', str(batch)))
        vl = []
        seed = 500
        for (C, hw, count) in VGG16_CONV_OUTPUTS:
            for _ in range(count):
                vl.append(laplace_activation((vb, C, hw, hw), seed, dev))
                seed += 1
        elems = sum(x.numel() for x in vl)

        def cfg5():
            for x in vl:
                ops.mid_tread_qdq(x, 4, clip=True, sym=False, want_entropy=True)
        t = timed(cfg5, reps=2)
        print('cfg5 VGG-16 b%d mid-tread+entropy: %.2f ms/forward  %.1f G elem/s  %.0f GB/s (16 B/elem) = %.1f%% of 8 TB/s' % (
            vb, t * 1e3, elems / t / 1e9, elems * 16 / t / 1e9, elems * 16 / t / 8e12 * 100))

        def cfg5n():
            for x in vl:
                ops.mid_tread_qdq(x, 4, clip=True, sym=False, want_entropy=False)
        t = timed(cfg5n, reps=2)
        print('     without entropy:               %.2f ms/forward  %.1f G elem/s  %.0f GB/s (16 B/elem)' % (
            t * 1e3, elems / t / 1e9, elems * 16 / t / 1e9))


if __name__ == '__main__':
    main()


def f3_line():
    """Stored-format variant of config 2: min/max -> parameters -> quantize+pack int4 (4.5 B/elem second pass)."""
    import ctypes
    from cnn_quantization_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda')
    layers, seed = [], 900
    for (C, hw, half, count) in RESNET50_CONV_OUTPUTS:
        if (hw * hw) % 4:
            continue
        for _ in range(count):
            layers.append((laplace_activation((512, C, hw, hw), seed, dev), half)); seed += 1
    elems = sum(x.numel() for x, _ in layers)
    packs = [torch.empty(x.numel() // 2, dtype=torch.uint8, device=dev) for x, _ in layers]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        for (x, half), pk in zip(layers, packs):
            N, C, HW = x.shape[0], x.shape[1], x.shape[2] * x.shape[3]
            G = lib.cnnq_pc_groups(N, C, HW, 1)
            pmm = torch.empty((G, 2, C), device=dev); qp = torch.empty((3, C), device=dev)
            lib.cnnq_pc_minmax(x.data_ptr(), N, C, HW, pmm.data_ptr(), st)
            lib.cnnq_pc_minmax_params(pmm.data_ptr(), G, C, 4, int(half), qp.data_ptr(), st)
            lib.cnnq_pc_quantize_pack4(x.data_ptr(), pk.data_ptr(), N, C, HW, qp.data_ptr(), st)
    t = timed(run)
    print('f3  ResNet-50 b512 (44 layers with HW%%4==0) min/max + quantize->packed int4: %.2f ms  %.1f G elem/s  '
          '%.0f GB/s (8.5 B/elem)' % (t * 1e3, elems / t / 1e9, elems * 8.5 / t / 1e9))


if __name__ == '__main__' and 'f3' in os.environ.get('CFGS', ''):
    f3_line()
