#!/usr/bin/env python3
"""Per-layer time of the bit-allocated packed storage pass (quantize_packed / dequantize_packed) on the
ResNet-50 conv outputs at BATCH (default 512).  Development aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import RESNET50_CONV_OUTPUTS, laplace_activation  # noqa: E402
from cnn_quantization_amd import _lib as L  # noqa: E402
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
    dev = torch.device('cuda')
    tq = td = byts = 0.
    keep = []
    for (C, hw, half, rep) in RESNET50_CONV_OUTPUTS:
        x = laplace_activation((batch, C, hw, hw), 3, dev)
        _, parts = ops.act_qdq_per_channel(x, 4, positive=half, clip='laplace', bit_alloc=True, want_parts=True)
        qp, bits = parts['qp'], parts['diag'][L.DIAG_BITS].contiguous()
        if os.environ.get('FORCE_BITS'):
            bits = torch.full_like(bits, float(os.environ['FORCE_BITS']))
        buf = torch.empty(ops.packed_capacity(x.shape), dtype=torch.uint8, device=dev)
        packed, rowoff = ops.quantize_packed(x, qp, bits, out=buf)
        used = batch * int(rowoff[C])
        # rotate over distinct buffers: a repeatedly written output of <= 200 MB would stay in the Infinity Cache
        R = 6 if x.numel() * 4 <= (512 << 20) else 2
        xs = [x] + [x.clone() for _ in range(R - 1)]
        pks = [buf] + [buf.clone() for _ in range(R - 1)]
        ys = [torch.empty_like(x) for _ in range(R)]
        pform = int(os.environ.get('PACK_FORM', '0'))
        if pform == 3 and (hw * hw) % 4:
            pform = 0
        a = timed(lambda: [ops.quantize_packed(xx, qp, bits, out=pp, form=pform, rowoff=rowoff) for xx, pp in zip(xs, pks)]) / R
        y = ys[0]
        b = timed(lambda: [ops.dequantize_packed(pp, x.shape, qp, bits, rowoff, out=yy) for pp, yy in zip(pks, ys)]) / R
        del xs, pks, ys
        gb = (x.numel() * 4 + used) / 1e9
        print('C=%4d hw=%3d x%-2d  %.2f bits  pack %.3f ms (%.2f TB/s)  unpack %.3f ms (%.2f TB/s)' % (
            C, hw, rep, used * 8 / x.numel(), a, gb / a, b, gb / b), flush=True)
        tq += a * rep
        td += b * rep
        byts += gb * rep
        for _ in range(rep):
            keep.append((x, qp, bits, torch.empty_like(buf), rowoff, torch.empty_like(x)))
        del x, y, buf, packed
    # the same passes over the 53 tensors back to back (cold inputs, as bench.py times them)
    pf = int(os.environ.get('PACK_FORM', '0'))
    a = timed(lambda: [ops.quantize_packed(x, qp, bits, out=b, rowoff=ro, form=(pf if (x.shape[2] * x.shape[3]) % 4 == 0 else 0)) for x, qp, bits, b, ro, yy in keep], reps=3)
    b = timed(lambda: [ops.dequantize_packed(b_, x.shape, qp, bits, ro, out=yy) for x, qp, bits, b_, ro, yy in keep], reps=3)
    print('back to back: pack %.3f ms  unpack %.3f ms' % (a, b))
    print('total pack %.3f ms (%.2f TB/s)  unpack %.3f ms (%.2f TB/s)' % (tq, byts / tq, td, byts / td))


if __name__ == '__main__':
    main()
