"""Round 4 probe: the load direction of the packed storage (flat / lean kernel) by the size of the rotated buffer set."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import laplace_activation
from cnn_quantization_amd import _lib as L, ops
dev = torch.device('cuda')
def timed(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    best = 1e9
    for _ in range(reps):
        e[0].record(); fn(); e[1].record(); torch.cuda.synchronize()
        best = min(best, e[0].elapsed_time(e[1]))
    return best
for (C, hw) in ((128, 56), (256, 56), (1024, 14)):
    x = laplace_activation((512, C, hw, hw), 3, dev)
    _, parts = ops.act_qdq_per_channel(x, 4, clip='laplace', bit_alloc=True, want_parts=True)
    qp, bits = parts['qp'], parts['diag'][L.DIAG_BITS].contiguous()
    buf = torch.empty(ops.packed_capacity(x.shape), dtype=torch.uint8, device=dev)
    packed, rowoff = ops.quantize_packed(x, qp, bits, out=buf)
    used = 512 * int(rowoff[C])
    gb = (x.numel() * 4 + used) / 1e9
    for R in (1, 2, 3, 4, 6, 8):
        if R * x.numel() * 4 > 14e9:
            continue
        pks = [buf] + [buf.clone() for _ in range(R - 1)]
        ys = [torch.empty_like(x) for _ in range(R)]
        line = 'C=%d hw=%d, %d buffer pairs (%.1f GB of y):' % (C, hw, R, R * x.numel() * 4 / 1e9)
        for form, nm in ((3, 'flat'), (2, 'lean')):
            t = timed(lambda: [ops.dequantize_packed(pp, x.shape, qp, bits, rowoff, out=yy, form=form) for pp, yy in zip(pks, ys)]) / R
            line += '  %s %.3f ms %.2f TB/s' % (nm, t, gb / t)
        print(line, flush=True)
        del pks, ys
    del x, buf, packed
