import os, sys
sys.path.insert(0, '/root/repo')
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
for (C, hw) in ((256, 56), (1024, 14), (2048, 7)):
    xs = [laplace_activation((512, C, hw, hw), 3 + i, dev) for i in range(4)]
    _, parts = ops.act_qdq_per_channel(xs[0], 4, want_parts=True)
    qp = parts['qp']
    n = xs[0].numel()
    if hw % 2 == 0:
        t = timed(lambda: [ops.quantize_pack4(x, qp) for x in xs]) / 4
        print('C=%d hw=%d quantize_pack4 (uniform 4 bit, k_q_pack4): %.3f ms %.2f TB/s (4.5 B/elem)' % (C, hw, t, n * 4.5 / t / 1e9))
    bits = torch.full((C,), 4., device=dev)
    ro = ops.packed_layout(bits, hw * hw)
    bufs = [torch.empty(ops.packed_capacity(xs[0].shape), dtype=torch.uint8, device=dev) for _ in xs]
    t = timed(lambda: [ops.quantize_packed(x, qp, bits, out=b, rowoff=ro) for x, b in zip(xs, bufs)]) / 4
    print('C=%d hw=%d quantize_packed (k_pack_lean, 4 bits everywhere): %.3f ms %.2f TB/s' % (C, hw, t, n * 4.5 / t / 1e9))
    ys = [torch.empty_like(x) for x in xs]
    t = timed(lambda: [ops.pc_qdq(x, 512, C, hw * hw, qp, out=y) for x, y in zip(xs, ys)]) / 4
    print('C=%d hw=%d pc_qdq (k_qdq, 8 B/elem): %.3f ms %.2f TB/s' % (C, hw, t, n * 8 / t / 1e9))
    G = L.load().cnnq_pc_groups(512, C, hw * hw, 1)
    t = timed(lambda: [ops.pc_minmax(x, 512, C, hw * hw) if hasattr(ops, 'pc_minmax') else None for x in xs]) / 4 if hasattr(ops, 'pc_minmax') else 0
    if t: print('C=%d hw=%d pc_minmax (read only): %.3f ms %.2f TB/s' % (C, hw, t, n * 4 / t / 1e9))
    del xs, bufs, ys
