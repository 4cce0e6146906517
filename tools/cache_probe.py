"""Does the 256 MB Infinity Cache serve the second read?  hot = same buffer every launch,
cold = rotate over > 1 GB of buffers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnn_quantization_amd import ops

def timeit(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps

for (N, C, hw) in [(512, 512, 7), (512, 256, 14), (512, 128, 28), (512, 64, 56), (256, 64, 112), (512, 256, 56)]:
    HW = hw * hw
    n = N * C * HW
    nbuf = max(2, int(1.5e9 // (n * 4)))
    xs = [torch.randn(N, C, hw, hw, device='cuda') for _ in range(nbuf)]
    ys = [torch.empty_like(xs[0]) for _ in range(min(nbuf, 8))]
    st, _ = ops.pc_stats(xs[0], N, C, HW); qp, _ = ops.pc_params(st, 4)
    k = [0]
    def s_hot(): ops.pc_moments(xs[0], N, C, HW)
    def s_cold():
        k[0] += 1; ops.pc_moments(xs[k[0] % nbuf], N, C, HW)
    def q_hot(): ops.pc_qdq(xs[0], N, C, HW, qp, out=ys[0])
    def q_cold():
        k[0] += 1; ops.pc_qdq(xs[k[0] % nbuf], N, C, HW, qp, out=ys[k[0] % len(ys)])
    def seq_cold():
        k[0] += 1; i = k[0] % nbuf
        ops.pc_moments(xs[i], N, C, HW); ops.pc_qdq(xs[i], N, C, HW, qp, out=ys[k[0] % len(ys)])
    r = [timeit(f) for f in (s_hot, s_cold, q_hot, q_cold, seq_cold)]
    print('[%d,%d,%d,%d] %6.1f MB nbuf=%d | stats hot %6.1f us %5.0f GB/s, cold %6.1f us %5.0f GB/s | qdq hot %6.1f us %5.0f GB/s, cold %6.1f us %5.0f GB/s | moments+qdq cold %6.1f us (sum %6.1f)' % (
        N, C, hw, hw, n * 4 / 1e6, nbuf, r[0] * 1e6, n * 4 / r[0] / 1e9, r[1] * 1e6, n * 4 / r[1] / 1e9,
        r[2] * 1e6, n * 8 / r[2] / 1e9, r[3] * 1e6, n * 8 / r[3] / 1e9, r[4] * 1e6, (r[1] + r[3]) * 1e6))
    del xs, ys; torch.cuda.empty_cache()
