"""k_minmax and the table-driven k_qdq (cnnq_pc_qdq) vs the batch split S (CNNQ_DBG_S), alternating like the
real sequence, rotating buffers."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnn_quantization_amd import _lib
lib = _lib.load()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
tot = [0., 0., 0]
for (N, C, hw, cnt) in [(512, 64, 112, 1), (512, 256, 56, 4), (512, 128, 56, 1), (512, 512, 28, 5), (512, 64, 56, 6), (512, 256, 28, 1), (512, 1024, 14, 7), (512, 128, 28, 7), (512, 512, 14, 1), (512, 2048, 7, 4), (512, 256, 14, 11), (512, 512, 7, 5)]:
    HW = hw * hw
    n = N * C * HW
    nbuf = max(2, int(1.7e9 // (n * 4)))
    xs = [torch.randn(N, C, hw, hw, device='cuda') for _ in range(nbuf)]
    ys = [torch.empty_like(xs[0]) for _ in range(nbuf)]
    G = lib.cnnq_pc_groups(N, C, HW, 1)
    pmm = torch.empty(G, 2, C, dtype=torch.float32, device='cuda')
    qp = torch.empty(3, C, device='cuda'); qp[0] = 0.37; qp[1] = 7.; qp[2] = 15.
    reps = max(6, min(40, int(3e9 // (n * 4))))
    evs = []
    for rep in range(reps + 2):
        x, y = xs[rep % nbuf], ys[rep % nbuf]
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        lib.cnnq_pc_minmax(x.data_ptr(), N, C, HW, pmm.data_ptr(), st)
        e[1].record()
        lib.cnnq_pc_qdq(x.data_ptr(), y.data_ptr(), N, C, HW, qp.data_ptr(), None, None, 1, st)
        e[2].record()
        evs.append(e)
    torch.cuda.synchronize()
    evs = evs[2:]
    tm = sum(e[0].elapsed_time(e[1]) for e in evs) / len(evs) * 1e-3
    tq = sum(e[1].elapsed_time(e[2]) for e in evs) / len(evs) * 1e-3
    tot[0] += tm * cnt; tot[1] += tq * cnt; tot[2] += n * cnt
    print('[%d,%d,%d,%d] G=%d: minmax %6.1f us %5.0f | qdq %6.1f us %5.0f GB/s' % (N, C, hw, hw, G, tm * 1e6, n * 4 / tm / 1e9, tq * 1e6, n * 8 / tq / 1e9))
    del xs, ys; torch.cuda.empty_cache()
print('S=%s weighted: minmax %.0f GB/s, qdq %.0f GB/s, sum %.2f ms -> %.0f GB/s(12B)' % (os.environ.get('CNNQ_DBG_S', 'default'), tot[2] * 4 / tot[0] / 1e9, tot[2] * 8 / tot[1] / 1e9, (tot[0] + tot[1]) * 1e3, tot[2] * 12 / (tot[0] + tot[1]) / 1e9))
