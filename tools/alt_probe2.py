import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnn_quantization_amd import _lib
lib = ctypes.CDLL(sys.argv[1]) if len(sys.argv) > 1 else _lib.load()
for n, (r, a) in _lib.SIGNATURES.items():
    getattr(lib, n).restype = r; getattr(lib, n).argtypes = a
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
tot_t, tot_e = 0., 0
for (N, C, hw, cnt) in [(512, 64, 112, 1), (512, 256, 56, 4), (512, 128, 56, 1), (512, 512, 28, 5), (512, 64, 56, 6), (512, 256, 28, 1), (512, 1024, 14, 7), (512, 128, 28, 7), (512, 512, 14, 1), (512, 2048, 7, 4), (512, 256, 14, 11), (512, 512, 7, 5)]:
    HW = hw * hw
    n = N * C * HW
    nbuf = max(2, int(1.7e9 // (n * 4)))
    xs = [torch.randn(N, C, hw, hw, device='cuda') for _ in range(nbuf)]
    ys = [torch.empty_like(xs[0]) for _ in range(nbuf)]
    G = lib.cnnq_pc_groups(N, C, HW, 1)
    mm = torch.empty(G, 2, C, dtype=torch.float32, device='cuda')
    reps = max(6, min(40, int(3e9 // (n * 4))))
    evs = []
    for rep in range(reps + 2):
        x, y = xs[rep % nbuf], ys[rep % nbuf]
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        lib.cnnq_pc_minmax(x.data_ptr(), N, C, HW, mm.data_ptr(), st)
        e[1].record()
        lib.cnnq_pc_qdq_minmax(x.data_ptr(), y.data_ptr(), N, C, HW, 4, 0, mm.data_ptr(), G, None, None, None, 1, st)
        e[2].record()
        evs.append(e)
    torch.cuda.synchronize()
    evs = evs[2:]
    tm = sum(e[0].elapsed_time(e[1]) for e in evs) / len(evs) * 1e-3
    tq = sum(e[1].elapsed_time(e[2]) for e in evs) / len(evs) * 1e-3
    tt = evs[0][0].elapsed_time(evs[-1][2]) / len(evs) * 1e-3
    tot_t += tt * cnt; tot_e += n * cnt
    print('[%d,%d,%d,%d] %6.1f MB: minmax %6.1f us %5.0f | qdq %6.1f us %5.0f | pair %6.1f us %5.0f GB/s(12B)' % (N, C, hw, hw, n * 4 / 1e6, tm * 1e6, n * 4 / tm / 1e9, tq * 1e6, n * 8 / tq / 1e9, tt * 1e6, n * 12 / tt / 1e9))
    del xs, ys; torch.cuda.empty_cache()
print('weighted %.0f GB/s(12B), %.2f ms per forward' % (tot_e * 12 / tot_t / 1e9, tot_t * 1e3))
