import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_qdq'))
lib.useq.restype = ctypes.c_float
P = ctypes.c_void_p
lib.useq.argtypes = [ctypes.c_int, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, ctypes.c_int]
names = {8: 'read-only alone', 9: 'qdq fwd alone', 0: 'ro + qdq fwd', 1: 'ro + qdq REV', 2: 'ro + qdq REV ntS', 3: 'ro + qdq REV ntLS', 4: 'ro + qdq fwd ntS'}
for (N, C, HW) in [(512, 1, 12544), (512, 2, 12544), (512, 3, 12544), (512, 4, 12544), (512, 6, 12544), (512, 8, 12544), (512, 16, 12544)]:
    n = N * C * HW
    nbuf = max(3, int(2.5e9 // (n * 8)))
    xs = [torch.randn(N, C, HW, device='cuda') for _ in range(nbuf)]
    ys = [torch.empty_like(xs[0]) for _ in range(nbuf)]
    xp = (ctypes.c_void_p * nbuf)(*[t.data_ptr() for t in xs]); yp = (ctypes.c_void_p * nbuf)(*[t.data_ptr() for t in ys])
    qp = torch.empty(3, C, device='cuda'); qp[0] = 0.37; qp[1] = 7.; qp[2] = 15.
    scratch = torch.empty(1 << 20, device='cuda')
    S = max(1, min(N, 4096 // (C * 4)))
    out = []
    for v in (8, 9, 0, 1, 2, 3, 4):
        ms = lib.useq(v, xp, yp, nbuf, N, C, HW, S, qp.data_ptr(), scratch.data_ptr(), 20)
        out.append('%s %.1f us' % (names[v], ms * 1e3))
    print('x=%6.1f MB S=%d nbuf=%d | %s' % (n * 4 / 1e6, S, nbuf, ' | '.join(out)))
    del xs, ys; torch.cuda.empty_cache()
