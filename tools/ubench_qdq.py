import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_qdq'))
lib.ubench.restype = ctypes.c_float
lib.ubench.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int]
names = {0: 'cw div J4 u2', 1: 'cw COPY J4 u2', 2: 'cw rcp J4 u2', 3: 'cw div J4 ntS', 4: 'cw div J4 ntLS', 5: 'cw div J2 u2',
         6: 'cw div J4 u1', 7: 'cw div J4 u4', 8: 'cw COPY J4 ntS', 9: 'cw div J1 u4', 10: 'flat COPY', 11: 'flat div',
         12: 'flat COPY ntS', 13: 'flat div ntS', 14: 'flat rcp', 15: 'cw COPY J2', 16: 'cw div J2 u4', 17: 'cw div J2 u4 ntS'}
shapes = [(512, 64, 112 * 112), (512, 256, 56 * 56)]
for (N, C, HW) in shapes:
    x = torch.randn(N, C, HW, device='cuda'); y = torch.empty_like(x)
    qp = torch.empty(3, C, device='cuda'); qp[0] = 0.37; qp[1] = 7.; qp[2] = 15.
    n = x.numel()
    print('shape', (N, C, HW))
    for v in sorted(names):
        res = []
        for S, grid in ((8, 2048), (16, 4096), (24, 8192), (32, 16384)):
            ms = lib.ubench(v, x.data_ptr(), y.data_ptr(), N, C, HW, S, grid, qp.data_ptr(), 10)
            res.append('%5.0f' % (n * 8 / ms / 1e6))
        print('  v%-2d %-18s GB/s @S=8/16/24/32 (flat grid 2k/4k/8k/16k): %s' % (v, names[v], ' '.join(res)))
