"""Tile-shape sweep for the register-resident single-launch kernels: copy rate of "load the whole tile, then store it"
as a function of rows x columns per workgroup and of the block -> tile order (tools/ubench_tile.hip)."""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = __import__('_ubuild').so('ubench_tile')
lib = ctypes.CDLL(so)
lib.utile.restype = ctypes.c_float
lib.utile.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
for (N, C, hw) in ((512, 256, 56), (512, 512, 28), (512, 1024, 14)):
    P4 = C * hw * hw // 4
    x = torch.randn(N * P4 * 4, device='cuda'); y = torch.empty_like(x)
    print('tensor [%d,%d,%d,%d] %.0f MB' % (N, C, hw, hw, x.numel() * 4 / 1e6))
    for (J, R) in ((1, 32), (1, 16), (1, 8), (1, 4), (1, 1), (2, 16), (2, 8), (2, 4), (2, 1), (4, 8), (4, 4), (4, 2), (4, 1), (8, 4), (8, 2), (8, 1), (16, 2), (16, 1)):
        r = []
        for dep in (0, 1):
            for order in (0, 1):
                ms = lib.utile(J, R, order, dep, x.data_ptr(), y.data_ptr(), N, P4, 5)
                r.append(x.numel() * 8 / ms / 1e6)
        print('  J=%2d (row segment %5d B) R=%2d tile %4d KB: per-load dependency %5.0f / %5.0f GB/s (col- / row-major blocks)   '
              'whole-tile dependency %5.0f / %5.0f' % (J, J * 4096, R, J * R * 4, r[0], r[1], r[2], r[3]), flush=True)
    del x, y
