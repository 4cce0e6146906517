"""Workgroup-structure sweep for the single-launch min/max + Q/DQ kernels (tools/ubench_pipe.hip): one-shot
strided-row tiles (round 2) vs one-shot flat tiles vs persistent double-buffered flat tiles, with a simulated
exchange delay between a tile's loads and its stores and with / without the real Q/DQ arithmetic."""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_pipe'))
lib.upipe.restype = ctypes.c_float
lib.upipe.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6
VAR = [('rows', 0, 32, 3), ('rows', 0, 16, 6), ('flat', 1, 32, 3), ('flat', 1, 16, 6), ('flat', 1, 8, 8),
       ('pipe', 2, 16, 3), ('pipe', 2, 16, 2), ('pipe', 2, 8, 6), ('pipe', 2, 8, 4), ('pipe', 2, 24, 2), ('pipe', 2, 4, 8)]
shapes = ((512, 256, 56), (512, 64, 112), (512, 512, 28), (512, 1024, 14))
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
PK = len(sys.argv) > 2 and sys.argv[2] == 'pk'       # the packed regime: 1/8 of the store bytes (4.5 B per element)
if PK:
    lib.upipe_set_pk(1)
    VAR = [('flat', 1, 32, 3), ('flat', 1, 16, 6), ('pipe', 2, 16, 3), ('pipe', 2, 20, 3), ('pipe', 2, 16, 2), ('pipe', 2, 8, 6), ('pipe', 2, 24, 2)]
for (N, C, hw) in shapes:
    n = N * C * hw * hw
    x = torch.randn(n, device='cuda'); y = torch.empty_like(x)
    print('tensor [%d,%d,%d,%d] %.0f MB   (GB/s of read+write; columns: delay 0 / 4 / 8 us)' % (N, C, hw, hw, n * 4 / 1e6))
    for alu in (0, 1):
        for (name, kind, K, occ) in VAR:
            r = []
            for d in (0, 400, 800):
                ms = lib.upipe(kind, K, occ, x.data_ptr(), y.data_ptr(), N, C, hw * hw, d, alu, 5)
                r.append(n * (4.5 if PK else 8) / ms / 1e6 if ms > 0 else -1)
            print('  alu=%d %-4s K=%2d occ=%d : %6.0f %6.0f %6.0f' % (alu, name, K, occ, r[0], r[1], r[2]), flush=True)
    del x, y
