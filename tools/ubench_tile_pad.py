"""Does the sample stride matter for the tall resident window?  tools/ubench_tile.hip with R = 32 rows x 4 KB tiles in
row-major block order (consecutive workgroups = consecutive samples of the same columns: what a per-channel exchange
forces) and in column-major order, on [512, P] tensors whose row length P is padded by a few hundred bytes - if rows that
are multiples of 64 KB apart collide on the same HBM channels / banks, an odd stride should be faster."""
import ctypes, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_tile'))
lib.utile.restype = ctypes.c_float
lib.utile.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
N = 512
for base in (256 * 56 * 56 // 4, 1024 * 14 * 14 // 4):
    for pad4 in (0, 16, 64, 80, 272, 1040, 4112, 16400):          # float4 units: 256 B, 1 KB, 1.25 KB, 4.25 KB, 16.25 KB, ...
        P4 = base + pad4
        x = torch.randn(N * P4 * 4, device='cuda'); y = torch.empty_like(x)
        r = [x.numel() * 8 / lib.utile(1, 32, order, 1, x.data_ptr(), y.data_ptr(), N, P4, 5) / 1e6 for order in (0, 1)]
        print('row %9d B (pad %6d B): column-major blocks %5.0f GB/s, row-major blocks %5.0f GB/s' % (P4 * 16, pad4 * 16, r[0], r[1]), flush=True)
        del x, y
