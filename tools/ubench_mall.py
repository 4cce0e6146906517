"""Infinity-Cache experiment (tools/ubench_mall.hip): (1) what a cache-resident re-read sweep runs at, by buffer size;
(2) statistics pass and Q/DQ pass chunk by chunk inside one launch, by chunk size - does the second read of x leave HBM?"""
import ctypes, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_mall'))
lib.umall_read.restype = ctypes.c_float
lib.umall_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
lib.umall_two.restype = ctypes.c_float
lib.umall_two.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong] + [ctypes.c_int] * 4
out = torch.zeros(16, device='cuda')
N, C, hw = 512, 256, 56
P = C * hw * hw
x = torch.randn(N * P, device='cuda'); y = torch.empty_like(x)
print('read-only sweep repeated 8x inside one launch (GB/s):')
for mb in (32, 64, 128, 192, 256, 384, 512, 1024):
    r = []
    for nt in (0, 1):
        ms = lib.umall_read(x.data_ptr(), out.data_ptr(), mb << 20, 8, nt, 4096, 3)
        r.append((mb << 20) * 8 / ms / 1e6)
    print('  %5d MB: plain %6.0f  nt %6.0f' % (mb, r[0], r[1]), flush=True)
print('two passes chunk by chunk in one launch, tensor [%d,%d,%d,%d] %.0f MB; GB/s on the 8 B/elem accounting (x bytes * 2 / time):' % (N, C, hw, hw, x.numel() * 4 / 1e6))
for k in (2, 4, 7, 14, 28, 49, 196):
    cf = 4096 * k
    for order in (0, 1):
        for a_nt in (0, 1):
            r = []
            for grid in (1024, 2048, 4096):
                ms = lib.umall_two(x.data_ptr(), y.data_ptr(), out.data_ptr(), N, P, cf, order, a_nt, grid, 3)
                r.append(x.numel() * 8 / ms / 1e6 if ms > 0 else -1)
            print('  chunk %5.0f MB (runs of %4d KB) order=%d statsNT=%d : grids 1024/2048/4096: %6.0f %6.0f %6.0f' % (
                N * cf * 4 / 2**20, cf * 4 // 1024, order, a_nt, r[0], r[1], r[2]), flush=True)
