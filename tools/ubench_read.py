import ctypes, os
import torch
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_copy'))
lib.uread.restype = ctypes.c_float
lib.uread.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
n = 512 * 64 * 112 * 112
xs = [torch.randn(n, device='cuda') for _ in range(2)]
out = torch.empty(1 << 22, device='cuda')
for B in (1, 4, 8):
    for nt in (0, 1):
        row = []
        for grid in (2048, 4096, 16384, 65536, (n // 4 + 256 * B - 1) // (256 * B)):
            ms = lib.uread(B, nt, xs[0].data_ptr(), out.data_ptr(), n // 4, grid, 5)
            row.append('%5.0f' % (n * 4 / ms / 1e6))
        print('read-only B=%d nt=%d grids(2k,4k,16k,64k,exact): %s GB/s' % (B, nt, ' '.join(row)))
