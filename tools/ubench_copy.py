import ctypes, os, itertools
import torch
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_copy'))
lib.ucopy.restype = ctypes.c_float
lib.ucopy.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
lib.umemcpy.restype = ctypes.c_float
lib.umemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
n = 512 * 64 * 112 * 112
x = torch.randn(n, device='cuda'); y = torch.empty_like(x)
print('hipMemcpy D2D: %.0f GB/s' % (n * 8 / lib.umemcpy(x.data_ptr(), y.data_ptr(), n * 4, 10) / 1e6))
best = []
for tpb, B, nt, contig in itertools.product((256, 512, 1024), (1, 2, 4, 8), (0, 2, 3), (0, 1)):
    row = []
    for grid in (1024, 2048, 4096, 8192, 32768, (n // 4 + tpb * B - 1) // (tpb * B)):
        ms = lib.ucopy(tpb, B, nt, contig, x.data_ptr(), y.data_ptr(), n // 4, grid, 5)
        row.append(n * 8 / ms / 1e6)
    best.append((max(row), tpb, B, nt, contig, row))
best.sort(reverse=True)
for b in best[:12]:
    print('%.0f GB/s  tpb=%d B=%d nt=%d contig=%d  grids(1k,2k,4k,8k,32k,exact): %s' % (b[0], b[1], b[2], b[3], b[4], ' '.join('%.0f' % v for v in b[5])))
print('worst: %.0f' % best[-1][0])
