"""Round 4: workgroup order / row ownership inside a co-resident block (tools/ubench_order.hip); GB/s of the bytes moved."""
import ctypes, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_order'))
lib.uord.restype = ctypes.c_float
lib.uord.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 8
for (N, C, hw) in ((512, 256, 56), (512, 1024, 14)):
    P4 = C * hw * hw // 4
    x = torch.randn(N * P4 * 4, device='cuda'); y = torch.empty_like(x)
    nb = x.numel() * 4
    ncol = P4 // 256
    print('tensor [%d,%d,%d,%d] %.0f MB, %d column blocks of 4 KB per sample; 32-row tiles (128 KB)' % (N, C, hw, hw, nb / 1e6, ncol), flush=True)
    for G in [g for g in (1, 2, 4, 7, 14, 16, 28, 49, 98, 196, 392, 784) if ncol % g == 0]:
        line = '  block of %3d column blocks (%5d KB per sample, %4d workgroups):' % (G, G * 4, G * N // 32)
        for mode, nm, mult in ((0, 'copy', 2), (2, 'write', 1)):
            for order in (0, 1):
                for rows in (0, 1):
                    ms = lib.uord(x.data_ptr(), y.data_ptr(), N, P4, G, order, rows, mode, 1, 4)
                    line += '  %s %s/%s %5.0f' % (nm, 'colfast' if order == 0 else 'rowfast', 'contig' if rows == 0 else 'interl', nb * mult / ms / 1e6)
        print(line, flush=True)
    del x, y
