"""Round 4: store cache-policy bits in the sample-strided order (tools/ubench_stpol.hip); GB/s of the bytes moved."""
import ctypes, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_stpol'))
lib.ustpol.restype = ctypes.c_float
lib.ustpol.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 6
NAMES = ['plain', 'nt', 'sc0', 'sc1', 'sc0 sc1', 'sc0 nt', 'sc1 nt', 'sc0 sc1 nt']
for (N, C, hw) in ((512, 256, 56), (512, 1024, 14)):
    P4 = C * hw * hw // 4
    x = torch.randn(N * P4 * 4, device='cuda'); y = torch.empty_like(x)
    nb = x.numel() * 4
    print('tensor [%d,%d,%d,%d]: 128 KB tiles (32 rows x 4 KB); address order / sample-strided order' % (N, C, hw, hw), flush=True)
    for pol in range(8):
        line = '  stores %-11s' % NAMES[pol]
        for mode, nm, mult in ((2, 'write', 1), (0, 'copy', 2)):
            r = [nb * mult / lib.ustpol(x.data_ptr(), y.data_ptr(), N, P4, order, mode, pol, 4) / 1e6 for order in (0, 1)]
            line += '   %s %5.0f / %5.0f' % (nm, r[0], r[1])
        print(line, flush=True)
    del x, y
