"""Round 4: read-only / write-only / copy rates of register tiles by tile shape and workgroup order (tools/ubench_rw.hip).
GB/s of the bytes each mode moves."""
import ctypes, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_rw'))
lib.urw.restype = ctypes.c_float
lib.urw.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3
out = torch.zeros(16, device='cuda')
for (N, C, hw) in ((512, 256, 56), (512, 1024, 14)):
    P4 = C * hw * hw // 4
    x = torch.randn(N * P4 * 4, device='cuda'); y = torch.empty_like(x)
    nb = x.numel() * 4
    print('tensor [%d,%d,%d,%d] %.0f MB; GB/s as: address order / sample-strided order' % (N, C, hw, hw, nb / 1e6))
    for (J, R) in ((1, 1), (1, 4), (4, 1), (1, 8), (1, 32), (2, 16), (4, 8), (8, 4), (16, 2), (32, 1)):
        line = '  J=%2d (segment %6d B) R=%2d tile %4d KB:' % (J, J * 4096, R, J * R * 4)
        for mode, nm, mult in ((1, 'read', 1), (2, 'write', 1), (0, 'copy', 2)):
            for nt in (1, 0):
                r = []
                for order in (0, 1):
                    ms = lib.urw(J, R, order, mode, nt, x.data_ptr(), y.data_ptr(), out.data_ptr(), N, P4, 4)
                    r.append(nb * mult / ms / 1e6)
                line += '  %s%s %5.0f / %5.0f' % (nm, ' nt' if nt else ' pl', r[0], r[1])
        print(line, flush=True)
    del x, y
