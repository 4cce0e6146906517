"""Round 4: register-tile copy / write-only rate by the WIDTH of a workgroup's row pieces (tools/ubench_width.hip): 245 float4
(5 channels of 14x14: 3920 bytes, off the 64-byte grid), 256 (4096), 252 / 248 / 240 (multiples of 64 / 128 bytes), on the
[512,1024,14,14] and [512,2048,7,7] layouts.  TB/s of the bytes each mode moves."""
import ctypes, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_width'))
lib.uwidth.restype = ctypes.c_float
lib.uwidth.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4
for (N, C, hw) in ((512, 1024, 14), (512, 2048, 7)):
    P4 = C * hw * hw // 4
    x = torch.randn(N * P4 * 4, device='cuda'); y = torch.empty_like(x)
    nb = x.numel() * 4
    print('tensor [%d,%d,%d,%d] %.0f MB, %d float4 per sample' % (N, C, hw, hw, nb / 1e6, P4))
    for rnd in range(2):
        for w in (245, 256, 252, 248, 240, 196, 192):
            r = []
            for mode, mult in ((0, 2), (2, 1)):
                ms = lib.uwidth(mode, x.data_ptr(), y.data_ptr(), N, P4, w, 6)
                r.append(nb * mult / ms / 1e9)
            print('  w=%3d (%4d B, %s the 64-byte grid): copy %.2f  write only %.2f' % (w, w * 16, 'on' if (w * 16) % 64 == 0 else 'OFF', r[0], r[1]), flush=True)
    del x, y
