"""Round 4: stores released in dispatch order through a ticket (tools/ubench_ticket.hip).  us per tensor and TB/s of the bytes moved."""
import ctypes, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_ticket'))
lib.utick.restype = ctypes.c_float
lib.utick.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 8
lib.utick_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
cnt = ctypes.c_void_p()
assert lib.utick_alloc(1 << 20, ctypes.byref(cnt)) == 0
out = torch.zeros(16, device='cuda')
for (N, C, hw) in ((512, 256, 56), (512, 64, 112), (512, 512, 28)):
    HW = hw * hw
    x = torch.randn(N * C * HW, device='cuda'); y = torch.empty_like(x)
    nb = x.numel() * 4
    print('tensor [%d,%d,%d,%d] %.0f MB' % (N, C, hw, hw, nb / 1e6), flush=True)
    for mode, nm, mult in ((0, 'write only', 1), (2, 'write only + meeting', 1), (1, 'copy', 2), (3, 'copy + meeting', 2)):
        for cb in (1, 4, 16):
            Gs = (N * HW // 4 + 8191) // 8192
            if (mode & 2) and cb > 1 and cb * Gs > 700:
                continue
            for nt in (1, 0):
                line = '  %-22s cb=%2d %s:' % (nm, cb, 'nt' if nt else 'pl')
                for W in (0, 8, 16, 32, 64, 128, 256, 512):
                    ms = lib.utick(x.data_ptr(), y.data_ptr(), out.data_ptr(), cnt, N, C, HW, cb, W, mode, nt, 3)
                    line += '  W=%d %.0f us %.2f' % (W, ms * 1e3, nb * mult / ms / 1e9)
                print(line, flush=True)
    del x, y
