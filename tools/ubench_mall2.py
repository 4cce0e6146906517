import ctypes, os, torch
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_mall'))
lib.umall_two.restype = ctypes.c_float
lib.umall_two.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong] + [ctypes.c_int] * 4
lib.umall_read.restype = ctypes.c_float
lib.umall_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
out = torch.zeros(16, device='cuda')
N, P = 32, 64 * 112 * 112
x = torch.randn(N * P, device='cuda'); y = torch.empty_like(x)
print('tensor %.0f MB' % (x.numel() * 4 / 1e6))
for a_nt in (0, 1):
    for grid in (512, 1024, 2048, 4096):
        ms = lib.umall_two(x.data_ptr(), y.data_ptr(), out.data_ptr(), N, P, P, 0, a_nt, grid, 5)
        print('two sweeps in one launch (persistent loop), statsNT=%d grid=%d: %.1f us' % (a_nt, grid, ms * 1e3))
for nt in (0, 1):
    ms = lib.umall_read(x.data_ptr(), out.data_ptr(), x.numel() * 4, 2, nt, 4096, 5)
    print('read the tensor twice in one launch nt=%d: %.1f us (%.0f GB/s)' % (nt, ms * 1e3, x.numel() * 8 / ms / 1e6))
    ms = lib.umall_read(x.data_ptr(), out.data_ptr(), x.numel() * 4, 1, nt, 4096, 5)
    print('read the tensor once  in one launch nt=%d: %.1f us (%.0f GB/s)' % (nt, ms * 1e3, x.numel() * 4 / ms / 1e6))
