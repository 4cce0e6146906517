import ctypes, os
import torch
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_policy'))
lib.pcopy.restype = ctypes.c_float
lib.pcopy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
lib.pread.restype = ctypes.c_float
lib.pread.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
names = ['-', 'nt', 'sc1', 'sc0 sc1', 'sc1 nt', 'sc0 sc1 nt', 'sc0', 'sc0 nt']
n = 512 * 64 * 112 * 112
x = torch.randn(n, device='cuda'); y = torch.empty_like(x); o = torch.zeros(4, device='cuda')
print('read-only (GB/s):', '  '.join('%s=%.0f' % (names[l], n * 4 / lib.pread(l, x.data_ptr(), o.data_ptr(), n // 4, 5) / 1e6) for l in range(8)))
print('copy GB/s, rows = load policy, columns = store policy:', names)
for l in range(8):
    print('%-11s' % names[l], ' '.join('%5.0f' % (n * 8 / lib.pcopy(l, s, x.data_ptr(), y.data_ptr(), n // 4, 5) / 1e6) for s in range(8)))
lib.pcopy_buf.restype = ctypes.c_float
lib.pcopy_buf.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
print('nt load + raw buffer store with aux bits (GB/s):', '  '.join('%s=%.0f' % (nm, n * 8 / lib.pcopy_buf(a, x.data_ptr(), y.data_ptr(), n // 4, 5) / 1e6)
      for a, nm in ((0, '-'), (2, 'nt'), (0x10, 'sc1'), (0x11, 'sc0 sc1'), (0x12, 'sc1 nt'), (0x13, 'sc0 sc1 nt'))))
assert torch.equal(x, y)
