"""Which kind of box is this?  Write-only and copy rates of 128 KB register tiles in address order and in the sample-strided
order (tools/ubench_rw.so): on some boxes of the pool the strided order costs the stores 20-30 %, on others nothing."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_rw'))
lib.urw.restype = ctypes.c_float
lib.urw.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3
out = torch.zeros(16, device='cuda')
N, P4 = 512, 256 * 56 * 56 // 4
x = torch.randn(N * P4 * 4, device='cuda'); y = torch.empty_like(x)
nb = x.numel() * 4
r = {}
for mode, nm, mult in ((1, 'read', 1), (2, 'write', 1), (0, 'copy', 2)):
    for order in (0, 1):
        ms = lib.urw(1, 32, order, mode, 1, x.data_ptr(), y.data_ptr(), out.data_ptr(), N, P4, 6)
        r[(nm, order)] = nb * mult / ms / 1e9
ms = lib.urw(1, 1, 0, 0, 1, x.data_ptr(), y.data_ptr(), out.data_ptr(), N, P4, 6)
print('box %s: streaming copy %.2f TB/s | 128 KB tiles, address order / sample-strided: read %.2f / %.2f, write %.2f / %.2f, copy %.2f / %.2f' % (
    bench.box_id(), nb * 2 / ms / 1e9, r[('read', 0)], r[('read', 1)], r[('write', 0)], r[('write', 1)], r[('copy', 0)], r[('copy', 1)]))
