"""Round 4: sample-strided register-tile copy / write with the workgroups' store loops in phase (all start at row 0) or
staggered (tools/ubench_phase.hip), on several (x, y) pairs of one process (placement differs per pair).  TB/s."""
import ctypes, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_phase'))
lib.uphase.restype = ctypes.c_float
lib.uphase.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 3
N, C, hw = 512, 256, 56
P4 = C * hw * hw // 4
pairs = [(torch.randn(N * P4 * 4, device='cuda'), torch.empty(N * P4 * 4, device='cuda')) for _ in range(4)]
nb = N * P4 * 16
for k, (x, y) in enumerate(pairs):
    for rnd in range(2):
        line = 'pair %d:' % k
        for mode, mult, nm in ((0, 2, 'copy'), (2, 1, 'write')):
            for rot in (0, 1, 2):
                ms = lib.uphase(mode, rot, x.data_ptr(), y.data_ptr(), N, P4, 6)
                line += '  %s rot%d %.2f' % (nm, rot, nb * mult / ms / 1e9)
        print(line, flush=True)
