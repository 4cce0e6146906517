"""Round 4: does the ALLOCATION matter?  The streaming / register-tile microbenchmark of tools/ubench_rw.hip over buffers
from hipMalloc, from hipExtMallocWithFlags(hipDeviceMallocContiguous) and from torch's caching allocator, one pair (warm
address translations) and eight pairs round-robin (26 GB: cold translations, as the 53 tensors of the bench are)."""
import ctypes, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_rw'))
P_, I_ = ctypes.c_void_p, ctypes.c_int
lib.urw_multi.restype = ctypes.c_float
lib.urw_multi.argtypes = [I_] * 4 + [ctypes.POINTER(P_)] * 2 + [I_, P_, I_, I_, I_]
lib.urw_alloc.argtypes = [ctypes.c_size_t, I_, ctypes.POINTER(P_)]
lib.urw_free.argtypes = [P_]
lib.urw_fill.argtypes = [P_, ctypes.c_size_t]
out = torch.zeros(16, device='cuda')
N, P4 = 512, 256 * 56 * 56 // 4
nb_bytes = N * P4 * 16
NB = 8
for kind, flags in (('hipMalloc', -1), ('hipExtMallocWithFlags(Contiguous)', 4), ('torch.empty', None)):
    xs, ys, keep = [], [], []
    ok = True
    for i in range(2 * NB):
        if flags is None:
            t = torch.empty(nb_bytes // 4, device='cuda'); t.fill_(1.5); keep.append(t); p = t.data_ptr()
        else:
            q = P_()
            rc = lib.urw_alloc(nb_bytes, flags, ctypes.byref(q))
            if rc != 0:
                print('%s: allocation failed (hipError %d)' % (kind, rc)); ok = False; break
            lib.urw_fill(q, nb_bytes); p = q.value
        (xs if i % 2 == 0 else ys).append(p)
    if ok:
        X = (P_ * NB)(*xs); Y = (P_ * NB)(*ys)
        line = '%-36s' % kind
        for (J, R, order, nm) in ((1, 1, 0, '4 KB address order'), (1, 32, 0, '128 KB address order'), (1, 32, 1, '128 KB sample-strided')):
            for mode, mn, mult in ((1, 'read', 1), (2, 'write', 1), (0, 'copy', 2)):
                warm = lib.urw_multi(J, R, order, mode, X, Y, 1, out.data_ptr(), N, P4, 6)
                cold = lib.urw_multi(J, R, order, mode, X, Y, NB, out.data_ptr(), N, P4, 2)
                line += ' | %s %s warm %.2f cold %.2f' % (nm, mn, nb_bytes * mult / warm / 1e9, nb_bytes * mult / cold / 1e9)
        print(line, flush=True)
    if flags is not None:
        for p in xs + ys:
            lib.urw_free(P_(p))
    del keep
    torch.cuda.empty_cache()
