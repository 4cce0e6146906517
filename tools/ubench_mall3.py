"""Round 4 Infinity-Cache experiment (tools/ubench_mall3.hip): statistics pass and Q/DQ pass chunk by chunk as ONE-SHOT
short workgroups in address order - in one launch (with and without the real counter dependency) and as separate
launches.  Prints us per tensor and TB/s on the 8 B/elem accounting (x bytes * 2 / time)."""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(__import__('_ubuild').so('ubench_mall3'))
P_, I_, L_ = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
lib.um3_one.restype = ctypes.c_float
lib.um3_one.argtypes = [P_] * 5 + [I_, L_, L_] + [I_] * 5
lib.um3_sep.restype = ctypes.c_float
lib.um3_sep.argtypes = [P_] * 5 + [I_, L_, L_] + [I_] * 4
lib.um3_pass.restype = ctypes.c_float
lib.um3_pass.argtypes = [P_] * 3 + [I_, L_] + [I_] * 3
lib.um3_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
out = torch.zeros(16, device='cuda')
done = ctypes.c_void_p(); part = ctypes.c_void_p()
assert lib.um3_alloc(1 << 20, ctypes.byref(done)) == 0
assert lib.um3_alloc(64 << 20, ctypes.byref(part)) == 0
quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
POL = {0: 'all plain', 1: 'A plain, B nt/nt', 2: 'all nt', 3: 'A plain, B plain ld / nt st', 4: 'A plain, B nt ld / plain st'}
for (N, C, hw) in ((512, 256, 56), (512, 64, 112), (512, 512, 28)):
    P = C * hw * hw
    x = torch.randn(N * P, device='cuda'); y = torch.empty_like(x)
    nb = x.numel() * 4
    def tbs(ms):
        return nb * 2 / ms / 1e9 if ms > 0 else -1
    print('tensor [%d,%d,%d,%d] %.0f MB' % (N, C, hw, hw, nb / 1e6), flush=True)
    for kind, nm in ((0, 'pass A alone (read)'), (1, 'pass B alone (read+write)')):
        for nt in (0, 1):
            ms = lib.um3_pass(x.data_ptr(), y.data_ptr(), out.data_ptr(), N, P, kind, nt, 5)
            print('  %-28s nt=%d : %7.1f us  (%.2f TB/s of its own bytes)' % (nm, nt, ms * 1e3, nb * (1 + kind) / ms / 1e9), flush=True)
    P4 = P // 4
    ks = [k for k in (1, 2, 4, 7, 8, 14, 16, 28, 49) if (P4 // 1024) % k == 0]
    for k in ks:
        cf = 4096 * k
        mb = N * cf * 4 / 2**20
        if mb > 200:
            continue
        for pol in ((1, 2, 3, 0, 4) if not quick else (1,)):
            r = []
            for (lag, dep) in ((1, 0), (2, 0), (1, 1), (2, 1)):
                ms = lib.um3_one(x.data_ptr(), y.data_ptr(), out.data_ptr(), done, part, N, P, cf, 4, lag, dep, pol, 4)
                r.append((ms * 1e3, tbs(ms)))
            ms = lib.um3_sep(x.data_ptr(), y.data_ptr(), out.data_ptr(), done, part, N, P, cf, 4, 1, pol, 3) if pol in (1, 2, 3) else -1
            print('  chunk %6.1f MB (runs %4d KB) %-28s one launch lag1 %6.1f us %5.2f | lag2 %6.1f %5.2f | dep lag1 %6.1f %5.2f | dep lag2 %6.1f %5.2f | separate launches %6.1f %5.2f' % (
                mb, cf * 4 // 1024, POL[pol], r[0][0], r[0][1], r[1][0], r[1][1], r[2][0], r[2][1], r[3][0], r[3][1], ms * 1e3, tbs(ms)), flush=True)
    # 32 KB tiles (8 loads per lane), the best policy only
    for k in [k for k in (2, 4, 8, 14, 28) if (P4 // 2048) % k == 0 and N * 8192 * k * 4 / 2**20 <= 200]:
        cf = 8192 * k
        r = []
        for (lag, dep) in ((1, 0), (2, 0), (1, 1), (2, 1)):
            ms = lib.um3_one(x.data_ptr(), y.data_ptr(), out.data_ptr(), done, part, N, P, cf, 8, lag, dep, 1, 4)
            r.append((ms * 1e3, tbs(ms)))
        print('  L=8 chunk %6.1f MB A plain, B nt/nt: lag1 %6.1f us %5.2f | lag2 %6.1f %5.2f | dep lag1 %6.1f %5.2f | dep lag2 %6.1f %5.2f' % (
            N * cf * 4 / 2**20, r[0][0], r[0][1], r[1][0], r[1][1], r[2][0], r[2][1], r[3][0], r[3][1]), flush=True)
    del x, y
