#!/usr/bin/env python3
"""Exhaustive check of the three-instruction divide (tools/ubench_divide.hip): all 2^32 dividends against batches of
scales - random ones over the range the path produces (1e-8 ... 1e3), powers of two, scales with an all-ones
significand (the theorem's exception), the scale floor - for 4- and 8-bit codes."""
import ctypes
import os
import subprocess
import sys
import time

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, 'ubench_divide.so')
if not os.path.exists(so):
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-shared', '-fPIC',
                    os.path.join(here, 'ubench_divide.hip'), '-o', so], check=True)
lib = ctypes.CDLL(so)
lib.udivide.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
NS = int(os.environ.get('NS', '512'))
g = torch.Generator().manual_seed(7)


def run(name, scales, qmax):
    scales = scales.float().contiguous().cuda()
    zps = torch.randint(0, int(qmax) + 1, (scales.numel(),), generator=g).float().cuda()
    counts = torch.zeros(4, dtype=torch.int64, device='cuda')
    t0 = time.time()
    rc = lib.udivide(scales.data_ptr(), zps.data_ptr(), scales.numel(), float(qmax), counts.data_ptr())
    assert rc == 0
    c = counts.cpu().tolist()
    n = scales.numel() * 2 ** 32
    print('%-34s qmax %3d  %4d scales x 2^32 dividends (%.1f s): quotient != IEEE %d (%.2e; %d of them normal), '
          'codes differ %d, mid-tread outputs differ %d' % (name, qmax, scales.numel(), time.time() - t0, c[0], c[0] / n,
                                                           c[3], c[1], c[2]), flush=True)
    return c


tot = [0, 0, 0, 0]
for qmax in (15.0, 255.0):
    rnd = torch.exp(torch.empty(NS).uniform_(-18.4, 6.9, generator=g))                       # 1e-8 ... 1e3
    ones = torch.arange(-40, 40).float().exp2() * (2 - 2.0 ** -23)                           # EVERY 1.11...1 x 2^k in range
    pow2 = torch.arange(-27, 10).float().exp2()
    for name, sc in (('random scales', rnd), ('all-ones significands', ones), ('powers of two', pow2),
                     ('scale floor 1e-8', torch.tensor([1e-8]))):
        c = run(name, sc, qmax)
        tot = [a + b for a, b in zip(tot, c)]
print('total: quotient mismatches %d (normal range: %d), code mismatches %d, mid-tread mismatches %d' % (tot[0], tot[3], tot[1], tot[2]))
