#!/usr/bin/env python3
"""Exhaustive check of the divide-free exact quotient of the single-launch kernels (qdq1_fast; tools/ubench_divide.hip):
all 2^32 dividends against batches of scales - random ones over the whole domain (1e-8 ... 2^30), every power of two,
every all-ones significand, the scale floor - with zero points 0, small, and far from zero, for 4- and 8-bit codes.
Counts quotient mismatches (2^-70 <= |x| <= 2^70) and (code, y) mismatches (all |x| <= 2^70) of the five-operation
form the kernels use and of the three-operation form without the second correction."""
import ctypes
import os
import subprocess
import sys
import time

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = __import__('_ubuild').so('ubench_divide', extra=('-ffp-contract=off',))
lib = ctypes.CDLL(so)
lib.udivide.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
NS = int(os.environ.get('NS', '384'))
g = torch.Generator().manual_seed(7)


def run(name, scales, qmax):
    scales = scales.float().contiguous().cuda()
    n = scales.numel()
    zps = torch.randint(0, int(qmax) + 1, (n,), generator=g).float()
    far = torch.randint(0, 4, (n,), generator=g)
    zps = torch.where(far == 0, torch.zeros(n), torch.where(far == 1, -torch.randint(1000, 4000000, (n,), generator=g).float(), zps)).cuda()
    counts = torch.zeros(5, dtype=torch.int64, device='cuda')
    t0 = time.time()
    rc = lib.udivide(scales.data_ptr(), zps.data_ptr(), n, float(qmax), counts.data_ptr())
    assert rc == 0
    c = counts.cpu().tolist()
    print('%-26s qmax %3d  %4d scales, %.3e dividends in the domain (%.1f s): quotient != x / s: five-op %d, three-op %d;  '
          '(code, y) differ: five-op %d, three-op %d' % (name, qmax, n, c[4], time.time() - t0, c[0], c[1], c[2], c[3]), flush=True)
    return c


tot = [0] * 5
for qmax in (15.0, 255.0):
    rnd = torch.exp(torch.empty(NS).uniform_(-18.4, 20.79, generator=g)).clamp(1e-8, 2.0 ** 30)   # 1e-8 ... 2^30
    ones = torch.arange(-26, 30).float().exp2() * (2 - 2.0 ** -23)                                 # EVERY 1.11...1 x 2^k in the domain
    pow2 = torch.arange(-26, 31).float().exp2()
    for name, sc in (('random scales', rnd), ('all-ones significands', ones), ('powers of two', pow2),
                     ('scale floor 1e-8', torch.tensor([1e-8]))):
        c = run(name, sc, qmax)
        tot = [a + b for a, b in zip(tot, c)]
print('total over %.3e (dividend, scale) pairs: quotient mismatches five-op %d, three-op %d; (code, y) mismatches five-op %d, '
      'three-op %d' % (tot[4], tot[0], tot[1], tot[2], tot[3]))
