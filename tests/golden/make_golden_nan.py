#!/usr/bin/env python3
"""NaN / inf golden case for config 2 (SURVEY.md 8 c3 "NaN policy"), recorded by RUNNING THE REFERENCE ITSELF
(same import recipe as make_golden.py; build container only).  torch.min / torch.max propagate NaN
(int_quantizer.py:416,423), so a NaN activation poisons exactly its channel; +-inf give an infinite range.

    python tests/golden/make_golden_nan.py        # rewrites tests/golden/nan.npz
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G  # noqa: E402  (imports the reference with the int_quantization stub)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    torch.set_num_threads(1)
    gen = torch.Generator().manual_seed(909)
    d, k = {}, 0
    for shape in ((6, 8, 14, 14), (5, 12, 7, 7), (9, 3, 56, 56), (34, 3, 28, 28)):
        x = G.laplace_nchw(gen, shape)
        x[1, 2, 3, 4] = float('nan')          # channel 2: NaN
        x[0, 0, 0, 0] = float('inf')          # channel 0: +inf
        x[2, 1, 1, 1] = float('-inf')         # channel 1: -inf
        for half in (False, True):
            q = G.iq.int_quantizer('int4', G.params())
            q.half_range = half
            y = q(x, 'conv0_activation', 'activation')
            d['c%d_x' % k], d['c%d_y' % k], d['c%d_half' % k] = x, y, np.int64(half)
            k += 1
    d['n_cases'] = np.int64(k)
    G.save('nan', d)


if __name__ == '__main__':
    main()
