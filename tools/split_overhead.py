#!/usr/bin/env python3
"""Cost of cutting each tensor's channels in two halves (the pipelined multi-GPU exchange of
ops._minmax_qdq_pipelined) on ONE GPU with the exchange itself left out: the same launches in the same
order, vs the unsplit sequence.  ResNet-50 b512 layer set."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from cnn_quantization_amd import _lib as L, ops  # noqa: E402
import math  # noqa: E402


def main():
    lib = L.load()
    layers = bench.build_workload(512, torch.device('cuda'))
    elems = sum(l['x'].numel() for l in layers)

    def unsplit():
        for l in layers:
            x, y, N, C, HW = l['x'], l['y'], l['N'], l['C'], l['HW']
            st = ops._stream(x)
            G = lib.cnnq_pc_groups(N, C, HW, 1)
            pmm = torch.empty((G, 2, C), device=x.device); local = torch.empty((1, 2, C), device=x.device)
            qp = torch.empty((3, C), device=x.device)
            lib.cnnq_pc_minmax(ops._ptr(x), N, C, HW, ops._ptr(pmm), st)
            lib.cnnq_pc_minmax_reduce(ops._ptr(pmm), G, C, ops._ptr(local), st)
            lib.cnnq_pc_minmax_params(ops._ptr(local), 1, C, 4, int(l['half']), ops._ptr(qp), st)
            lib.cnnq_pc_qdq(ops._ptr(x), ops._ptr(y), N, C, HW, ops._ptr(qp), None, None, 1, st)

    def split(min_mb=0):
        for l in layers:
            x, y, N, C, HW = l['x'], l['y'], l['N'], l['C'], l['HW']
            st = ops._stream(x)
            if x.numel() * 4 < min_mb << 20:
                G = lib.cnnq_pc_groups(N, C, HW, 1)
                pmm = torch.empty((G, 2, C), device=x.device); local = torch.empty((1, 2, C), device=x.device)
                qp = torch.empty((3, C), device=x.device)
                lib.cnnq_pc_minmax(ops._ptr(x), N, C, HW, ops._ptr(pmm), st)
                lib.cnnq_pc_minmax_reduce(ops._ptr(pmm), G, C, ops._ptr(local), st)
                lib.cnnq_pc_minmax_params(ops._ptr(local), 1, C, 4, int(l['half']), ops._ptr(qp), st)
                lib.cnnq_pc_qdq(ops._ptr(x), ops._ptr(y), N, C, HW, ops._ptr(qp), None, None, 1, st)
                continue
            stride = C * HW
            m = 4 // math.gcd(HW % 4, 4) if HW % 4 else 1
            ca = (C // 2) - (C // 2) % m
            pend = []
            for c0, c1 in ((0, ca), (ca, C)):
                Cs = c1 - c0
                G = lib.cnnq_pc_groups(N, Cs, HW, ops._slice_aligned(x, c0, HW, stride))
                pmm = torch.empty((G, 2, Cs), device=x.device); local = torch.empty((1, 2, Cs), device=x.device)
                lib.cnnq_pc_minmax_strided(ops._slice_ptr(x, c0, HW), N, Cs, HW, stride, ops._ptr(pmm), st)
                lib.cnnq_pc_minmax_reduce(ops._ptr(pmm), G, Cs, ops._ptr(local), st)
                pend.append((c0, Cs, local))
            for c0, Cs, local in pend:
                qp = torch.empty((3, Cs), device=x.device)
                lib.cnnq_pc_minmax_params(ops._ptr(local), 1, Cs, 4, int(l['half']), ops._ptr(qp), st)
                lib.cnnq_pc_qdq_strided(ops._slice_ptr(x, c0, HW), ops._slice_ptr(y, c0, HW), N, Cs, HW, stride,
                                        ops._ptr(qp), None, None, 1, st)

    for name, fn in (('unsplit', unsplit), ('split', split), ('split>=150MB', lambda: split(150)),
                     ('split>=300MB', lambda: split(300)), ('split>=600MB', lambda: split(600)), ('unsplit', unsplit)):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print('%-13s %.2f ms/forward  %.1f G elem/s' % (name, best * 1e3, elems / best / 1e9))


if __name__ == '__main__':
    main()
