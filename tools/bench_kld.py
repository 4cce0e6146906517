#!/usr/bin/env python3
"""Timing of the KLD calibration kernels (development aid): per-sample 2001-bin histogram + 994-candidate
search on ResNet-50-shaped activations.  The reference does this on the host in Python/numpy loops."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import laplace_activation  # noqa: E402
from cnn_quantization_amd import _lib, ops  # noqa: E402


def main():
    dev = torch.device('cuda')
    for shape in ((512, 64, 112, 112), (512, 256, 56, 56), (512, 2048, 7, 7), (32, 64, 112, 112)):
        x = laplace_activation(shape, 7, dev)
        rows, n = shape[0], x.numel()
        ops.kld_thresholds(x)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        lib = _lib.load()
        rowmm = ops.tensor_row_stats(x, rows)
        hist = torch.empty((rows, 2001), dtype=torch.int32, device=dev)
        div = torch.empty((rows, 994), dtype=torch.float64, device=dev)
        out = torch.empty((rows, 3), dtype=torch.float64, device=dev)
        st = ops._stream(x)
        ev[0].record()
        rowmm = ops.tensor_row_stats(x, rows)
        ev[1].record()
        lib.cnnq_kld_hist(ops._ptr(x), rows, n // rows, ops._ptr(rowmm), ops._ptr(hist), st)
        ev[2].record()
        lib.cnnq_kld_search(ops._ptr(hist), rows, ops._ptr(rowmm), ops._ptr(div), ops._ptr(out), st)
        ev[3].record()
        torch.cuda.synchronize()
        t = [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
        print('%-22s rows min/max %.3f ms (%.0f GB/s)  hist %.3f ms (%.0f GB/s)  search+pick %.3f ms   kld_th %.5f' % (
            shape, t[0], n * 4 / t[0] / 1e6, t[1], n * 4 / t[1] / 1e6, t[2], out[:, 0].max().item()))
        del x


if __name__ == '__main__':
    main()
