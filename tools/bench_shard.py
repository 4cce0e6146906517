#!/usr/bin/env python3
"""Per-layer time of the config-2 hot call (ops.act_qdq_per_channel, the route the product picks) at a given batch - the
64-sample shard of the 8-GPU run by default - over rotating distinct buffers, HIP events around back-to-back calls.

    python tools/bench_shard.py [--batch 64] [--tag name]
    CNNQ_GRP_K=16 python tools/bench_shard.py --tag K16      (development builds: kernel sweeps, one process per setting)

One line per layer shape: us per call, TB/s on the 8 bytes per element actually moved, and the plan; a last line with the
per-forward sum (53 tensors).  `--json` appends a machine-readable record to gpurun_out/r5/shard_sweep.jsonl."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from cnn_quantization_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--reps', type=int, default=40)
    ap.add_argument('--rotate-mb', type=int, default=1200)
    ap.add_argument('--tag', type=str, default='default')
    ap.add_argument('--json', action='store_true')
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device('cuda')
    total = 0.
    rec = {'tag': a.tag, 'batch': a.batch, 'layers': {}}
    for (C, hw, half, count) in bench.RESNET50_CONV_OUTPUTS:
        N, HW = a.batch, hw * hw
        n = N * C * HW
        nbuf = max(2, min(24, (a.rotate_mb << 20) // (8 * n) + 1))
        xs = [bench.laplace_activation((N, C, hw, hw), 100 + i, dev) for i in range(nbuf)]
        ys = [torch.empty_like(xs[0]) for _ in range(nbuf)]
        w = (ctypes.c_int32 * 8)()
        g = (ctypes.c_int32 * 8)()
        rw = lib.cnnq_pc_resident_describe(N, C, HW, w)
        rg = lib.cnnq_pc_group_describe(N, C, HW, g)
        for i in range(nbuf):
            ops.act_qdq_per_channel(xs[i], 4, positive=half, out=ys[i])
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(a.reps):
                ops.act_qdq_per_channel(xs[r % nbuf], 4, positive=half, out=ys[r % nbuf])
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / a.reps
            best = t if best is None else min(best, t)
        plan = ('whole A=%d T=%d K=%d k=%d wgs=%d' % (w[0], w[1], w[2], w[3], w[6]) if rw == 0 else 'whole -') + ' | ' + (
            'group A=%d K=%d mode=%d Gs=%d wgs=%d' % (g[0], g[1], g[2], g[5], g[7]) if rg == 0 else 'group -')
        print('%-4s C=%4d %3dx%-3d half=%d x%2d  %7.1f us  %5.2f TB/s  %s' % (a.tag, C, hw, hw, half, count, best * 1e6, n * 8 / best / 1e12, plan),
              flush=True)
        rec['layers']['%dx%d%s' % (C, hw, '+' if half else '')] = best * 1e6
        total += best * count
        del xs, ys
    print('%-4s per forward (sum over 53 tensors, one by one): %.3f ms' % (a.tag, total * 1e3))
    rec['sum_ms'] = total * 1e3
    rec['status'] = ops.group_status(torch.empty(1, device=dev))
    if a.json:
        os.makedirs(os.path.join('gpurun_out', 'r5'), exist_ok=True)
        with open(os.path.join('gpurun_out', 'r5', 'shard_sweep.jsonl'), 'a') as f:
            f.write(json.dumps(rec) + '\n')


if __name__ == '__main__':
    main()
