"""Per-layer timing of config 2 on the ResNet-50 layer set: the register-resident single launch
(cnnq_pc_minmax_qdq_resident, where a layer has one) against the three-launch chain (cnnq_pc_minmax_qdq), inputs rotated over enough
distinct buffers that nothing is re-read from the Infinity Cache.  Prints one line per distinct shape and the
per-forward totals; checks that both give the same bits.

    python tools/bench_resident.py [--batch 64] [--reps 20]
    CNNQ_RES_T=1024 python tools/bench_resident.py ...    (kernel sweeps: force the workgroup size)
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from cnn_quantization_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--rotate-mb', type=int, default=700, help='distinct input bytes to rotate over per shape')
    ap.add_argument('--shapes', type=str, default='', help='subset, e.g. "256x14,64x112"')
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device('cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    want = set(args.shapes.split(',')) if args.shapes else None
    tot = {'chain': 0., 'res': 0.}
    elems_total = 0
    print('batch %d, T=%s' % (args.batch, os.environ.get('CNNQ_RES_T', 'auto')))
    for (C, hw, half, count) in bench.RESNET50_CONV_OUTPUTS:
        if want and ('%dx%d' % (C, hw)) not in want:
            continue
        N, HW = args.batch, hw * hw
        n = N * C * HW
        nbuf = max(2, min(16, (args.rotate_mb << 20) // (4 * n) + 1))
        xs = [bench.laplace_activation((N, C, hw, hw), 100 + i, dev) for i in range(nbuf)]
        ys = [torch.empty_like(xs[0]) for _ in range(nbuf)]
        yr = [torch.empty_like(xs[0]) for _ in range(nbuf)]
        G = lib.cnnq_pc_groups(N, C, HW, 1)
        pmm = torch.empty((G, 2, C), dtype=torch.float32, device=dev)
        qp = torch.empty((3, C), dtype=torch.float32, device=dev)
        qp2 = torch.empty((3, C), dtype=torch.float32, device=dev)
        d = (ctypes.c_int32 * 8)()
        rc = lib.cnnq_pc_resident_describe(N, C, HW, d)
        if rc != 0:
            d = [0] * 8

        def chain(i):
            _lib.check(lib.cnnq_pc_minmax_qdq(xs[i].data_ptr(), ys[i].data_ptr(), N, C, HW, 4, int(half), pmm.data_ptr(),
                                              qp.data_ptr(), None, None, st), 'chain')

        def res(i):
            if rc != 0:
                return chain(i)
            _lib.check(lib.cnnq_pc_minmax_qdq_resident(xs[i].data_ptr(), yr[i].data_ptr(), N, C, HW, 4, int(half),
                                                       qp2.data_ptr(), None, st), 'resident')
        times = {}
        for name, fn in (('chain', chain), ('res', res)):
            for i in range(nbuf):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(args.reps):
                fn(r % nbuf)
            e1.record()
            torch.cuda.synchronize()
            times[name] = e0.elapsed_time(e1) * 1e-3 / args.reps
        same = rc != 0 or (all(torch.equal(a, b) for a, b in zip(ys, yr)) and torch.equal(qp, qp2))
        print('C=%4d HW=%5d half=%d x%2d | A=%d T=%4d K=%2d k=%3d CL=%3d RL=%2d wgs=%4d | chain %7.1f us %5.0f GB/s(12B) | %s %7.1f us '
              '%5.0f GB/s(8B) %5.0f GB/s(12B-equiv) | x%.2f | same=%s' % (
                  C, HW, half, count, d[0], d[1], d[2], d[3], d[4], d[5], d[6], times['chain'] * 1e6, n * 12 / times['chain'] / 1e9,
                  'resident' if rc == 0 else 'chain   ', times['res'] * 1e6, n * 8 / times['res'] / 1e9, n * 12 / times['res'] / 1e9,
                  times['chain'] / times['res'], same), flush=True)
        for k in tot:
            tot[k] += times[k] * count
        elems_total += n * count
        del xs, ys, yr
    for k in tot:
        print('per forward %-6s %8.3f ms  %6.1f G elem/s  %5.0f GB/s on the 12 B/elem accounting (%.1f %% of 8 TB/s)' % (
            k, tot[k] * 1e3, elems_total / tot[k] / 1e9, elems_total * 12 / tot[k] / 1e9, elems_total * 12 / tot[k] / 8e12 * 100))


if __name__ == '__main__':
    main()
