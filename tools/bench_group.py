"""Per-layer check + timing of the group-exchange single launch (cnnq_pc_minmax_qdq_group) against the three-launch
chain on the ResNet-50 layer set: rotating distinct buffers (nothing re-read from a cache, a different tensor in
every launch on the SAME never-re-zeroed workspace), all outputs compared bit for bit after several rounds.

    python tools/bench_group.py [--batch 512] [--reps 10] [--rounds 3]
    CNNQ_GRP_K=16 python tools/bench_group.py ...      (kernel sweeps: force the tile height)
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
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--rotate-mb', type=int, default=700)
    ap.add_argument('--shapes', type=str, default='')
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device('cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = ctypes.c_void_p()
    _lib.check(lib.cnnq_group_ws_alloc(32 << 20, ctypes.byref(ws)), 'alloc')      # fine-grained, zeroed ONCE
    want = set(args.shapes.split(',')) if args.shapes else None
    tot = {'chain': 0., 'group': 0.}
    elems_total = 0
    print('batch %d, K=%s target=%s' % (args.batch, os.environ.get('CNNQ_GRP_K', 'auto'), os.environ.get('CNNQ_GRP_WGS', 'dflt')))
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
        rc = lib.cnnq_pc_group_describe(N, C, HW, d)
        assert rc == 0 and lib.cnnq_pc_group_workspace(N, C, HW) <= (32 << 20)

        def chain(i):
            _lib.check(lib.cnnq_pc_minmax_qdq(xs[i].data_ptr(), ys[i].data_ptr(), N, C, HW, 4, int(half), pmm.data_ptr(),
                                              qp.data_ptr(), None, None, st), 'chain')

        def group(i):
            _lib.check(lib.cnnq_pc_minmax_qdq_group(xs[i].data_ptr(), yr[i].data_ptr(), N, C, HW, 4, int(half),
                                                    ws, qp2.data_ptr(), None, 0, st), 'group')
        for i in range(nbuf):
            chain(i)
        bad = 0
        for rnd in range(args.rounds):
            for i in range(nbuf):
                group(i)
            torch.cuda.synchronize()
            bad += sum(int((a != b).sum()) for a, b in zip(ys, yr))
            for t in yr:
                t.zero_()
        times = {}
        for name, fn in (('chain', chain), ('group', group)):
            fn(0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(args.reps):
                fn(r % nbuf)
            e1.record()
            torch.cuda.synchronize()
            times[name] = e0.elapsed_time(e1) * 1e-3 / args.reps
        stw = ctypes.c_uint32()
        lib.cnnq_group_ws_status(ws, ctypes.byref(stw))
        status = int(stw.value)
        print('C=%4d HW=%5d half=%d x%2d | A=%d K=%2d S=%3d Gs=%3d wgs=%5d | chain %7.1f us %5.0f GB/s(12B) | group %7.1f us '
              '%5.0f GB/s(8B) %5.0f GB/s(12B-equiv) | x%.2f | mismatches=%d status=%d' % (
                  C, HW, half, count, d[0], d[1], d[3], d[5], d[7], times['chain'] * 1e6, n * 12 / times['chain'] / 1e9,
                  times['group'] * 1e6, n * 8 / times['group'] / 1e9, n * 12 / times['group'] / 1e9,
                  times['chain'] / times['group'], bad, status), flush=True)
        for k in tot:
            tot[k] += times[k] * count
        elems_total += n * count
        del xs, ys, yr
    for k in tot:
        print('per forward %-6s %8.3f ms  %6.1f G elem/s  %5.0f GB/s on the 12 B/elem accounting (%.1f %% of 8 TB/s)' % (
            k, tot[k] * 1e3, elems_total / tot[k] / 1e9, elems_total * 12 / tot[k] / 1e9, elems_total * 12 / tot[k] / 8e12 * 100))


if __name__ == '__main__':
    main()
