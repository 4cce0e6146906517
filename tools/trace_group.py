"""Per-workgroup phase timeline of k_mmq_group (development build with -DGRP_TRACE, see tools/runs/r3_diag.sh):
lane 0 of every workgroup stamps start / loads issued / tile reduced / published / group met / parameters ready /
stores issued with the 100 MHz clock.  Prints per-phase percentiles and, per 2 us bucket of the launch, how many
workgroups sit in the load phase, in the exchange and in the Q/DQ + store phase.

    CNNQ_HIP_LIB=tools/libcnnq_trace.so python tools/trace_group.py [--shapes 256x56,64x112]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from cnn_quantization_amd import _lib


def pct(a, q):
    return float(np.percentile(a, q)) if len(a) else float('nan')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--shapes', type=str, default='256x56,64x112,512x28,1024x14,256x14')
    ap.add_argument('--save', type=str, default='', help='directory for the raw stamps (npz per shape)')
    ap.add_argument('--packed', action='store_true', help='trace the packed-nibble form (OUT = 2) of the single launch')
    args = ap.parse_args()
    lib = _lib.load()
    lib.cnnq_debug_group_trace.restype = ctypes.c_int
    lib.cnnq_debug_group_trace.argtypes = [ctypes.c_void_p]
    dev = torch.device('cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = ctypes.c_void_p()
    _lib.check(lib.cnnq_group_ws_alloc(32 << 20, ctypes.byref(ws)), 'alloc')
    want = set(args.shapes.split(','))
    for (C, hw, half, count) in bench.RESNET50_CONV_OUTPUTS:
        if ('%dx%d' % (C, hw)) not in want:
            continue
        N, HW = args.batch, hw * hw
        xs = [bench.laplace_activation((N, C, hw, hw), 100 + i, dev) for i in range(2)]
        ys = [torch.empty_like(xs[0]) for _ in range(2)]
        qp = torch.empty((3, C), dtype=torch.float32, device=dev)
        d = (ctypes.c_int32 * 8)()
        assert lib.cnnq_pc_group_describe(N, C, HW, d) == 0
        A, K, mode, S, ncb, Gs, ngroups, wgs = list(d)
        tr = torch.zeros((wgs, 16), dtype=torch.int64, device=dev)

        pk = [torch.empty(xs[0].numel() // 2, dtype=torch.uint8, device=dev) for _ in range(2)] if args.packed else None

        def run(i):
            if args.packed:
                _lib.check(lib.cnnq_pc_minmax_qdq_single(xs[i].data_ptr(), None, N, C, HW, 4, int(half), ws, 32 << 20,
                                                         qp.data_ptr(), None, None, None, pk[i].data_ptr(), st), 'single')
                return
            _lib.check(lib.cnnq_pc_minmax_qdq_group(xs[i].data_ptr(), ys[i].data_ptr(), N, C, HW, 4, int(half), ws,
                                                    qp.data_ptr(), None, 0, st), 'group')
        _lib.check(lib.cnnq_debug_group_trace(None), 'trace off')
        run(0); run(1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(0); run(1); e1.record(); torch.cuda.synchronize()
        t_plain = e0.elapsed_time(e1) / 2 * 1e3
        _lib.check(lib.cnnq_debug_group_trace(tr.data_ptr()), 'trace on')
        run(0)
        torch.cuda.synchronize()
        tr.zero_()
        e0.record(); run(1); e1.record(); torch.cuda.synchronize()
        t_traced = e0.elapsed_time(e1) * 1e3
        _lib.check(lib.cnnq_debug_group_trace(None), 'trace off')
        t = tr.cpu().numpy()
        if args.save:
            os.makedirs(args.save, exist_ok=True)
            np.savez_compressed(os.path.join(args.save, 'trace_%dx%d.npz' % (C, hw)), t=t, plain_us=t_plain, traced_us=t_traced, Gs=Gs, K=K)
        hw_id = t[:, 7]
        T = t[:, :7].astype(np.float64) / 100.0     # us
        T -= T[:, 0].min()
        span = T[:, 6].max()
        print('\n[%d,%d,%d,%d] A=%d K=%d mode=%d S=%d Gs=%d groups=%d wgs=%d | launch %.1f us plain, %.1f us traced, '
              'stamped span %.1f us | %.0f GB/s (8 B/elem)' % (N, C, hw, hw, A, K, mode, S, Gs, ngroups, wgs, t_plain,
                                                             t_traced, span, N * C * HW * 8 / t_plain / 1e3))
        names = ['start->loads issued', 'issued->tile reduced', 'reduced->published', 'published->met',
                 'met->params', 'params->stores issued', 'stores issued->drained', 'whole workgroup']
        Tend = t[:, 8].astype(np.float64) / 100.0 - (t[:, 0].astype(np.float64) / 100.0).min()
        segs = [T[:, 1] - T[:, 0], T[:, 2] - T[:, 1], T[:, 3] - T[:, 2], T[:, 4] - T[:, 3], T[:, 5] - T[:, 4],
                T[:, 6] - T[:, 5], Tend - T[:, 6], Tend - T[:, 0]]
        for nm, sgm in zip(names, segs):
            print('   %-24s p10 %6.2f  p50 %6.2f  p90 %6.2f  max %7.2f us' % (nm, pct(sgm, 10), pct(sgm, 50), pct(sgm, 90), sgm.max()))
        # residency over time
        nb = int(span / 2) + 1
        occ = np.zeros((3, nb))
        for ph, (a, b) in enumerate(((0, 2), (2, 5), (5, 6))):
            for w in range(0, wgs):
                i0, i1 = int(T[w, a] / 2), int(T[w, b] / 2)
                occ[ph, i0:i1 + 1] += 1
        step = max(1, nb // 40)
        print('   t(us): workgroups loading / exchanging / computing+storing   (every %d us)' % (2 * step))
        print('   ' + ' '.join('%d:%d/%d/%d' % (i * 2, occ[0, i], occ[1, i], occ[2, i]) for i in range(0, nb, step)))
        xcc = (hw_id >> 32) & 0xf
        print('   workgroups per XCC: %s' % np.bincount(xcc.astype(np.int64), minlength=8).tolist())
        starts = np.sort(T[:, 0])
        print('   start times (us) at every 10%% of the workgroups: %s' % ' '.join('%.0f' % starts[int(q * (wgs - 1))] for q in np.linspace(0, 1, 11)))
        g = np.arange(wgs) // Gs
        skews = [T[g == k, 3].max() - T[g == k, 3].min() for k in range(0, ngroups, max(1, ngroups // 64))]
        print('   arrival skew inside a group (last - first publish): p50 %.2f  p90 %.2f  max %.2f us' % (pct(skews, 50), pct(skews, 90), max(skews)))
        del xs, ys, tr


if __name__ == '__main__':
    main()
