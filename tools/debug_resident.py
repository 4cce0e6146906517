"""Diagnose the resident kernel: which channels differ from torch / the chain, with and without the exchange."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnn_quantization_amd import ops, _lib as L
lib = L.load()
dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for (C, hw, half, count) in bench.RESNET50_CONV_OUTPUTS:
    N, HW = B, hw * hw
    d = (ctypes.c_int32 * 8)(); lib.cnnq_pc_resident_describe(N, C, HW, d)
    for trial in range(3):
        x = bench.laplace_activation((N, C, hw, hw), 500 + trial, dev)
        mn_t, mx_t = x.amin(dim=(0, 2, 3)), x.amax(dim=(0, 2, 3))
        for flags in (0, 1):
            y, parts = ops.minmax_qdq_resident(x, N, C, HW, 4, half, want_parts=True, flags=flags)
            torch.cuda.synchronize()
            st = parts['stats']
            bad_mn = (st[0] != mn_t).nonzero().flatten().tolist()
            bad_mx = (st[1] != mx_t).nonzero().flatten().tolist()
            yc, codes, pc = ops.minmax_qdq_fused(x, N, C, HW, 4, half, want_codes=True, want_parts=True)
            bad_qp = (parts['qp'] != pc['qp']).any(0).nonzero().flatten().tolist()
            ne = (y != yc)
            bad_ch = ne.any(dim=0).any(dim=-1).any(dim=-1).nonzero().flatten().tolist()
            print('C=%d hw=%d half=%d plan=%s trial=%d flags=%d: bad_min_ch=%s(%d) bad_max_ch=%s(%d) bad_qp_ch=%s(%d) y_mismatch=%d in channels %s(%d) status=%d' % (
                C, hw, half, list(d), trial, flags, bad_mn[:6], len(bad_mn), bad_mx[:6], len(bad_mx), bad_qp[:6], len(bad_qp), int(ne.sum()), bad_ch[:6], len(bad_ch), ops.resident_status(x)), flush=True)
            if bad_ch and trial == 0 and flags == 0:
                c = bad_ch[0]
                idx = ne[:, c].nonzero()[:5].tolist()
                print('   first mismatches in channel %d at (n,h,w): %s ; per-sample mismatch counts: %s' % (c, idx, ne[:, c].sum(dim=(1, 2)).tolist()[:64]))
                print('   min/max resident %r %r torch %r %r ; qp resident %s chain %s' % (float(st[0, c]), float(st[1, c]), float(mn_t[c]), float(mx_t[c]), parts['qp'][:, c].tolist(), pc['qp'][:, c].tolist()))
