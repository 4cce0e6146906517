#!/usr/bin/env python3
"""Config 2 on layers whose channel population exceeds the plain register tiles (VGG-16 b512: [512,64,224,224], 103 MB per
channel): the three-launch chain against the single launch with eight more tile rows in LDS (K = 32 + 8: 628 members per
channel, one channel resident at a time).  Development build with knobs: CNNQ_FLAT_KL=8."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnn_quantization_amd import _lib
lib = _lib.load()
dev = torch.device('cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = ctypes.c_void_p()
_lib.check(lib.cnnq_group_ws_alloc(32 << 20, ctypes.byref(ws)), 'alloc')
for shape in [(512, 64, 224, 224), (512, 128, 112, 112), (512, 256, 56, 56)]:
    N, C, H, W = shape
    HW = H * W
    x = bench.laplace_activation(shape, 7, dev)
    y, yr = torch.empty_like(x), torch.empty_like(x)
    G = lib.cnnq_pc_groups(N, C, HW, 1)
    pmm = torch.empty((G, 2, C), device=dev)
    qp, qp2 = torch.empty((3, C), device=dev), torch.empty((3, C), device=dev)
    d = (ctypes.c_int32 * 8)()
    rc = lib.cnnq_pc_group_describe(N, C, HW, d)
    def chain():
        _lib.check(lib.cnnq_pc_minmax_qdq(x.data_ptr(), y.data_ptr(), N, C, HW, 4, 0, pmm.data_ptr(), qp.data_ptr(), None, None, st), 'chain')
    def single():
        return lib.cnnq_pc_minmax_qdq_group(x.data_ptr(), yr.data_ptr(), N, C, HW, 4, 0, ws, qp2.data_ptr(), None, 0, st)
    chain()
    rc2 = single()
    torch.cuda.synchronize()
    ok = rc2 == 0 and bool(torch.equal(y, yr))
    res = {}
    for name, fn in (('chain', chain), ('single', single)):
        if name == 'single' and rc2 != 0:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 5
    s = ctypes.c_uint32()
    lib.cnnq_group_ws_status(ws, ctypes.byref(s))
    print(shape, 'describe rc', rc, list(d), 'single rc', rc2, 'equal', ok, {k: '%.3f ms' % v for k, v in res.items()},
          {k: '%.2f TB/s' % (x.numel() * (12 if k == 'chain' else 8) / v / 1e9) for k, v in res.items()}, 'status', s.value, flush=True)
    del x, y, yr
