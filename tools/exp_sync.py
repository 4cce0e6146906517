"""Experiment: which workspace memory / poll form makes the in-kernel group exchange correct and fast."""
import os, sys, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnn_quantization_amd import _lib as L
lib = L.load()
dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = sys.argv[2] if len(sys.argv) > 2 else 'cached'        # cached | uncached | memset
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
if mode == 'uncached':
    p = ctypes.c_void_p()
    L.check(lib.cnnq_ws_alloc_uncached(8 << 20, ctypes.byref(p)), 'alloc')
    ws_ptr = p.value
else:
    ws = torch.zeros(8 << 20, dtype=torch.uint8, device=dev)
    ws_ptr = ws.data_ptr()
print('mode', mode, 'lib', os.environ.get('CNNQ_HIP_LIB', 'default'))
for (C, hw, half, count) in bench.RESNET50_CONV_OUTPUTS:
    N, HW = B, hw * hw
    n = N * C * HW
    nbuf = max(2, min(8, (400 << 20) // (4 * n) + 1))
    xs = [bench.laplace_activation((N, C, hw, hw), 300 + i, dev) for i in range(nbuf)]
    yr = [torch.empty_like(xs[0]) for _ in range(nbuf)]
    ys = [torch.empty_like(xs[0]) for _ in range(nbuf)]
    G = lib.cnnq_pc_groups(N, C, HW, 1)
    pmm = torch.empty((G, 2, C), dtype=torch.float32, device=dev)
    qp = torch.empty((3, C), dtype=torch.float32, device=dev)
    qp2 = torch.empty((3, C), dtype=torch.float32, device=dev)
    for i in range(nbuf):
        L.check(lib.cnnq_pc_minmax_qdq(xs[i].data_ptr(), ys[i].data_ptr(), N, C, HW, 4, int(half), pmm.data_ptr(), qp.data_ptr(), None, None, st), 'chain')
    def res(i):
        if mode == 'memset':
            ws[:65536].zero_()
        L.check(lib.cnnq_pc_minmax_qdq_resident(xs[i].data_ptr(), yr[i].data_ptr(), N, C, HW, 4, int(half), ws_ptr, qp2.data_ptr(), None, 0, st), 'res')
    bad = 0
    for rep in range(3):
        for i in range(nbuf):
            res(i)
        torch.cuda.synchronize()
        bad += sum(int((a != b).sum()) for a, b in zip(ys, yr))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(20):
        res(r % nbuf)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 20
    print('C=%4d HW=%5d: resident %7.1f us %5.0f GB/s(8B)  mismatching elements over 3 rounds: %d' % (C, HW, t * 1e6, n * 8 / t / 1e9, bad), flush=True)
    del xs, yr, ys
