"""Round 4 probe: does the single launch speed up when y shares x's pages (in place: half the distinct pages in flight)?
Timing only - the library forbids in-place use (the cold path re-reads x)."""
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
N = 512
for (C, hw) in ((256, 56), (64, 112), (512, 28), (1024, 14)):
    HW = hw * hw
    xs = [bench.laplace_activation((N, C, hw, hw), 5 + i, dev) for i in range(2)]
    ys = [torch.empty_like(xs[0]) for _ in range(2)]
    qp = torch.empty((3, C), dtype=torch.float32, device=dev)
    def run(x, y):
        _lib.check(lib.cnnq_pc_minmax_qdq_group(x.data_ptr(), y.data_ptr(), N, C, HW, 4, 0, ws, qp.data_ptr(), None, 0, st), 'g')
    res = {}
    for name, pairs in (('distinct y', list(zip(xs, ys))), ('y = x (in place)', list(zip(xs, xs))), ('distinct y again', list(zip(xs, ys)))):
        for x, y in pairs: run(x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(8):
            x, y = pairs[r % 2]; run(x, y)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 8 * 1e3
        print('[%d,%d,%d,%d] %-18s %7.1f us  %.2f TB/s (8B)' % (N, C, hw, hw, name, t, N * C * HW * 8 / t / 1e6), flush=True)
    del xs, ys
