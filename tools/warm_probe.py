"""Round 4 probe: does the single launch (cnnq_pc_minmax_qdq_group) run faster on a small rotated set of buffers (warm) than on a
large one (cold, as the 53 tensors of the bench are)?  The load pass of the packed storage shows a cliff at ~2 GB."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnn_quantization_amd import _lib
lib = _lib.load()
tl = ctypes.CDLL(__import__('_ubuild').so('ubench_touch'))
tl.utouch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
TOUCH = int(os.environ.get('TOUCH_KB', '0')) << 10          # 0: off; else one load per this many bytes of x and y before every launch
dev = torch.device('cuda')
sink = torch.zeros(4, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = ctypes.c_void_p()
_lib.check(lib.cnnq_group_ws_alloc(32 << 20, ctypes.byref(ws)), 'alloc')
N = 512
for (C, hw) in ((64, 56), (128, 56), (256, 56), (1024, 14), (256, 14)):
    HW = hw * hw
    n = N * C * HW
    qp = torch.empty((3, C), dtype=torch.float32, device=dev)
    line = '[512,%d,%d,%d] %4.0f MB:' % (C, hw, hw, n * 4 / 1e6)
    for R in (1, 2, 4, 8):
        if 2 * R * n * 4 > 30e9:
            continue
        xs = [bench.laplace_activation((N, C, hw, hw), 5 + i, dev) for i in range(R)]
        ys = [torch.empty_like(xs[0]) for _ in range(R)]
        def run(i):
            if TOUCH:
                tl.utouch(xs[i].data_ptr(), n * 4, TOUCH, sink.data_ptr(), st)
                tl.utouch(ys[i].data_ptr(), n * 4, TOUCH, sink.data_ptr(), st)
            _lib.check(lib.cnnq_pc_minmax_qdq_group(xs[i].data_ptr(), ys[i].data_ptr(), N, C, HW, 4, 0, ws, qp.data_ptr(), None, 0, st), 'g')
        for i in range(R): run(i)
        torch.cuda.synchronize()
        reps = max(8, 2 * R)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(reps): run(r % R)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e-3
        line += '  %d pair(s) (%.1f GB) %.1f us %.2f TB/s' % (R, 2 * R * n * 4 / 1e9, t * 1e6, n * 8 / t / 1e12)
        del xs, ys
    print(line, flush=True)
