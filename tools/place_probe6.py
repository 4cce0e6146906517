"""Round 4 probe: x at the start of one big allocation, y at x + D for D in steps of 256 MB."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device('cuda')
GB = 20
arena = torch.empty(GB << 28, dtype=torch.float32, device=dev)
import bench
from cnn_quantization_amd import _lib
lib = _lib.load()
hip = ctypes.CDLL('libamdhip64.so')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = ctypes.c_void_p()
_lib.check(lib.cnnq_group_ws_alloc(32 << 20, ctypes.byref(ws)), 'alloc')
N, C, hw = 512, 256, 56
HW = hw * hw
n = N * C * HW
nb = n * 4
src = bench.laplace_activation((N, C, hw, hw), 5, dev)
qp = torch.empty((3, C), dtype=torch.float32, device=dev)
base = (arena.data_ptr() + (2 << 20) - 1) & ~((2 << 20) - 1)
def measure(xp, yp):
    hip.hipMemcpyAsync(ctypes.c_void_p(xp), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(nb), 3, st)
    def run():
        _lib.check(lib.cnnq_pc_minmax_qdq_group(xp, yp, N, C, HW, 4, 0, ws, qp.data_ptr(), None, 0, st), 'g')
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(5): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 * 1e3
for xoff in (0, 1 << 30, 3 << 30):
    line = 'x at +%d MB; y at x + D, D (MB): us  ' % (xoff >> 20)
    D = ((nb + (2 << 20) - 1) >> 21) << 21
    while xoff + D + nb <= (GB << 30) - (4 << 20):
        line += ' %d:%.0f' % (D >> 20, measure(base + xoff, base + xoff + D))
        D += 256 << 20
    print(line, flush=True)
