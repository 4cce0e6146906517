"""Round 4 probe: x and y carved out of ONE big allocation at swept offsets (is a favourable placement a matter of the
buffers' position modulo some power of two of the physical address, reachable through offsets inside a physically contiguous
block?)."""
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
N, C, hw = 512, 256, 56
HW = hw * hw
nb = N * C * HW * 4
src = bench.laplace_activation((N, C, hw, hw), 5, dev)
qp = torch.empty((3, C), dtype=torch.float32, device=dev)
pool = torch.empty(12 << 30, dtype=torch.uint8, device=dev)
base = (pool.data_ptr() + (1 << 21) - 1) & ~((1 << 21) - 1)
def measure(xp, yp):
    ctypes.CDLL('libamdhip64.so').hipMemcpyAsync(ctypes.c_void_p(xp), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(nb), 3, st)
    run = lambda: _lib.check(lib.cnnq_pc_minmax_qdq_group(xp, yp, N, C, HW, 4, 0, ws, qp.data_ptr(), None, 0, st), 'g')
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 * 1e3
print('pool at %#x' % base)
for a_mb in range(0, 4096, 256):
    line = 'x at +%4d MB:' % a_mb
    for gap_mb in (2048, 2048 + 512, 4096 + 128):
        xp = base + (a_mb << 20)
        yp = xp + (gap_mb << 20)
        if yp + nb > base + (12 << 30) - (4 << 20):
            continue
        line += '  y at x + %4d MB: %5.0f us' % (gap_mb, measure(xp, yp))
    print(line, flush=True)
