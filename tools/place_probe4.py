"""Round 4 probe: (x, y) pairs carved out of ONE big allocation made first thing in the process, then out of a second one."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device('cuda')
GB = int(os.environ.get('ARENA_GB', '24'))
arenas = [torch.empty(GB << 28, dtype=torch.float32, device=dev)]          # before anything else touches the device heap
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
arenas.append(torch.empty(GB << 28, dtype=torch.float32, device=dev))
def measure(xp, yp):
    hip.hipMemcpyAsync(ctypes.c_void_p(xp), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(nb), 3, st)
    def run():
        _lib.check(lib.cnnq_pc_minmax_qdq_group(xp, yp, N, C, HW, 4, 0, ws, qp.data_ptr(), None, 0, st), 'g')
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(6): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 6 * 1e3
for k, a in enumerate(arenas):
    base = (a.data_ptr() + (2 << 20) - 1) & ~((2 << 20) - 1)
    pairs = (a.numel() * 4 - (4 << 20)) // (2 * nb)
    line = 'arena %d (%d GB at 0x%x, %s):' % (k, GB, a.data_ptr(), 'first allocation of the process' if k == 0 else 'allocated after the inputs were generated')
    for i in range(int(pairs)):
        t = measure(base + 2 * i * nb, base + (2 * i + 1) * nb)
        line += '  %.0f' % t
    # x from the front, y from the back
    t = measure(base, base + (2 * int(pairs) - 1) * nb)
    print(line + '   | x first / y last region: %.0f' % t, flush=True)
