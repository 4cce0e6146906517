"""Round 4 probe: the single launch on several allocations of the same tensors (kept alive together), from torch's allocator,
from hipMalloc and from hipExtMallocWithFlags(hipDeviceMallocContiguous): which regions are fast?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnn_quantization_amd import _lib
lib = _lib.load()
al = ctypes.CDLL(__import__('_ubuild').so('ubench_rw'))
al.urw_alloc.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
hip = ctypes.CDLL('libamdhip64.so')
dev = torch.device('cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = ctypes.c_void_p()
_lib.check(lib.cnnq_group_ws_alloc(32 << 20, ctypes.byref(ws)), 'alloc')
N, C, hw = int(os.environ.get('NB', '512')), 256, 56
HW = hw * hw
n = N * C * HW
nb = n * 4
src = bench.laplace_activation((N, C, hw, hw), 5, dev)
qp = torch.empty((3, C), dtype=torch.float32, device=dev)
keep = []
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
print('batch %d: %d MB per tensor, %d pages of 2 MB for x and y together' % (N, nb >> 20, 2 * nb >> 21))
for kind in ('torch.empty (x and y separate tensors)', 'hipMalloc'):
    line = '%-42s' % kind
    for i in range(8):
        if kind.startswith('torch'):
            x = torch.empty(n, dtype=torch.float32, device=dev); y = torch.empty(n, dtype=torch.float32, device=dev)
            keep.append((x, y)); xp, yp = x.data_ptr(), y.data_ptr()
        else:
            flags = -1 if kind == 'hipMalloc' else 4
            px, py = ctypes.c_void_p(), ctypes.c_void_p()
            if al.urw_alloc(nb, flags, ctypes.byref(px)) or al.urw_alloc(nb, flags, ctypes.byref(py)):
                line += '  alloc failed'; break
            xp, yp = px.value, py.value
        t = measure(xp, yp)
        line += '  %.0f us (%.2f TB/s)' % (t, n * 8 / t / 1e6)
    print(line, flush=True)
