"""Round 4 probe: the single launch on buffers whose physical chunks are mapped in creation order / shuffled (tools/scatter_alloc.hip)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnn_quantization_amd import _lib
lib = _lib.load()
sa = ctypes.CDLL(__import__('_ubuild').so('scatter_alloc'))
sa.scat_alloc.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint, ctypes.POINTER(ctypes.c_void_p)]
sa.scat_granularity.restype = ctypes.c_size_t
hip = ctypes.CDLL('libamdhip64.so')
dev = torch.device('cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = ctypes.c_void_p()
_lib.check(lib.cnnq_group_ws_alloc(32 << 20, ctypes.byref(ws)), 'alloc')
print('allocation granularity: minimum %d, recommended %d bytes' % (sa.scat_granularity(0), sa.scat_granularity(1)))
N, C, hw = 512, 256, 56
HW = hw * hw
n = N * C * HW
nb = n * 4
src = bench.laplace_activation((N, C, hw, hw), 5, dev)
ref = torch.empty_like(src)
qp = torch.empty((3, C), dtype=torch.float32, device=dev)
_lib.check(lib.cnnq_pc_minmax_qdq_group(src.data_ptr(), ref.data_ptr(), N, C, HW, 4, 0, ws, qp.data_ptr(), None, 0, st), 'g')
def measure(xp, yp):
    hip.hipMemcpyAsync(ctypes.c_void_p(xp), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(nb), 3, st)
    def run():
        _lib.check(lib.cnnq_pc_minmax_qdq_group(xp, yp, N, C, HW, 4, 0, ws, qp.data_ptr(), None, 0, st), 'g')
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(6): run()
    e1.record(); torch.cuda.synchronize()
    chk = torch.empty_like(ref)
    hip.hipMemcpyAsync(ctypes.c_void_p(chk.data_ptr()), ctypes.c_void_p(yp), ctypes.c_size_t(nb), 3, st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 6 * 1e3, bool(torch.equal(chk, ref))
for chunk in (2 << 20, 16 << 20, 64 << 10, 256 << 20):
    for order, nm in ((0, 'creation order'), (1, 'shuffled'), (1, 'shuffled (other seed)'), (2, 'reversed')):
        px, py = ctypes.c_void_p(), ctypes.c_void_p()
        rc = sa.scat_alloc(nb, chunk, order, 7 + hash(nm) % 100, ctypes.byref(px)) or sa.scat_alloc(nb, chunk, order, 11 + hash(nm) % 100, ctypes.byref(py))
        if rc:
            print('chunk %d KB %s: allocation failed (%d)' % (chunk >> 10, nm, rc)); continue
        t, ok = measure(px.value, py.value)
        print('chunks of %6d KB, %-22s: %.0f us  %.2f TB/s  bits %s' % (chunk >> 10, nm, t, n * 8 / t / 1e6, 'ok' if ok else 'DIFFER'), flush=True)
