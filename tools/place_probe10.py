"""Round 4 probe: a map of the fast / slow placements - K consecutive buffers of 1.64 GB built from 2 MB physical chunks
(scat_alloc, creation order: the driver hands the chunks out as it pleases, presumably walking through free memory), the
single launch timed on consecutive pairs (buffer 2i -> 2i+1) and on every buffer against buffer 0 and against itself shifted."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnn_quantization_amd import _lib
lib = _lib.load()
sa = ctypes.CDLL(__import__('_ubuild').so('scatter_alloc'))
sa.scat_alloc.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint, ctypes.POINTER(ctypes.c_void_p)]
hip = ctypes.CDLL('libamdhip64.so')
dev = torch.device('cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = ctypes.c_void_p()
_lib.check(lib.cnnq_group_ws_alloc(32 << 20, ctypes.byref(ws)), 'alloc')
N, C, hw = 512, 256, 56
HW = hw * hw
nb = N * C * HW * 4
K = int(os.environ.get('NBUF', '40'))
CH = int(os.environ.get('CHUNK_MB', '2')) << 20
src = bench.laplace_activation((N, C, hw, hw), 5, dev)
qp = torch.empty((3, C), dtype=torch.float32, device=dev)
bufs = []
for i in range(K):
    p = ctypes.c_void_p()
    rc = sa.scat_alloc(nb, CH, 0, 0, ctypes.byref(p))
    assert rc == 0, (i, rc)
    bufs.append(p.value)
def measure(xp, yp):
    hip.hipMemcpyAsync(ctypes.c_void_p(xp), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(nb), 3, st)
    run = lambda: _lib.check(lib.cnnq_pc_minmax_qdq_group(xp, yp, N, C, HW, 4, 0, ws, qp.data_ptr(), None, 0, st), 'g')
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 4 * 1e3
print('%d buffers of %.2f GB from %d MB chunks; us per launch' % (K, nb / 2**30, CH >> 20))
print('consecutive pairs (x = buffer 2i, y = buffer 2i+1):')
print(' '.join('%4.0f' % measure(bufs[2 * i], bufs[2 * i + 1]) for i in range(K // 2)), flush=True)
print('every buffer as y, x = buffer 0:')
print(' '.join('%4.0f' % measure(bufs[0], bufs[j]) for j in range(1, K)), flush=True)
print('every buffer as y, x = buffer %d:' % (K - 1))
print(' '.join('%4.0f' % measure(bufs[K - 1], bufs[j]) for j in range(0, K - 1)), flush=True)
