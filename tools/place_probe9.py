"""Round 4 probe: ONE set of physical 2 MB chunks for x and one for y, mapped under several virtual orders
(tools/scatter_alloc.hip scat_views): does the ORDER of the same physical memory decide between the fast and the slow placement?"""
import ctypes, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnn_quantization_amd import _lib
lib = _lib.load()
sa = ctypes.CDLL(__import__('_ubuild').so('scatter_alloc'))
hip = ctypes.CDLL('libamdhip64.so')
dev = torch.device('cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = ctypes.c_void_p()
_lib.check(lib.cnnq_group_ws_alloc(32 << 20, ctypes.byref(ws)), 'alloc')
N, C, hw = 512, 256, 56
HW = hw * hw
nb = N * C * HW * 4
CH = int(os.environ.get('CHUNK_MB', '2')) << 20
nch = (nb + CH - 1) // CH
src = bench.laplace_activation((N, C, hw, hw), 5, dev)
qp = torch.empty((3, C), dtype=torch.float32, device=dev)
names, perms = [], []
def add(name, p):
    names.append(name); perms.append(list(p))
add('creation order', range(nch))
add('reversed', reversed(range(nch)))
for seed in (1, 2, 3):
    p = list(range(nch)); random.Random(seed).shuffle(p); add('shuffled %d' % seed, p)
add('even chunks first', list(range(0, nch, 2)) + list(range(1, nch, 2)))
add('blocks of 8 reversed', [b * 8 + (7 - i) for b in range((nch + 7) // 8) for i in range(8) if b * 8 + (7 - i) < nch])
V = len(perms)
arr = (ctypes.c_uint32 * (V * nch))(*[v for p in perms for v in p])
def views():
    out = (ctypes.c_void_p * V)()
    rc = sa.scat_views(ctypes.c_size_t(nb), ctypes.c_size_t(CH), V, arr, out)
    assert rc == 0, rc
    return [out[i] for i in range(V)]
xv, yv = views(), views()
def measure(xp, yp):
    hip.hipMemcpyAsync(ctypes.c_void_p(xp), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(nb), 3, st)
    run = lambda: _lib.check(lib.cnnq_pc_minmax_qdq_group(xp, yp, N, C, HW, 4, 0, ws, qp.data_ptr(), None, 0, st), 'g')
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 * 1e3
print('chunks of %d MB, %d per buffer; us per launch, rows: order of x, columns: order of y' % (CH >> 20, nch))
print('%-22s' % '' + ''.join('%10s' % n[:9] for n in names))
for i in range(V):
    print('%-22s' % names[i] + ''.join('%10.0f' % measure(xv[i], yv[j]) for j in range(V)), flush=True)
