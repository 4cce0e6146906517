"""Round 4 probe: is a fast placement a property of x's region, of y's region or of the pair?  Eight separate 1.64 GB regions;
read-only ablation on each as x, write-only ablation on each as y, address-order streaming write / read on each
(tools/ubench_rw.so), then the product kernel on pairs."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnn_quantization_amd import _lib
here = os.path.dirname(os.path.abspath(__file__))
lib = _lib.load()
def other(path):
    l = ctypes.CDLL(path)
    l.cnnq_pc_minmax_qdq_group.restype = ctypes.c_int
    l.cnnq_pc_minmax_qdq_group.argtypes = _lib.SIGNATURES['cnnq_pc_minmax_qdq_group'][1]
    return l
ro, wo = other(os.path.join(here, 'alt', 'libcnnq_abl3.so')), other(os.path.join(here, 'alt', 'libcnnq_abl6.so'))
rw = ctypes.CDLL(__import__('_ubuild').so('ubench_rw'))
rw.urw.restype = ctypes.c_float
rw.urw.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3
dev = torch.device('cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = ctypes.c_void_p()
_lib.check(lib.cnnq_group_ws_alloc(32 << 20, ctypes.byref(ws)), 'alloc')
N, C, hw = 512, 256, 56
HW = hw * hw
n = N * C * HW
src = bench.laplace_activation((N, C, hw, hw), 5, dev)
qp = torch.empty((3, C), dtype=torch.float32, device=dev)
out = torch.zeros(16, device=dev)
R = 8
regs = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(R)]
for r in regs: r.copy_(src.view(-1))
def t_kernel(l, xp, yp, reps=5):
    def run():
        l.cnnq_pc_minmax_qdq_group(xp, yp, N, C, HW, 4, 0, ws, qp.data_ptr(), None, 0, st)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print('region address        read-only (as x)  write-only (as y)  stream read  stream write  strided write (128 KB tiles)')
P4 = C * HW // 4
for i, r in enumerate(regs):
    p = r.data_ptr()
    a = t_kernel(ro, p, regs[(i + 1) % R].data_ptr())
    b = t_kernel(wo, regs[(i + 1) % R].data_ptr(), p)
    sr = rw.urw(1, 1, 0, 1, 1, p, p, out.data_ptr(), N, P4, 4) * 1e3
    sw = rw.urw(1, 1, 0, 2, 1, p, p, out.data_ptr(), N, P4, 4) * 1e3
    tw = rw.urw(1, 32, 1, 2, 1, p, p, out.data_ptr(), N, P4, 4) * 1e3
    print('%d 0x%x   %7.1f us        %7.1f us        %7.1f us   %7.1f us   %7.1f us' % (i, p, a, b, sr, sw, tw), flush=True)
for r in regs: r.copy_(src.view(-1))
print('product kernel, us per launch, rows = x region, columns = y region')
for i in range(R):
    line = 'x%d ' % i
    for j in range(R):
        if i == j:
            line += '     - '
            continue
        regs[i].copy_(src.view(-1))
        line += ' %6.1f' % t_kernel(lib, regs[i].data_ptr(), regs[j].data_ptr(), 3)
    print(line, flush=True)
