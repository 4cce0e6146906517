"""Round 4 probe: does the PLACEMENT of y relative to x matter for the single launch?  x and y live in one allocation, y at
x + tensor bytes + d; us per launch of cnnq_pc_minmax_qdq_group by d.  (Different allocations of the same tensors differ
by +-10 % on this kernel: tools/warm_probe.py.)"""
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
shapes = [(256, 56), (64, 112)] if len(sys.argv) < 2 else [tuple(int(v) for v in s.split('x')) for s in sys.argv[1].split(',')]
for (C, hw) in shapes:
    HW = hw * hw
    n = N * C * HW
    nb = n * 4
    slack = 192 << 20
    for trial in range(3):                                   # three different allocations
        arena = torch.empty((2 * nb + slack) // 4, dtype=torch.float32, device=dev)
        base = arena.data_ptr()
        x0 = (base + 4095) & ~4095
        src = bench.laplace_activation((N, C, hw, hw), 5, dev)
        xv = arena[(x0 - base) // 4:(x0 - base) // 4 + n].view(N, C, hw, hw)
        xv.copy_(src)
        qp = torch.empty((3, C), dtype=torch.float32, device=dev)
        print('[512,%d,%d,%d] allocation %d at 0x%x (x at +0x%x):' % (C, hw, hw, trial, base, x0 - base), flush=True)
        # (a) the tensors' start inside the allocation
        line = ''
        for dx in [0, 64 << 10, 1 << 20, 2 << 20, 4 << 20, 6 << 20, 8 << 20, 12 << 20, 16 << 20, 20 << 20, 24 << 20, 32 << 20, 40 << 20, 48 << 20, 64 << 20, 80 << 20]:
            xs0 = x0 + dx
            if xs0 + 2 * nb > base + arena.numel() * 4:
                continue
            xv2 = arena[(xs0 - base) // 4:(xs0 - base) // 4 + n]
            if dx:
                xv2.copy_(src.view(-1))
            yp = xs0 + nb
            def run2():
                _lib.check(lib.cnnq_pc_minmax_qdq_group(xs0, yp, N, C, HW, 4, 0, ws, qp.data_ptr(), None, 0, st), 'g')
            run2(); run2(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(6): run2()
            e1.record(); torch.cuda.synchronize()
            line += '  x+%dK %.0f' % (dx >> 10, e0.elapsed_time(e1) / 6 * 1e3)
        print(' start of x inside the allocation:' + line, flush=True)
        xv.copy_(src)
        del src
        ds = [0, 4 << 10, 8 << 10, 16 << 10, 32 << 10, 64 << 10, 128 << 10, 256 << 10, 512 << 10, 1 << 20, 2 << 20, 3 << 20, 4 << 20,
              8 << 20, 16 << 20, 32 << 20, 64 << 20, (33 << 20) + (4 << 10)]
        line = ''
        for d in ds:
            yp = x0 + nb + d
            def run():
                _lib.check(lib.cnnq_pc_minmax_qdq_group(x0, yp, N, C, HW, 4, 0, ws, qp.data_ptr(), None, 0, st), 'g')
            run(); run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(6): run()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 6 * 1e-3
            line += '  d=%dK %.0f' % (d >> 10, t * 1e6)
        print(line, flush=True)
        keep = arena            # keep this allocation alive so that the next one lands elsewhere
        arena = None
