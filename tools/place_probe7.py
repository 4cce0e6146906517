"""Round 4 probe: per-layer time of the single launch on the bench workload's own buffers next to the buffers' virtual addresses
(is the fast / slow placement population visible in address bits?)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnn_quantization_amd import ops
PAD_GB = float(os.environ.get('PAD_GB', '0'))      # allocated first and kept: shifts every later buffer's address
pad = torch.empty(int(PAD_GB * (1 << 30)), dtype=torch.uint8, device='cuda') if PAD_GB > 0 else None
layers = bench.build_workload(512, torch.device('cuda'))
big = [L for L in layers if L['x'].numel() >= 400e6]
def t_of(L, reps=6):
    f = lambda: ops.act_qdq_per_channel(L['x'], 4, positive=L['half'], out=L['y'])
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for L in big:
    x, y = L['x'], L['y']
    n = x.numel()
    us = t_of(L)
    xa, ya = x.data_ptr(), y.data_ptr()
    print('%-18s %7.1f us %5.2f TB/s  x %#014x y %#014x  x mod 16 GB %5.2f  y mod 16 GB %5.2f' % (
        'x'.join(map(str, x.shape)), us, n * 8 / us / 1e6, xa, ya, (xa % (16 << 30)) / 2**30, (ya % (16 << 30)) / 2**30), flush=True)
