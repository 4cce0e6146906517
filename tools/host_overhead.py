import sys, time, torch
sys.path.insert(0, '/root/repo')
from cnn_quantization_amd import ops
x = torch.randn(2, 8, 4, 4, device='cuda'); y = torch.empty_like(x)
for name, fn in (('act_qdq_per_channel cfg2', lambda: ops.act_qdq_per_channel(x, 4, out=y)),
                 ('act_qdq_per_channel cfg3', lambda: ops.act_qdq_per_channel(x, 4, clip='laplace', bit_alloc=True, out=y)),
                 ('torch.empty x4', lambda: [torch.empty((4, 2, 8), device='cuda') for _ in range(4)])):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2000): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('%-28s host %.1f us per call (gpu drained after %.1f us more per call)' % (name, (t1 - t0) / 2000 * 1e6, (t2 - t1) / 2000 * 1e6))
