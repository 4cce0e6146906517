#!/usr/bin/env python3
"""Host time of one hot call of the multi-GPU path (forced exchange on a 1-rank nccl group), cProfile over 5000 calls."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['CNNQ_FORCE_EXCHANGE'] = '1'
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29791')
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from cnn_quantization_amd import ops  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
x = torch.randn(8, 64, 56, 56, device='cuda')
y = torch.empty_like(x)
fn = lambda: ops.act_qdq_per_channel(x, 4, out=y)
for _ in range(200):
    fn()
torch.cuda.synchronize()
n = 5000
t0 = time.perf_counter()
for _ in range(n):
    fn()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host %.2f us per call (gpu drained after %.2f us more per call)' % ((t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    fn()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
dist.destroy_process_group()
