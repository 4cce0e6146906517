#!/usr/bin/env python3
"""Device-side cost of one torch.distributed all_gather_into_tensor of a [2, C] fp32 record on a 1-rank RCCL group
(what the per-tensor statistics exchange costs on this box, without any peer): events around 200 back-to-back
collectives, and around 200 collectives each sandwiched between two small kernels on the compute stream (the real
launch pattern: the stream hop to RCCL's stream and back is part of the price)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29655')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
for C in (64, 512, 2048):
    rec = torch.randn(2, C, device='cuda')
    out = torch.empty(1, 2, C, device='cuda')
    a = torch.zeros(1024, device='cuda')

    def bare():
        dist.all_gather_into_tensor(out.view(-1), rec.view(-1))

    def sandwiched():
        a.add_(1.0)
        dist.all_gather_into_tensor(out.view(-1), rec.view(-1))
        a.add_(1.0)

    def kernels_only():
        a.add_(1.0)
        a.add_(1.0)
    res = {}
    for name, fn in (('bare', bare), ('sandwiched', sandwiched), ('kernels_only', kernels_only)):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 5
    print('C=%4d: all_gather alone %.1f us; between two small kernels %.1f us (the two kernels alone %.1f us) -> %.1f us per exchange'
          % (C, res['bare'], res['sandwiched'], res['kernels_only'], res['sandwiched'] - res['kernels_only']))
dist.destroy_process_group()
