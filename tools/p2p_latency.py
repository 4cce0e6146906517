#!/usr/bin/env python3
"""Per-exchange device time of the peer-to-peer all-gather vs the backend collective (development aid).
torchrun --nproc-per-node N tools/p2p_latency.py   (N ranks; CNNQ_BENCH_BACKEND=gloo lets them share one GPU,
which exercises the protocol but says nothing about xGMI latency)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from cnn_quantization_amd import distributed as D  # noqa: E402


def main():
    backend = os.environ.get('CNNQ_BENCH_BACKEND', 'nccl')
    lr = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(lr if backend == 'nccl' else 0)
    dist.init_process_group(backend)
    os.environ['CNNQ_P2P_EXCHANGE'] = '1'
    ex = D.p2p_exchange(None)
    rec = torch.randn(2, 2048, device='cuda')
    for name, fn in (('p2p', (lambda: ex.all_gather(rec)) if ex else None), ('collective', lambda: D.collective_all_gather(rec))):
        if fn is None:
            print('p2p exchange unavailable'); continue
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fn()
        e1.record(); torch.cuda.synchronize()
        if dist.get_rank() == 0:
            print('%-10s %.1f us per all-gather of [2, 2048] fp32 (world %d)' % (name, e0.elapsed_time(e1) * 5, dist.get_world_size()))
    if ex:
        print('rank %d healthy: %s' % (dist.get_rank(), ex.healthy()))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
