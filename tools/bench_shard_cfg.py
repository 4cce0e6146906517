#!/usr/bin/env python3
"""BASELINE configs 3 / 4 / 5 at the shard of a batch-sharded run on ONE GPU (round 6): a 1-rank process group forced through the
exchange - the in-launch route (--xrank 1: cnnq_pc_aciq_fused_xrank / _stats_xrank / _midtread_fused_xrank) or the collective
(--xrank 0: the chain around RCCL all_gathers) - or no exchange at all (--plain).  Wall clock per forward (best of 5 around a
synchronised region) and the host time per call (the same loop timed without the final synchronisation), for rocprofv3.

    python tools/bench_shard_cfg.py --config 3 [--batch 64] [--xrank 1 | --xrank 0 | --plain] [--reps 5]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', type=int, default=3, choices=[3, 4, 5])
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--xrank', default='1', choices=['0', '1'])
    ap.add_argument('--plain', action='store_true')
    ap.add_argument('--reps', type=int, default=5)
    a = ap.parse_args()
    os.environ['CNNQ_XRANK'] = a.xrank
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    if not a.plain:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29571')
        os.environ['CNNQ_FORCE_EXCHANGE'] = '1'
        dist.init_process_group('nccl', device_id=dev, rank=0, world_size=1)
    from cnn_quantization_amd import ops, distributed as D
    ops.reload_switches()
    layers, seed = [], 100
    shapes = [(C, hw, True, n) for (C, hw, n) in bench.VGG16_CONV_OUTPUTS] if a.config == 5 else bench.RESNET50_CONV_OUTPUTS
    for (C, hw, half, count) in shapes:
        for _ in range(count):
            layers.append((bench.laplace_activation((a.batch, C, hw, hw), seed, dev), half))
            seed += 1
    elems = sum(x.numel() for x, _ in layers)
    ys = [torch.empty_like(x) for x, _ in layers]

    def fwd():
        for (x, half), y in zip(layers, ys):
            if a.config == 3:
                ops.act_qdq_per_channel(x, 4, positive=half, clip='laplace', bit_alloc=True, out=y)
            elif a.config == 4:
                ops.pc_stats(x, x.shape[0], x.shape[1], x.shape[2] * x.shape[3], need_b=True, need_kurt=True, need_relu=True)
            else:
                ops.mid_tread_qdq(x, 4, clip=True, sym=False, want_entropy=True)
    fwd()
    torch.cuda.synchronize()
    best, host = 1e9, 1e9
    for _ in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fwd()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best, host = min(best, t2 - t0), min(host, t1 - t0)
    route = 'no exchange' if a.plain else ('in-launch exchange' if D.xrank_exchange(None) is not None else 'collective')
    print('config %d b%d %-20s %.3f ms per forward (%.1f G elem/s), host %.3f ms = %.1f us per call, %d calls' % (
        a.config, a.batch, route, best * 1e3, elems / best / 1e9, host * 1e3, host / len(layers) * 1e6, len(layers)), flush=True)
    if not a.plain:
        ex = D.xrank_exchange(None)
        if ex is not None:
            print('healthy', ex.healthy())
            ex.close()
        from cnn_quantization_amd import rccl
        rccl.close_all()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
