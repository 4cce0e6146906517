"""The IN-LAUNCH cross-rank exchange of config 2 (CNNQ_XRANK=1; csrc/cnnq_xrank.hip.h, cnnq_pc_minmax_qdq_xrank): with the
batch sharded, the single-launch kernels push their channels' extrema into every rank's hipIpc window and wait for the
others' inside the launch - one read of x.  No multi-GPU node has been available, so the protocol is exercised
  * with TWO processes on ONE GPU (gloo rendezvous, both on cuda:0; their launches run concurrently on the device): the
    concatenated result must equal the oracle on the whole batch bit for bit - tile shapes of all three kernels, uneven
    shards, and a shape without a single-launch kernel (two passes around the same window protocol), dozens of launches in
    a row on the two-parity windows - and neither rank may have seen a wait expire;
  * with ONE rank forced through the exchange (CNNQ_FORCE_EXCHANGE=1): same bits as the plain single launch.
Needs an MI355X: `pytest -m gpu`."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (N, C, H, W): flat tiles, group tiles (A = 4 / 1), whole channels, no single-launch kernel (H*W = 45), odd batch
SHAPES = [(40, 6, 56, 56), (70, 40, 7, 7), (37, 24, 14, 14), (8, 32, 14, 14), (12, 64, 7, 7), (7, 16, 5, 9), (130, 4, 28, 28)]


def _batch(i, shape):
    gen = torch.Generator().manual_seed(100 + i)
    N, C = shape[:2]
    return (torch.randn(shape, generator=gen) * (torch.rand(1, C, 1, 1, generator=gen) * 3 + 0.2)
            + torch.randn(1, C, 1, 1, generator=gen)).contiguous()


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['CNNQ_XRANK'] = '1'
    os.environ['CNNQ_XRANK_TIMEOUT_MS'] = '3000'
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from cnn_quantization_amd import ops, distributed as D
    ops.reload_switches()
    ex = D.xrank_exchange(None)                      # collective: windows, handles, verification against the collective path
    out = {'ok': ex is not None, 'why': getattr(ex, 'why', '')}
    if ex is not None:
        seq0 = ex.seq
        for rnd in range(3):                         # the same shapes again: parities alternate, workspaces are reused
            for i, shape in enumerate(SHAPES):
                x = _batch(i, shape)
                n0, n1 = D.shard_batch(shape[0], rank, world)
                xs = x[n0:n1].contiguous().cuda()
                for half in (False, True):
                    y = ops.act_qdq_per_channel(xs, 4, positive=half)
                    if rnd == 2:
                        out['y_%d_%d' % (i, half)] = y.cpu()
        out['launches'] = ex.seq - seq0
        torch.cuda.synchronize()
        out['healthy'] = ex.healthy()
        out['group_status'] = ops.group_status(xs)
    torch.save(out, os.path.join(tmp, 'rank%d.pt' % rank))
    dist.barrier()
    if ex is not None:
        ex.close()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_the_whole_batch(tmp_path):
    from oracle import quant_oracle as O
    world = 2
    port = 31200 + os.getpid() % 1500
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r)) for r in range(world)]
    assert all(p['ok'] for p in parts), [p['why'] for p in parts]          # the exchange verified on this box
    assert all(p['healthy'] for p in parts), 'a wait for a peer expired'
    assert all(p['launches'] == 3 * len(SHAPES) * 2 for p in parts)        # every call went through the in-launch exchange
    assert all(p['group_status'] == 0 for p in parts)
    for i, shape in enumerate(SHAPES):
        x = _batch(i, shape)
        for half in (False, True):
            ref = O.act_per_channel_qdq(x, 4, half_range=half)
            y = torch.cat([p['y_%d_%d' % (i, half)] for p in parts])
            assert torch.equal(y, torch.as_tensor(ref)), (shape, half)


def _single(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['CNNQ_XRANK'] = '1'
    os.environ['CNNQ_FORCE_EXCHANGE'] = '1'
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=0, world_size=1)
    from cnn_quantization_amd import ops, distributed as D
    ops.reload_switches()
    ex = D.xrank_exchange(None)
    ok = ex is not None
    same = True
    if ok:
        for i, shape in enumerate(SHAPES + [(64, 256, 56, 56), (64, 1024, 14, 14), (64, 64, 112, 112)]):
            x = _batch(i, shape).cuda()
            for half in (False, True):
                N, C, H, W = shape
                y = ops.act_qdq_per_channel(x, 4, positive=half)                       # through the window of one rank
                ref = ops.minmax_qdq_fused(x, N, C, H * W, 4, half, _xrank=False)      # statistics pass, all_gather, Q/DQ pass
                same = same and bool(torch.equal(y, ref))
                one = ops.minmax_qdq_single(x, N, C, H * W, 4, half)                   # the plain single launch, where there is one
                same = same and (one is None or bool(torch.equal(y, one)))
        torch.cuda.synchronize()
        same = same and ex.healthy() and ex.seq > 0
    torch.save({'ok': ok, 'same': same}, os.path.join(tmp, 'single.pt'))
    if ex is not None:
        ex.close()
    dist.destroy_process_group()


def test_one_rank_forced_through_the_exchange(tmp_path):
    port = 32800 + os.getpid() % 1500
    mp.spawn(_single, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r = torch.load(os.path.join(str(tmp_path), 'single.pt'))
    assert r['ok'] and r['same']
