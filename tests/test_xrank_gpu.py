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
        # round 4: the codes, the entropy of the GLOBAL batch's codes and the parameters out of the same in-launch exchange
        for i, shape in enumerate(SHAPES):
            x = _batch(i, shape)
            n0, n1 = D.shard_batch(shape[0], rank, world)
            xs = x[n0:n1].contiguous().cuda()
            for half in (False, True):
                y, codes, ent, parts = ops.act_qdq_per_channel(xs, 4, positive=half, want_codes=True, want_entropy=True, want_parts=True)
                y2, ent2 = ops.act_qdq_per_channel(xs, 4, positive=half, want_entropy=True)
                out['o_%d_%d' % (i, half)] = (y.cpu(), codes.cpu(), float(ent), parts['qp'].cpu(), parts['stats'].cpu(), float(ent2),
                                              bool(torch.equal(y, y2)))
        out['launches_out'] = ex.seq - seq0 - out['launches']
        # configs 3 / 5 and the seven statistics of a sharded run move their moment records through the collective whatever
        # CNNQ_XRANK says (two routes, and the in-launch one is config 2's): same calls, switch on and off, bit for bit
        x = _batch(3, (38, 24, 14, 14))
        n0, n1 = D.shard_batch(38, rank, world)
        xs = x[n0:n1].contiguous().cuda()
        def stats_paths():
            r = [ops.act_qdq_per_channel(xs, 4, clip='laplace', bit_alloc=True),
                 ops.mid_tread_qdq(xs, 4, clip=True, sym=False)[0],
                 ops.pc_stats(xs, xs.shape[0], 24, 196, need_b=True, need_kurt=True, need_relu=True)[0]]
            torch.cuda.synchronize()
            return [t.cpu() for t in r]
        a = stats_paths()
        os.environ['CNNQ_XRANK'] = '0'
        ops.reload_switches()
        b = stats_paths()
        os.environ['CNNQ_XRANK'] = '1'
        ops.reload_switches()
        out['stats_same'] = all(bool(torch.equal(u, v)) for u, v in zip(a, b))
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
            # codes / entropy / parameters of the global batch from the sharded single launch
            ref2, rp = O.act_per_channel_qdq(x, 4, half_range=half, return_parts=True)
            eref = float(O.shannon_entropy(rp['codes']))
            outs = [p['o_%d_%d' % (i, half)] for p in parts]
            assert torch.equal(torch.cat([o[0] for o in outs]), torch.as_tensor(ref2)), (shape, half)
            assert torch.equal(torch.cat([o[1] for o in outs]).float(), rp['codes'].float()), (shape, half)
            for o in outs:
                assert abs(o[2] - eref) <= 2e-5 * max(1., eref) and abs(o[5] - eref) <= 2e-5 * max(1., eref), (shape, half, o[2], eref)
                assert torch.equal(o[3][0], rp['scale'].flatten()) and torch.equal(o[3][1], rp['zero_point'].flatten()), (shape, half)
                assert torch.equal(o[4][1], torch.as_tensor(rp['max']).flatten()) and o[6]
    assert all(p['launches_out'] == 2 * len(SHAPES) * 2 for p in parts)    # they too went through the in-launch exchange
    assert all(p['stats_same'] for p in parts)


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


def _graph(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['CNNQ_XRANK'] = '1'
    os.environ['CNNQ_FORCE_EXCHANGE'] = '1'
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=0, world_size=1)
    from cnn_quantization_amd import ops, distributed as D
    ops.reload_switches()
    side = torch.cuda.Stream()                           # the exchange binds to ONE stream: a capturable one from the start
    with torch.cuda.stream(side):
        ex = D.xrank_exchange(None)
    ok, same, replays = ex is not None, True, 0
    if ok:
        shapes = [(40, 6, 56, 56), (70, 40, 7, 7), (37, 24, 14, 14), (8, 32, 14, 14), (7, 16, 5, 9)]
        xs = [_batch(i, sh).cuda() for i, sh in enumerate(shapes)]
        ys = [torch.empty_like(x) for x in xs]
        refs = [ops.minmax_qdq_fused(x, x.shape[0], x.shape[1], x.shape[2] * x.shape[3], 4, False, _xrank=False) for x in xs]
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for x, y in zip(xs, ys):                     # eagerly once on this stream: workspaces and the sequence word exist
                ops.act_qdq_per_channel(x, 4, out=y)
            seq_before = int(ex.seq_dev.item())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for x, y in zip(xs, ys):
                    ops.act_qdq_per_channel(x, 4, out=y)
            for rep in range(3):                          # odd and even sequence numbers: both parities of the windows
                for y in ys:
                    y.zero_()
                g.replay()
                side.synchronize()
                replays += 1
                same = same and all(bool(torch.equal(y, r)) for y, r in zip(ys, refs))
            same = same and int(ex.seq_dev.item()) == seq_before + 3 * len(xs)     # the device word advanced once per replayed launch
            same = same and ex.healthy()
        torch.cuda.current_stream().wait_stream(side)
    torch.save({'ok': ok, 'same': same, 'replays': replays}, os.path.join(tmp, 'graph.pt'))
    if ex is not None:
        ex.close()
    dist.destroy_process_group()


def test_exchange_launches_replay_from_a_graph(tmp_path):
    """Round 4: the sequence number of the in-launch exchange lives in device memory (a one-thread kernel advances it behind
    every launch), so the launches can be captured and replayed: three replays of five tensors through the window of one
    forced rank, the collective path's bits every time."""
    port = 34100 + os.getpid() % 1500
    mp.spawn(_graph, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r = torch.load(os.path.join(str(tmp_path), 'graph.pt'))
    assert r['ok'] and r['same'] and r['replays'] == 3
