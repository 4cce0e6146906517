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
        # (configs 3 / 5 and the seven statistics of a sharded run: since round 6 their sums meet inside the single launch too -
        #  tests/test_sharded_single_gpu.py)
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
            seq_before = int(ex.seq_dev[0].item())
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
            same = same and int(ex.seq_dev[0].item()) == seq_before + 3 * len(xs)     # the device word advanced once per replayed launch
            same = same and ex.healthy()
        torch.cuda.current_stream().wait_stream(side)
    torch.save({'ok': ok, 'same': same, 'replays': replays}, os.path.join(tmp, 'graph.pt'))
    if ex is not None:
        ex.close()
    dist.destroy_process_group()


def _graph_after_larger(rank, world, port, tmp):
    """a graph that holds a SINGLE exchange launch, captured after eager launches with MORE channels, replayed
    five times, with an eager launch between capture and first replay and another between two replays.  With the host's
    bookkeeping of round 5 the slots [zero_c, C) of the eager launches' parities stayed dirty and a later launch folded stale
    extrema; the launches now record and clean their slot counts on the device (cdev)."""
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['CNNQ_XRANK'] = '1'
    os.environ['CNNQ_XRANK_TIMEOUT_MS'] = '3000'
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from cnn_quantization_amd import ops, distributed as D
    ops.reload_switches()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ex = D.xrank_exchange(None)
    ok, same, rest = ex is not None, True, -1
    if ok:
        # two ranks sharing the GPU, each with its half of every batch: a stale slot of the PEER is what a dirty parity shows
        def shard(x):
            n0, n1 = D.shard_batch(x.shape[0], rank, world)
            return x[n0:n1].contiguous().cuda()
        big = [shard(_batch(20 + i, (12, 96, 14, 14)) * (1 + i)) for i in range(3)]       # 96 channels, different ranges
        small = shard(_batch(31, (40, 6, 56, 56)))                                          # 6 channels
        small2 = shard(_batch(32, (40, 6, 56, 56)) * 7.)                                    # the same slots, other extrema
        ref = lambda x: ops.minmax_qdq_fused(x, x.shape[0], x.shape[1], x.shape[2] * x.shape[3], 4, False, _xrank=False)
        refs_big, ref_small, ref_small2 = [ref(x) for x in big], ref(small), ref(small2)
        ys = torch.empty_like(small)
        with torch.cuda.stream(side):
            for x in big:
                ops.act_qdq_per_channel(x, 4)                                               # eager, C = 96
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                ops.act_qdq_per_channel(small, 4, out=ys)                                   # ONE captured launch, C = 6
            same = same and bool(torch.equal(ops.act_qdq_per_channel(big[1], 4), refs_big[1]))   # eager between capture and replay
            for rep in range(5):
                ys.zero_()
                g.replay()
                side.synchronize()
                same = same and bool(torch.equal(ys, ref_small))
                if rep == 2:
                    same = same and bool(torch.equal(ops.act_qdq_per_channel(big[2], 4), refs_big[2]))   # eager between replays
                    same = same and bool(torch.equal(ops.act_qdq_per_channel(small2, 4), ref_small2))
            for x, r in zip(big, refs_big):                                                 # and eager again afterwards
                same = same and bool(torch.equal(ops.act_qdq_per_channel(x, 4), r))
            side.synchronize()
            same = same and ex.healthy()
            # every parity's slots are zero again once two more launches have passed (the trail the launches keep themselves)
            for _ in range(2):
                ops.act_qdq_per_channel(small, 4)
            side.synchronize()
            cdev = ex.seq_dev[4:8].tolist()
            rest = sum(1 for v in cdev if v != 0)
        torch.cuda.current_stream().wait_stream(side)
    torch.save({'ok': ok, 'same': same, 'rest': rest}, os.path.join(tmp, 'graph2_%d.pt' % rank))
    dist.barrier()
    if ex is not None:
        ex.close()
    dist.destroy_process_group()


def test_single_captured_launch_after_eager_launches_with_more_channels(tmp_path):
    port = 37700 + os.getpid() % 1500
    mp.spawn(_graph_after_larger, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rk in range(2):
        r = torch.load(os.path.join(str(tmp_path), 'graph2_%d.pt' % rk))
        assert r['ok'] and r['same'], rk
        assert 0 <= r['rest'] <= 2      # at most the two launches still ahead of their clean-up hold slots


def test_exchange_launches_replay_from_a_graph(tmp_path):
    """Round 4: the sequence number of the in-launch exchange lives in device memory (a one-thread kernel advances it behind
    every launch), so the launches can be captured and replayed: three replays of five tensors through the window of one
    forced rank, the collective path's bits every time."""
    port = 34100 + os.getpid() % 1500
    mp.spawn(_graph, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r = torch.load(os.path.join(str(tmp_path), 'graph.pt'))
    assert r['ok'] and r['same'] and r['replays'] == 3
