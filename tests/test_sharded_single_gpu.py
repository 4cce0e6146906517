"""Round 6: configs 3 / 5 / 4 of a BATCH SHARD with the cross-rank exchange inside the single launch (cnnq_pc_aciq_fused_xrank,
cnnq_pc_midtread_fused_xrank, cnnq_pc_stats_xrank; csrc/cnnq_xrank.hip.h, cnnq_aciq.hip.h, cnnq_stats1.hip.h).  The reference
has no counterpart (inference_sim.py:196-200: DataParallel replicas quantize with their own sub-batch's statistics); what is
reproduced is int_quantizer.py:327-359 / 185-225 and statistic_manager_perchannel.py:45-79 on the GLOBAL batch.

No multi-GPU node has been available, so - as for config 2 (test_xrank_gpu.py) - TWO processes share ONE GPU (gloo rendezvous,
CNNQ_XRANK=1).  What must hold, per tile shape (flat tiles, row pieces, straddling 7x7 rows, a shape with no single-launch plan:
the chain's passes around the same window slots, uneven shards):

* every rank ends up with the SAME tables, bit for bit (the ranks' sums are added in rank order by every reader);
* the table is the global batch's: extrema exact, mean / std / b / kurtosis / std_pos within the fp64-sum tier of a reduction
  over the whole batch;
* given that table, everything downstream is the ORACLE's arithmetic on the whole batch, bit for bit: config 3's alpha / delta /
  offset / scale / zero point / qmax / y (tests/_direct.aciq_on_table), config 5's omega / alpha multiplier / Delta / clamp
  bounds / y and the entropy of the global batch's codes (midtread_on_table);
* the forced recompute path of the LOCAL meeting (flag 1) changes no bit; no wait for a peer expired; every call consumed exactly
  one launch number on every rank.
And ONE rank forced through the exchange (CNNQ_FORCE_EXCHANGE=1) reproduces the one-GPU single launch bit for bit (one rank's sum
IS the total).  Needs an MI355X: `pytest -m gpu`."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import bits_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (N, C, H, W): flat tiles / row pieces of one channel / whole channels per workgroup + exchange over the batch splits /
# straddling float4s (A = 4) / no single-launch plan (H*W = 45) / uneven shards of a flat shape / straddling rows too large for
# the statistics' single launch to pay (> 8 MB: the chain's passes around the window slots)
SHAPES = [(40, 6, 56, 56), (24, 3, 32, 32), (70, 12, 14, 14), (130, 24, 7, 7), (7, 16, 5, 9), (33, 5, 28, 28), (260, 200, 7, 7)]


def acts(shape, seed, relu=False):
    gen = torch.Generator().manual_seed(seed)
    C = shape[1]
    x = torch.empty(shape).exponential_(1.0, generator=gen) * (torch.rand(shape, generator=gen) < 0.5).float().mul_(2).sub_(1)
    x = x * (torch.rand(1, C, 1, 1, generator=gen) * 3 + 0.05) + torch.randn(1, C, 1, 1, generator=gen) * 0.3
    if relu:
        x = x.clamp_(min=0)
    if C > 2:
        x[:, C // 2] = 0.25 if not relu else 0.          # a constant channel
    return x.contiguous()


def _same(a, b):
    """bitwise equality of two device tensors (NaN == NaN: the kurtosis of a constant channel is NaN, as in the reference)"""
    it = torch.int32 if a.dtype == torch.float32 else torch.int64
    return a.shape == b.shape and bool(torch.equal(a.contiguous().view(it), b.contiguous().view(it)))


def _worker(rank, world, port, tmp, force):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['CNNQ_XRANK'] = '1'
    os.environ['CNNQ_XRANK_TIMEOUT_MS'] = '3000'
    if force:
        os.environ['CNNQ_FORCE_EXCHANGE'] = '1'
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from cnn_quantization_amd import _lib as L, ops, distributed as D
    ops.reload_switches()
    ex = D.xrank_exchange(None)
    out = {'ok': ex is not None, 'why': getattr(ex, 'why', '')}
    if ex is not None:
        tabs_mt = ops._midtread_tables(torch.device('cuda', 0))
        for i, shape in enumerate(SHAPES):
            N, C, H, W = shape
            HW = H * W
            for half in (False, True):
                x = acts(shape, 300 + i, relu=half)
                n0, n1 = D.shard_batch(N, rank, world)
                xs = x[n0:n1].contiguous().cuda()
                n = n1 - n0
                key = '%d_%d' % (i, half)
                s0 = ex.seq
                # config 3, with and without bit allocation; flag 1: the local meeting skipped, every member's sum recomputed
                for ba in (False, True):
                    y, p = ops.act_qdq_per_channel(xs, 4, positive=half, clip='laplace', bit_alloc=ba, want_parts=True)
                    cfg = ops._params_cfg(4, half, 'laplace', ba, False, None, True, False)
                    y1, p1 = ops._aciq_qdq_xrank(xs, n, C, HW, cfg, None, True, None, flags=1)
                    out['c3_%s_%d' % (key, ba)] = (y.cpu(), p['stats'].cpu(), p['qp'].cpu(), p['diag'].cpu(),
                                                   bool(torch.equal(y, y1)) and bool(torch.equal(p['stats'], p1['stats']))
                                                   and bool(torch.equal(p['qp'], p1['qp'])))
                # config 5 with the entropy of the GLOBAL batch's codes
                y, ent, p = ops.mid_tread_qdq(xs, 4, clip=True, sym=not half, want_entropy=True, want_parts=True)
                y1, ent1, p1 = ops._mid_tread_qdq_xrank(xs, n, C, HW, 4, not half, tabs_mt, None, True, True, flags=1)
                out['c5_' + key] = (y.cpu(), p['stats'].cpu(), p['mt'].cpu(), float(ent),
                                    bool(torch.equal(y, y1)) and bool(torch.equal(p['mt'], p1['mt'])) and abs(float(ent) - float(ent1)) < 1e-6)
                # config 4: the seven statistics
                st, mom = ops.pc_stats(xs, n, C, HW, need_b=True, need_kurt=True, need_relu=True)
                st1, mom1 = ops._pc_stats_xrank(xs, n, C, HW, True, True, True, None, flags=1)
                st2, _ = ops.pc_stats(xs, n, C, HW)          # extrema / mean / std only: three words per channel travel
                out['c4_' + key] = (st.cpu(), mom.cpu(), _same(st, st1) and _same(mom, mom1), st2.cpu())
                out['launches_' + key] = ex.seq - s0
        torch.cuda.synchronize()
        out['healthy'] = ex.healthy()
        st = ops.group_status(xs, clear=True)
        out['group_status'] = st
        if force:
            # one rank: the single launch of one GPU, bit for bit (stats / qp / y)
            same = True
            for i, shape in enumerate(SHAPES):
                N, C, H, W = shape
                x = acts(shape, 300 + i).cuda()
                os.environ['CNNQ_FORCE_EXCHANGE'] = '0'
                one = ops.aciq_qdq_single(x, N, C, H * W, 4, False, True, None, True, want_parts=True)
                st_one = ops.pc_stats_single(x, N, C, H * W, True, True, True, flags=0)
                os.environ['CNNQ_FORCE_EXCHANGE'] = '1'
                y, p = ops.act_qdq_per_channel(x, 4, clip='laplace', bit_alloc=True, want_parts=True)
                st_x, _ = ops.pc_stats(x, N, C, H * W, need_b=True, need_kurt=True, need_relu=True)
                if one is not None:
                    same = same and bool(torch.equal(one[0], y)) and bool(torch.equal(one[1]['stats'], p['stats'])) \
                        and bool(torch.equal(one[1]['qp'], p['qp']))
                if st_one is not None:
                    same = same and _same(st_one[0], st_x)
            out['same_as_one_gpu'] = same
    torch.save(out, os.path.join(tmp, 'rank%d.pt' % rank))
    dist.barrier()
    if ex is not None:
        ex.close()
    dist.destroy_process_group()


def _stat_rows(x):
    """fp64 reductions over the whole batch (rows of the table)."""
    C = x.shape[1]
    t = x.double().transpose(0, 1).reshape(C, -1)
    return t


def test_two_ranks_on_one_gpu_reproduce_the_global_batch(tmp_path):
    from cnn_quantization_amd import _lib as L
    from _direct import aciq_on_table, midtread_on_table
    world = 2
    port = 35300 + os.getpid() % 1500
    mp.spawn(_worker, args=(world, port, str(tmp_path), False), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r)) for r in range(world)]
    assert all(p['ok'] for p in parts), [p['why'] for p in parts]
    assert all(p['healthy'] for p in parts), 'a wait for a peer expired'
    # the test hook was used (bit 1); no local wait expired (bit 0)
    assert all(p['group_status'] & 1 == 0 for p in parts)
    for i, shape in enumerate(SHAPES):
        N, C, H, W = shape
        for half in (False, True):
            key = '%d_%d' % (i, half)
            x = acts(shape, 300 + i, relu=half)
            t64 = _stat_rows(x)
            # 2 x (c3 + its flagged twin) + (c5 + twin) + (c4 + twin + the short form) = 9 launch numbers on every rank
            assert all(p['launches_' + key] == 9 for p in parts), (shape, [p['launches_' + key] for p in parts])
            # ---- config 3
            for ba in (False, True):
                outs = [p['c3_%s_%d' % (key, ba)] for p in parts]
                st = outs[0][1]
                for o in outs[1:]:
                    assert bits_equal(o[1], st) and bits_equal(o[2], outs[0][2]) and bits_equal(o[3], outs[0][3]), (shape, half, ba)
                assert all(o[4] for o in outs), 'recompute path differs'
                assert torch.equal(st[L.STAT_MAX], x.amax(dim=(0, 2, 3))) and torch.equal(st[L.STAT_MIN], x.amin(dim=(0, 2, 3)))
                np.testing.assert_allclose(st[L.STAT_MEAN].double(), t64.mean(1), rtol=2e-6, atol=1e-7)
                np.testing.assert_allclose(st[L.STAT_STD].double(), t64.std(1), rtol=2e-6, atol=1e-9)
                b64 = (t64 - st[L.STAT_MEAN].double()[:, None]).abs().mean(1)
                np.testing.assert_allclose(st[L.STAT_B].double(), b64, rtol=2e-6, atol=1e-9)
                ref = aciq_on_table(x, st, outs[0][3][L.DIAG_BITS], 4, half, ba)
                qp, diag = outs[0][2], outs[0][3]
                assert bits_equal(diag[L.DIAG_ALPHA], ref['alpha']) and bits_equal(diag[L.DIAG_DELTA], ref['delta'])
                assert bits_equal(diag[L.DIAG_OFFSET], ref['offset'])
                assert bits_equal(qp[L.QP_SCALE], ref['scale']) and bits_equal(qp[L.QP_ZP], ref['zp']) and bits_equal(qp[L.QP_QMAX], ref['qmax'])
                y = torch.cat([o[0] for o in outs])
                assert bits_equal(y, ref['y']), (shape, half, ba)
            # ---- config 5
            outs = [p['c5_' + key] for p in parts]
            st, mt = outs[0][1], outs[0][2]
            for o in outs[1:]:
                assert bits_equal(o[1], st) and bits_equal(o[2], mt) and o[3] == outs[0][3], (shape, half)
            assert all(o[4] for o in outs), 'recompute path differs'
            ref = midtread_on_table(x, st, 4, not half)
            assert np.array_equal(mt[L.MT_OMEGA].numpy(), ref['omega'].numpy()), (shape, half)
            assert bits_equal(mt[L.MT_ALPHA], ref['alpha_mult']) and bits_equal(mt[L.MT_DELTA], ref['delta'])
            assert np.array_equal(mt[L.MT_CMAX].numpy(), ref['c_max'].numpy()) and np.array_equal(mt[L.MT_CMIN].numpy(), ref['c_min'].numpy())
            y = torch.cat([o[0] for o in outs])
            assert np.array_equal(y.numpy(), ref['y'].numpy()), (shape, half)        # (values: the sign of a zero clamp bound)
            nz = y.numpy() != 0
            assert bits_equal(y.numpy()[nz], ref['y'].numpy()[nz])
            assert abs(outs[0][3] - ref['entropy']) <= 2e-5 * max(1., ref['entropy']), (shape, half, outs[0][3], ref['entropy'])
            # ---- config 4
            outs = [p['c4_' + key] for p in parts]
            st, mom = outs[0][0], outs[0][1]
            for o in outs[1:]:
                assert bits_equal(o[0], st) and np.array_equal(o[1].numpy().view(np.int64), mom.numpy().view(np.int64)), (shape, half)
            assert all(o[2] for o in outs), 'recompute path differs'
            assert torch.equal(st[L.STAT_MAX], x.amax(dim=(0, 2, 3))) and torch.equal(st[L.STAT_MIN], x.amin(dim=(0, 2, 3)))
            assert float(mom[L.MOM_COUNT][0]) == N * H * W
            mean = st[L.STAT_MEAN].double()
            np.testing.assert_allclose(mean, t64.mean(1), rtol=2e-6, atol=1e-7)
            np.testing.assert_allclose(st[L.STAT_STD].double(), t64.std(1), rtol=2e-6, atol=1e-9)
            np.testing.assert_allclose(st[L.STAT_B].double(), (t64 - mean[:, None]).abs().mean(1), rtol=2e-6, atol=1e-9)
            np.testing.assert_allclose(st[L.STAT_STD_POS].double(), t64.clamp(min=0).std(1), rtol=2e-6, atol=1e-9)
            assert bool(torch.isfinite(st[:L.STAT_KURT]).all()) and bool(torch.isfinite(st[L.STAT_STD_POS]).all())
            sd = st[L.STAT_STD].double()
            live = sd > 0
            z = (t64 - mean[:, None]) / sd[:, None]
            kurt = (z ** 4).mean(1) - 3.
            np.testing.assert_allclose(st[L.STAT_KURT].double()[live], kurt[live], rtol=2e-5, atol=2e-5)
            short = outs[0][3]
            for r in (L.STAT_MIN, L.STAT_MAX, L.STAT_MEAN, L.STAT_STD):
                assert bits_equal(short[r], st[r]), (shape, half, r)


def test_one_rank_forced_through_the_exchange_equals_the_one_gpu_launch(tmp_path):
    port = 36900 + os.getpid() % 1500
    mp.spawn(_worker, args=(1, port, str(tmp_path), True), nprocs=1, join=True)
    r = torch.load(os.path.join(str(tmp_path), 'rank0.pt'))
    assert r['ok'], r['why']
    assert r['healthy'] and r['same_as_one_gpu']


def _graph_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CNNQ_XRANK='1', CNNQ_FORCE_EXCHANGE='1')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=0, world_size=1)
    from cnn_quantization_amd import ops, distributed as D
    ops.reload_switches()
    side = torch.cuda.Stream()                           # the exchange binds to ONE stream: a capturable one from the start
    with torch.cuda.stream(side):
        ex = D.xrank_exchange(None)
    ok, same = ex is not None, True
    if ok:
        shapes = [(40, 6, 56, 56), (70, 12, 14, 14), (130, 24, 7, 7), (7, 16, 5, 9)]
        xs = [acts(sh, 400 + i).cuda() for i, sh in enumerate(shapes)]
        ys = [torch.empty_like(x) for x in xs]
        with torch.cuda.stream(side):
            refs, rstats = [], []
            for x, y in zip(xs, ys):                     # eagerly once: workspaces and the sequence words exist
                refs.append(ops.act_qdq_per_channel(x, 4, clip='laplace', bit_alloc=True).clone())
                rstats.append(ops.pc_stats(x, x.shape[0], x.shape[1], x.shape[2] * x.shape[3], need_b=True, need_kurt=True, need_relu=True)[0].clone())
            side.synchronize()
            seq_before = int(ex.seq_dev[0].item())
            g = torch.cuda.CUDAGraph()
            sts = []
            with torch.cuda.graph(g, stream=side):
                for x, y in zip(xs, ys):
                    ops.act_qdq_per_channel(x, 4, clip='laplace', bit_alloc=True, out=y)
                    sts.append(ops.pc_stats(x, x.shape[0], x.shape[1], x.shape[2] * x.shape[3], need_b=True, need_kurt=True, need_relu=True)[0])
            for rep in range(3):
                for y in ys:
                    y.zero_()
                g.replay()
                side.synchronize()
                same = same and all(bool(torch.equal(y, r)) for y, r in zip(ys, refs))
                same = same and all(_same(a, b) for a, b in zip(sts, rstats))
            same = same and int(ex.seq_dev[0].item()) == seq_before + 3 * 2 * len(xs)      # one launch number per call and replay
            same = same and int(ex.seq_dev[4:8].abs().sum()) == 0                          # every launch cleaned up behind itself
            same = same and ex.healthy()
        torch.cuda.current_stream().wait_stream(side)
    torch.save({'ok': ok, 'same': same}, os.path.join(tmp, 'graph.pt'))
    if ex is not None:
        ex.close()
    dist.destroy_process_group()


def test_sharded_configs_3_and_4_replay_from_a_graph(tmp_path):
    """The sharded single-call routes are capturable (device-side launch numbers, the slot bookkeeping on the device): config 3
    and the seven statistics of four shard shapes (flat tiles, row pieces, straddling rows, no plan) captured once, replayed
    three times through the window of one forced rank - the eager bits every time."""
    port = 33600 + os.getpid() % 1500
    mp.spawn(_graph_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r = torch.load(os.path.join(str(tmp_path), 'graph.pt'))
    assert r['ok'] and r['same']
