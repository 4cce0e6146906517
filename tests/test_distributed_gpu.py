"""The product's distributed path on a real GPU: two processes (gloo rendezvous, both on cuda:0)
each quantize half of a batch with GLOBAL statistics; the concatenated result must equal the
single-process result on the whole batch - bit for bit for config 2 (exact min/max exchange),
within the statistics tier for ACIQ.  The same code runs over RCCL with one GPU per rank."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _full_batch():
    gen = torch.Generator().manual_seed(4242)
    x = torch.randn(12, 24, 14, 14, generator=gen) * (torch.rand(1, 24, 1, 1, generator=gen) * 3 + 0.2)
    return x.contiguous()


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from cnn_quantization_amd import ops, distributed as D
    x = _full_batch()
    n0, n1 = D.shard_batch(x.shape[0], rank, world)
    xs = x[n0:n1].contiguous().cuda()
    out = {}
    for half in (False, True):
        y, codes, ent = ops.act_qdq_per_channel(xs, 4, positive=half, want_codes=True, want_entropy=True)
        out['cfg2_y_%d' % half], out['cfg2_codes_%d' % half], out['cfg2_ent_%d' % half] = y.cpu(), codes.cpu(), ent.cpu()
    y = ops.act_qdq_per_channel(xs, 4, clip='laplace', bit_alloc=True)
    out['cfg3_y'] = y.cpu()
    st, mom = ops.pc_stats(xs, xs.shape[0], xs.shape[1], 14 * 14, need_b=True, need_kurt=True, need_relu=True)
    out['stats'] = st.cpu()
    # per-tensor calibration statistics (-sm collect without -pcq_a): global on every rank, one writer
    os.environ['HOME'] = tmp
    from cnn_quantization_amd.inference.statistic_manager import StatisticManager
    sm = StatisticManager('pt_dist', load_stats=False, batch_avg=True, kld_threshold=True)
    sm.save_tensor_stats(xs, 'act', 'layer0')
    out['pt_row'] = torch.from_numpy(sm.stats['layer0'][0].copy())
    out['pt_names'] = list(sm.stats_names)
    sm.__exit__()
    out['pt_file'] = os.path.exists(os.path.join(tmp, 'mxt-sim', 'statistics', 'pt_dist', 'pt_dist_summary.csv'))
    torch.cuda.synchronize()
    torch.save(out, os.path.join(tmp, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_gpu(tmp_path):
    from cnn_quantization_amd import ops
    from oracle import quant_oracle as O
    world = 2
    port = 29700 + os.getpid() % 1500
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r)) for r in range(world)]
    x = _full_batch()
    for half in (False, True):
        ref, rp = O.act_per_channel_qdq(x, 4, half_range=half, return_parts=True)
        y = torch.cat([p['cfg2_y_%d' % half] for p in parts])
        codes = torch.cat([p['cfg2_codes_%d' % half] for p in parts])
        assert torch.equal(y, ref)                                   # bit-identical to the full batch
        assert torch.equal(codes.float(), rp['codes'])
        ent_ref = O.shannon_entropy(rp['codes'].int())
        for p in parts:                                              # every rank holds the GLOBAL entropy
            assert abs(float(p['cfg2_ent_%d' % half]) - float(ent_ref)) < 1e-5
    # single-process run of the same product path on the whole batch
    y1 = ops.act_qdq_per_channel(x.cuda(), 4, clip='laplace', bit_alloc=True).cpu()
    y2 = torch.cat([p['cfg3_y'] for p in parts])
    assert ((y1 - y2).abs() > 1e-5).float().mean() < 1e-3
    st1, _ = ops.pc_stats(x.cuda(), 12, 24, 196, need_b=True, need_kurt=True, need_relu=True)
    for p in parts:
        assert torch.equal(p['stats'][:2], st1.cpu()[:2])            # min / max exact
        assert torch.allclose(p['stats'], st1.cpu(), rtol=1e-5, atol=1e-5)
    # the per-tensor statistics manager: both ranks hold the row a single process computes on the whole batch,
    # and the summary file exists once both have left __exit__
    import numpy as np
    from cnn_quantization_amd.inference.statistic_manager import StatisticManager
    from cnn_quantization_amd.utils.misc import Singleton
    os.environ['HOME'] = str(tmp_path / 'single')
    Singleton.reset(StatisticManager)
    sm = StatisticManager('pt_single', load_stats=False, batch_avg=True, kld_threshold=True)
    sm.save_tensor_stats(x.cuda(), 'act', 'layer0')
    ref_row = sm.stats['layer0'][0]
    names = list(sm.stats_names)
    for p in parts:
        assert p['pt_file'] and p['pt_names'] == names
        row = p['pt_row'].numpy()
        for nm in ('max', 'min', 'dim', 'kld_th'):
            assert row[names.index(nm)] == ref_row[names.index(nm)], nm
        np.testing.assert_allclose(row, ref_row, rtol=2e-5, atol=1e-6)
    assert torch.equal(parts[0]['pt_row'], parts[1]['pt_row'])
    Singleton.reset(StatisticManager)
