"""The multi-GPU protocol on CPU: two gloo processes, each holding half of the batch, exchange
fp64 moment records with cnn_quantization_amd.distributed.all_gather_records and merge them in
rank order.  The merged statistics must equal those of the full batch computed in one process,
min/max exactly (so config 2 is bit-identical for any world size).

The record producer / merger used here is the ORACLE's (numpy) because the product's are HIP
kernels; what is under test is the sharding, the record layout and the exchange."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NMOM = 7


def moment_record(x):
    """[NMOM, C] float64 record of x [N, C, H, W] - same rows as include/cnnq_hip.h CNNQ_MOM_*."""
    C = x.shape[1]
    t = x.transpose(0, 1).reshape(C, -1).double()
    r = torch.clamp(t, min=0)
    return torch.stack([t.min(-1)[0], t.max(-1)[0], t.sum(-1), (t * t).sum(-1),
                        torch.full((C,), float(t.shape[1]), dtype=torch.float64), r.sum(-1), (r * r).sum(-1)])


def merge_records(recs):
    """Rank-ordered merge, the arithmetic of cnnq_pc_combine."""
    mn = recs[:, 0].min(0)[0]
    mx = recs[:, 1].max(0)[0]
    out = recs.sum(0)
    out[0], out[1] = mn, mx
    return out


def stats_from_record(m):
    cnt = m[4]
    mean = m[2] / cnt
    var = (m[3] - m[2] * mean) / (cnt - 1)
    return m[0].float(), m[1].float(), mean.float(), var.clamp(min=0).sqrt().float()


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from cnn_quantization_amd import distributed as D
    gen = torch.Generator().manual_seed(99)
    x = torch.randn(10, 6, 5, 7, generator=gen) * 2 + 0.3       # same full batch on every rank
    n0, n1 = D.shard_batch(x.shape[0], rank, world)
    local = moment_record(x[n0:n1])
    gathered = D.all_gather_records(local, None)
    assert gathered.shape == (world, NMOM, 6)
    merged = merge_records(gathered)
    full = moment_record(x)
    mn, mx, mean, std = stats_from_record(merged)
    fmn, fmx, fmean, fstd = stats_from_record(full)
    assert torch.equal(mn, fmn) and torch.equal(mx, fmx)          # exact
    assert torch.allclose(mean, fmean, rtol=1e-6, atol=1e-7) and torch.allclose(std, fstd, rtol=1e-6)
    assert float(merged[4][0]) == x.numel() / 6
    # histogram all-reduce (global entropy)
    h = torch.bincount(torch.randint(0, 16, (100,), generator=gen), minlength=256)
    hh = D.all_reduce_sum_(h.clone(), None)
    assert torch.equal(hh, h * world)
    # per-sample extrema tables (per-tensor path)
    st = torch.zeros(7, 5)
    st[0] = torch.arange(5.) + 10 * rank
    st[1] = torch.arange(5.) + 100 + 10 * rank
    m = D.merge_row_minmax(st, 5, True, None)
    assert m.shape == (7, 5 * world)
    assert m[0].tolist() == [float(v + 10 * r) for r in range(world) for v in range(5)]
    # uneven shards (10 samples over 3 ranks = 4 + 3 + 3): every rank ends up with all 10 rows in batch order
    per_sample_min = x.reshape(10, -1).min(-1)[0]
    per_sample_max = x.reshape(10, -1).max(-1)[0]
    rows = n1 - n0
    st = torch.zeros(7, rows)
    st[0], st[1] = per_sample_min[n0:n1], per_sample_max[n0:n1]
    m = D.merge_row_minmax(st, rows, True, None)
    assert m.shape == (7, 10)
    assert torch.equal(m[0], per_sample_min) and torch.equal(m[1], per_sample_max)
    # the exchange selection: 'auto' is the default (round 6: the recovery lives in the product, D.xrank_checkpoint); auto never
    # starts the in-launch exchange on a backend whose ranks may share a GPU; nothing here touches a device
    for mode in ('auto', '0', 'bogus'):
        os.environ['CNNQ_XRANK'] = mode
        assert D.xrank_mode() == ('auto' if mode == 'auto' else '0')
        assert D.xrank_exchange(None) is None
    os.environ.pop('CNNQ_XRANK')
    assert D.xrank_mode() == 'auto'
    assert D.xrank_exchange(None) is None
    assert D.xrank_checkpoint(None) is True                 # no exchange: nothing to check, no synchronisation
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, 'ok%d' % rank), 'w').write('ok')


@pytest.mark.parametrize('world', [2, 3])
def test_gloo_stats_exchange(tmp_path, world):
    port = 29500 + os.getpid() % 2000 + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), 'ok%d' % r)) for r in range(world))


def test_shard_batch_partitions():
    from cnn_quantization_amd import distributed as D
    for n in (1, 7, 512, 513):
        for w in (1, 2, 3, 8):
            parts = [D.shard_batch(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    assert D.world_size() == 1 and D.rank() == 0



def test_xrank_mode_without_a_process_group(monkeypatch):
    """No process group: every mode answers None (the collective / single-GPU paths), nothing is imported or allocated."""
    sys.path.insert(0, ROOT)
    from cnn_quantization_amd import distributed as D
    for mode, want in (('auto', 'auto'), ('1', '1'), ('0', '0'), ('', '0')):
        monkeypatch.setenv('CNNQ_XRANK', mode)
        assert D.xrank_mode() == want
        assert D.xrank_exchange(None) is None
    monkeypatch.delenv('CNNQ_XRANK')
    assert D.xrank_mode() == 'auto'                    # round 6: the default (D.xrank_checkpoint is the recovery)
    D.set_xrank_mode('0')                              # a program may still rule the in-launch exchange out for itself
    assert D.xrank_mode() == '0' and D.xrank_exchange(None) is None
    D.set_xrank_mode(None)
    assert D.xrank_mode() == 'auto'
    assert D.xrank_checkpoint(None) is True            # no process group: nothing to check
    for mode in ('raise', 'checkpoint', 'raise'):
        D.set_xrank_recovery(mode)
    with pytest.raises(ValueError):
        D.set_xrank_recovery('bogus')
