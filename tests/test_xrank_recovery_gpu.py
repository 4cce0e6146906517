"""Round 6: the recovery of the in-launch exchange lives in the PRODUCT (distributed.xrank_checkpoint, set_xrank_recovery), so that
CNNQ_XRANK=auto can be the default: a wait for a peer that expires on one rank mid-forward is found at the forward's own
synchronisation point by all ranks together, the group's exchange is closed everywhere, the batch is redone through the
collective and the job stays there.  Two processes share one GPU (gloo); the test hook CNNQ_XRANK_TEST_EXPIRE_AT raises rank 0's
status word at its n-th exchanging launch, exactly what an expired wait leaves behind (that launch and all later ones of rank 0
give up at once and write NaN; rank 0 keeps pushing, so rank 1 never waits in vain).

* a "forward" of quantizer calls (config 2 on six layer shapes, config 3 on two): the first attempt leaves NaN on rank 0 and
  nothing on rank 1, the checkpoint answers False on BOTH ranks, the redo holds no NaN, config 2's layers equal the oracle on the
  whole batch bit for bit, both ranks end on the collective (no exchange left, the next checkpoint answers True);
* the same through the harness (--sharded): one batch redone on every rank, finite logits, no exchange left.
Reference behaviour being replaced: inference_sim.py:196-200 (DataParallel replicas).  Needs an MI355X: `pytest -m gpu`."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = [(40, 6, 56, 56), (70, 40, 7, 7), (37, 24, 14, 14), (8, 32, 14, 14), (12, 64, 7, 7), (130, 4, 28, 28)]


def _batch(i, shape):
    gen = torch.Generator().manual_seed(500 + i)
    C = shape[1]
    return (torch.randn(shape, generator=gen) * (torch.rand(1, C, 1, 1, generator=gen) * 3 + 0.2)
            + torch.randn(1, C, 1, 1, generator=gen)).contiguous()


def _setup(rank, world, port, expire_at):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['CNNQ_XRANK'] = '1'
    os.environ['CNNQ_XRANK_TIMEOUT_MS'] = '3000'
    os.environ['CNNQ_XRANK_TEST_EXPIRE_AT'] = str(expire_at)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)


def _forward_worker(rank, world, port, tmp):
    _setup(rank, world, port, 0)
    from cnn_quantization_amd import ops, distributed as D
    ops.reload_switches()
    D.set_xrank_recovery('checkpoint')
    ex = D.xrank_exchange(None)                      # (verification happens here: the hook below counts launches from now)
    out = {'ok': ex is not None}
    if ex is not None:
        ex.expire_at = ex.calls + 4 if rank == 0 else 0       # rank 0, fourth launch of the forward
        shards = []
        for i, shape in enumerate(SHAPES):
            n0, n1 = D.shard_batch(shape[0], rank, world)
            shards.append(_batch(i, shape)[n0:n1].contiguous().cuda())

        def forward():
            ys = [ops.act_qdq_per_channel(x, 4) for x in shards]
            ys += [ops.act_qdq_per_channel(x, 4, clip='laplace', bit_alloc=True) for x in shards[:2]]
            torch.cuda.synchronize()
            return ys
        first = forward()
        out['first_nan'] = [bool(torch.isnan(y).any()) for y in first]
        out['checkpoint'] = D.xrank_checkpoint(None)
        redo = forward()
        out['redo_nan'] = any(bool(torch.isnan(y).any()) for y in redo)
        out['redo'] = [y.cpu() for y in redo]
        out['exchange_left'] = D.xrank_exchange(None) is not None
        out['checkpoint2'] = D.xrank_checkpoint(None)
        again = forward()                            # and the job stays on the collective: the same bits
        out['stable'] = all(bool(torch.equal(a, b)) for a, b in zip(again, redo))
    torch.save(out, os.path.join(tmp, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_an_expired_wait_mid_forward_is_recovered_at_the_checkpoint(tmp_path):
    from oracle import quant_oracle as O
    world = 2
    port = 38600 + os.getpid() % 1200
    mp.spawn(_forward_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r)) for r in range(world)]
    assert all(p['ok'] for p in parts)
    # the hook hit rank 0 from its fourth launch on.  (Rank 1 may see NaN later in the same forward: a rank that has given up
    # waiting runs ahead of its peers and its pushes lap their windows - bounded by one timeout, and everything since the last
    # checkpoint is redone anyway.)
    assert parts[0]['first_nan'][:3] == [False] * 3 and all(parts[0]['first_nan'][3:])
    assert parts[1]['first_nan'][:3] == [False] * 3
    assert [p['checkpoint'] for p in parts] == [False, False]          # every rank learns of it, together
    assert not any(p['redo_nan'] for p in parts)                       # no NaN escapes the redo
    assert not any(p['exchange_left'] for p in parts) and all(p['checkpoint2'] for p in parts)
    assert all(p['stable'] for p in parts)
    for i, shape in enumerate(SHAPES):                                 # config 2: the collective's bits = the oracle's
        ref = O.act_per_channel_qdq(_batch(i, shape), 4)
        y = torch.cat([p['redo'][i] for p in parts])
        assert torch.equal(y, torch.as_tensor(ref)), shape


def _harness_worker(rank, world, port, tmp):
    _setup(rank, world, port, 0)
    from cnn_quantization_amd import ops, distributed as D
    from cnn_quantization_amd.harness import inference_sim as H
    ops.reload_switches()
    ex = D.xrank_exchange(None)
    out = {'ok': ex is not None}
    if ex is not None:
        ex.expire_at = ex.calls + 12 if rank == 0 else 0     # mid-forward of the warm-up batch
        args = H.build_parser().parse_args(['-a', 'resnet18', '-b', '8', '--image-size', '64', '--batches', '1', '-pcq_a', '-pcq_w',
                                            '--qtype', 'int4', '-qw', 'int4', '--sharded'])
        res = H.run(args, quiet=True)
        out.update(redone=res['batches_redone'], finite=res['output_finite'], left=D.xrank_exchange(None) is not None,
                   rows=len(res['rows']), nan=int(torch.isnan(res['logits']).sum()), inf=int(torch.isinf(res['logits']).sum()))
    torch.save(out, os.path.join(tmp, 'h%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_the_harness_redoes_the_batch_and_stays_on_the_collective(tmp_path):
    world = 2
    port = 39900 + os.getpid() % 1200
    mp.spawn(_harness_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), 'h%d.pt' % r)) for r in range(world)]
    assert all(p['ok'] for p in parts)
    assert [p['redone'] for p in parts] == [1, 1]                      # both ranks redo the same batch, once
    assert all(p['finite'] for p in parts), [(p['nan'], p['inf']) for p in parts]
    assert not any(p['left'] for p in parts)
    assert all(p['rows'] > 10 for p in parts)
