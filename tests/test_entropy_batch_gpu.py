"""ops.entropy_batch (round 6): the entropies of the codes of every tensor quantized inside the block (-me; int_quantizer.py:445,
179, 217 log the value, nothing consumes it mid-forward) come from ONE launch at the end of the block instead of a dependent
one-workgroup launch behind every tensor.  Same arithmetic, so: the same bits as the per-tensor launches - config 2 (with and
without the codes), config 3's single launch, the mid-tread path - more tensors than a block's table sets (a flush in the middle),
the golden entropies of the reference, and the quantizer's logger called in layer order with the same values.
Needs an MI355X: `pytest -m gpu`."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(40, 6, 56, 56), (70, 40, 7, 7), (37, 24, 14, 14), (8, 32, 14, 14), (3, 16, 5, 9), (130, 4, 28, 28)]


@pytest.fixture(scope='module')
def ops():
    from cnn_quantization_amd import ops as _ops
    return _ops


def _x(i, shape):
    gen = torch.Generator().manual_seed(900 + i)
    C = shape[1]
    return (torch.randn(shape, generator=gen) * (torch.rand(1, C, 1, 1, generator=gen) * 3 + 0.2) + torch.randn(1, C, 1, 1, generator=gen)).cuda()


def test_batched_entropies_equal_the_per_tensor_launches(ops):
    xs = [_x(i, s) for i, s in enumerate(SHAPES)]
    ref2 = [ops.act_qdq_per_channel(x, 4, positive=bool(i & 1), want_entropy=True) for i, x in enumerate(xs)]
    ref3 = [ops.act_qdq_per_channel(x, 4, clip='laplace', bit_alloc=True, want_entropy=True) for x in xs]
    ref5 = [ops.mid_tread_qdq(x, 4, clip=True, sym=True, want_entropy=True) for x in xs]
    torch.cuda.synchronize()
    with ops.entropy_batch() as eb:
        got2 = [ops.act_qdq_per_channel(x, 4, positive=bool(i & 1), want_entropy=True) for i, x in enumerate(xs)]
        got3 = [ops.act_qdq_per_channel(x, 4, clip='laplace', bit_alloc=True, want_entropy=True) for x in xs]
        got5 = [ops.mid_tread_qdq(x, 4, clip=True, sym=True, want_entropy=True) for x in xs]
        assert eb.n + len(eb.mt) > 0                              # nothing has been launched for them yet
    torch.cuda.synchronize()
    for (y0, e0), (y1, e1) in zip(ref2 + ref3, got2 + got3):
        assert torch.equal(y0, y1) and float(e0) == float(e1)
    for (y0, e0), (y1, e1) in zip(ref5, got5):
        assert torch.equal(y0, y1) and float(e0) == float(e1)
    # the shared replica tables and the block's table sets are zero again
    st = ops._raw_stream(xs[0].device.index)
    assert int(ops._ENT_TABLES[(xs[0].device.index, st)].abs().sum()) == 0
    assert int(ops._hist_replicas(xs[0], st).abs().sum()) == 0


def test_more_tensors_than_table_sets(ops):
    x = _x(0, (8, 32, 14, 14))
    ref = float(ops.act_qdq_per_channel(x, 4, want_entropy=True)[1])
    n = ops.ENT_BATCH_CAP + 7
    with ops.entropy_batch():
        ents = [ops.act_qdq_per_channel(x, 4, want_entropy=True)[1] for _ in range(n)]
    torch.cuda.synchronize()
    assert all(float(e) == ref for e in ents)


def test_golden_entropies_through_the_batch(ops, golden):
    g = golden('act_pc')
    keys = [k[:-len('_entropy')] for k in g.keys() if k.endswith('_entropy') and k.startswith('cfg2_int4_s')][:6]
    assert keys
    with ops.entropy_batch():
        outs = []
        for key in keys:
            si = key.rsplit('_s', 1)[1].split('_')[0]
            x = g.t('x' + si).cuda()
            half = bool(g.np(key + '_half')) if (key + '_half') in g else False
            outs.append((key, ops.act_qdq_per_channel(x, 4, positive=half, want_entropy=True)))
    for key, (y, e) in outs:
        assert abs(float(e) - float(g.np(key + '_entropy'))) <= 2e-5 * max(1., float(g.np(key + '_entropy'))), key


def test_the_quantizer_logs_after_the_block_in_layer_order(ops):
    from cnn_quantization_amd.qtypes.int_quantizer import int_quantizer

    class Log:
        def __init__(self):
            self.rows = []

        def log_metric(self, key, value, step=None, meterId=None, weight=1.):
            self.rows.append((key, value, meterId, weight))
    params = dict(clipping='no', stats_kind='mean', true_zero=False, kld=False, pcq_weights=False, pcq_act=True,
                  bit_alloc_act=False, bit_alloc_weight=False, bit_alloc_rmode='round', bit_alloc_prior='gaus',
                  bit_alloc_target_act=None, bit_alloc_target_weight=None, bcorr_act=False, bcorr_weight=False,
                  vcorr_weight=False, measure_entropy=True, mtd_quant=False)
    xs = [_x(i, s) for i, s in enumerate(SHAPES[:4])]
    a, b = Log(), Log()
    qa = int_quantizer('int4', dict(params, logger=a))
    qb = int_quantizer('int4', dict(params, logger=b))
    ya = [qa(x, 'conv%d_activation' % i, 'activation') for i, x in enumerate(xs)]
    with ops.entropy_batch():
        yb = [qb(x, 'conv%d_activation' % i, 'activation') for i, x in enumerate(xs)]
        assert b.rows == []                                       # deferred: the values do not exist yet
    assert [r[0] for r in a.rows] == [r[0] for r in b.rows] and len(b.rows) == len(xs)
    assert a.rows == b.rows
    assert all(torch.equal(u, v) for u, v in zip(ya, yb))
