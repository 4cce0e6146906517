"""Host logic without a GPU: the dispatch of IntQuantizer (which pipeline, which flags, for which
attribute combination - reference int_quantizer.py:92-122 and the branches below it) checked with a
recording stand-in for cnn_quantization_amd.ops, and the statistics manager's accumulation / summary /
file format checked against what the REFERENCE wrote for the same batches (tests/golden/collect.npz)."""
import os
import pickle

import numpy as np
import pytest
import torch


def qparams(**kw):
    p = dict(clipping='no', stats_kind='mean', true_zero=False, kld=False, pcq_weights=False, pcq_act=True,
             bit_alloc_act=False, bit_alloc_weight=False, bit_alloc_rmode='round', bit_alloc_prior='gaus',
             bit_alloc_target_act=None, bit_alloc_target_weight=None, bcorr_act=False, bcorr_weight=False,
             vcorr_weight=False, logger=None, measure_entropy=False, mtd_quant=False)
    p.update(kw)
    return p


class Recorder:
    """Stands in for the ops module: records (function, kwargs) and returns tensors of the right shape."""

    def __init__(self):
        self.calls = []

    def act_qdq_per_channel(self, x, num_bits, **kw):
        self.calls.append(('act_qdq_per_channel', dict(kw, num_bits=num_bits)))
        return (x.clone(), torch.tensor(1.5)) if kw.get('want_entropy') else x.clone()

    def minmax_qdq_per_tensor(self, x, num_bits, **kw):
        self.calls.append(('minmax_qdq_per_tensor', dict(kw, num_bits=num_bits)))
        return x.clone()

    def mid_tread_qdq(self, x, target, **kw):
        self.calls.append(('mid_tread_qdq', dict(kw, target=target)))
        return x.clone(), (torch.tensor(2.5) if kw.get('want_entropy') else None)


@pytest.fixture
def rec(monkeypatch):
    import sys
    import cnn_quantization_amd.qtypes  # noqa: F401
    iq = sys.modules['cnn_quantization_amd.qtypes.int_quantizer']   # the package attribute is the factory function
    r = Recorder()
    monkeypatch.setattr(iq, 'ops', r)
    return r


def make(bits, **kw):
    from cnn_quantization_amd.qtypes import int_quantizer
    return int_quantizer('int%d' % bits, qparams(**kw))


def test_factory_and_repr():
    from cnn_quantization_amd.qtypes import IntQuantizer, int_quantizer
    q = int_quantizer('int4', qparams(clipping='laplace', bit_alloc_act=True))
    assert isinstance(q, IntQuantizer) and q.num_bits == 4
    assert int_quantizer('int', qparams()).num_bits == 32
    assert repr(q) == ('IntQuantizer - [bits: 4, clipping: laplace, bit_alloc_act: True, bit_alloc_weight: False, '
                       'bit_alloc_round: True, pcq_w: False, pcq_a: True, bcorr_act: False, bcorr_weight: False, '
                       'vcorr_weight: False, kind: mean]')
    assert q.bit_alloc_target_act == 4 and make(4, bit_alloc_target_act=5.3).bit_alloc_target_act == 5.3


def test_dispatch_order(rec):
    x4, x2 = torch.zeros(2, 3, 5, 5), torch.zeros(2, 10)
    # per-channel activation, config 2; half_range / force_positive -> positive
    q = make(4)
    q.half_range = True
    q(x4, 'conv1_activation', 'activation')
    name, kw = rec.calls[-1]
    assert name == 'act_qdq_per_channel' and kw['clip'] == 'no' and kw['positive'] and kw['num_bits'] == 4
    # 1x1 spatial or 2-D -> per-tensor min/max; 'activation' tags average per-sample extrema, classifier does not
    q.half_range = False
    q(torch.zeros(2, 3, 1, 1), 'id', 'activation')
    assert rec.calls[-1][0] == 'minmax_qdq_per_tensor' and rec.calls[-1][1]['avg_over_batch'] is True
    q(x2, 'id', 'activation_classifier')
    assert rec.calls[-1][1]['avg_over_batch'] is False
    # clipping beats pcq_w and pcq_a; bit allocation flags travel
    q = make(4, clipping='laplace', bit_alloc_act=True, bit_alloc_prior='laplace', bit_alloc_rmode='ceil',
             bit_alloc_target_act=5.3, pcq_weights=True)
    q(x4, 'id', 'activation')
    name, kw = rec.calls[-1]
    assert name == 'act_qdq_per_channel' and kw['clip'] == 'laplace' and kw['bit_alloc'] and kw['prior_is_b'] \
        and kw['round_mode'] is False and kw['target'] == 5.3
    q(x2, 'id', 'activation_linear')                    # not per-channel -> whole-tensor clipping branch
    assert rec.calls[-1][1].get('whole_tensor') is True and rec.calls[-1][1]['bit_alloc'] is False
    # weights per output channel (never exchanged across ranks)
    q = make(4, pcq_weights=True, bit_alloc_weight=True, bit_alloc_target_weight=3)
    q(torch.zeros(8, 3, 3, 3), 'w', 'weight')
    name, kw = rec.calls[-1]
    assert name == 'act_qdq_per_channel' and kw['per_channel_dim'] == 0 and kw['group'] is False and kw['target'] == 3
    # mid-tread variants
    q = make(4, clipping='laplace', mtd_quant=True, bit_alloc_target_act=4)
    q.force_positive = True
    q(x4, 'id', 'activation')
    assert rec.calls[-1][0] == 'mid_tread_qdq' and rec.calls[-1][1]['sym'] is False and rec.calls[-1][1]['clip'] is True
    q = make(4, pcq_weights=True, mtd_quant=True)
    q(torch.zeros(8, 3, 3, 3), 'w', 'weight')
    assert rec.calls[-1][0] == 'mid_tread_qdq' and rec.calls[-1][1]['clip'] is False and rec.calls[-1][1]['sym'] is True


def test_override_att_is_temporary_even_on_error(rec):
    q = make(4, pcq_weights=True)
    q(torch.zeros(4, 3, 3, 3), 'w', 'weight', override_att=('num_bits', 8))
    assert rec.calls[-1][1]['num_bits'] == 8 and q.num_bits == 4
    rec.act_qdq_per_channel = lambda *a, **k: (_ for _ in ()).throw(RuntimeError('boom'))
    with pytest.raises(RuntimeError):
        q(torch.zeros(4, 3, 3, 3), 'w', 'weight', override_att=('num_bits', 8))
    assert q.num_bits == 4


def test_entropy_is_logged_with_reference_keys(rec):
    rows = []

    class Log:
        def log_metric(self, key, value, step=None, meterId=None, weight=1.):
            rows.append((key, value, step, meterId, weight))
    q = make(4, measure_entropy=True, logger=Log())
    x = torch.zeros(2, 3, 5, 5)
    q(x, 'conv3_activation', 'activation')
    assert rows[-1] == ('conv3_activation.entropy', 1.5, 'auto', 'avg.entropy.act', x.numel())
    qw = make(4, pcq_weights=True, measure_entropy=True, logger=Log())
    qw(torch.zeros(4, 3, 3, 3), 'm.weight', 'weight')
    assert rows[-1][0] == 'm.weight.entropy' and rows[-1][3] == 'avg.entropy.weight'


def test_dummy_quantizer_signature():
    from cnn_quantization_amd.qtypes import DummyQuantizer
    d = DummyQuantizer()
    t = torch.zeros(3)
    assert d(t, 'tag') is t and repr(d) == 'DummyQuantizer - fp32'
    with pytest.raises(TypeError):                     # the manager's 5-positional call, as in the reference
        d(t, 'id', 'tag', None, None)


def test_stats_manager_summary_matches_reference(golden, tmp_path, monkeypatch):
    """Feed the manager the per-batch statistics the reference computed (golden) through a stand-in for
    ops.pc_stats: accumulation (vstack per batch), min/mean/max summary, pickle layout and the -sm use
    lookups must equal the reference's."""
    from cnn_quantization_amd import _lib as L
    from cnn_quantization_amd.inference import statistic_manager_perchannel as M
    from cnn_quantization_amd.utils.misc import Singleton
    g = golden('collect')
    monkeypatch.setenv('HOME', str(tmp_path))
    rows = {'max': L.STAT_MAX, 'min': L.STAT_MIN, 'std': L.STAT_STD, 'mean': L.STAT_MEAN, 'kurtosis': L.STAT_KURT,
            'b': L.STAT_B, 'std_pos': L.STAT_STD_POS}
    batch = [0]

    def fake_pc_stats(x, N, C, HW, **kw):
        table = torch.zeros(L.NSTAT, C)
        for name, row in rows.items():
            table[row] = torch.from_numpy(g.np('b0_%s' % name)[batch[0]])
        return table, None
    monkeypatch.setattr(M.ops, 'pc_stats', fake_pc_stats)
    Singleton.reset(M.StatisticManagerPerChannel)
    sm = M.StatisticManagerPerChannel('golden_arch_0', load_stats=False)
    for k in range(3):
        batch[0] = k
        sm.save_tensor_stats(g.t('b0_x%d' % k), 'activation', 'conv0_activation')
    sm.save_tensor_stats(torch.zeros(4, 10), 'activation_linear', 'linear0_activation')     # skipped: FC
    sm.save_tensor_stats(torch.zeros(4, 6, 1, 1), 'activation', 'conv9_activation')          # skipped: 1x1
    assert sorted(sm.stats) == ['conv0_activation']
    sm.__exit__()
    path = os.path.join(str(tmp_path), 'mxt-sim/statistics/per_channel/golden_arch_0',
                        'golden_arch_0_statistics_perchannel_summary.pkl')
    df = pickle.load(open(path, 'rb'))['conv0_activation']
    assert list(df.columns) == [str(c) for c in g.np('b0_summary_columns')]
    assert [str(t) for t in df.dtypes] == [str(t) for t in g.np('b0_summary_dtypes')]
    assert np.array_equal(df.values.astype(np.float32).view(np.uint32), g.np('b0_summary_values').view(np.uint32))
    Singleton.reset(M.StatisticManagerPerChannel)
    sm2 = M.StatisticManagerPerChannel('golden_arch_0', load_stats=True)
    assert np.array_equal(np.asarray(sm2.get_tensor_stat('conv0_activation', 'max', 'mean')), g.np('b0_use_mean_max'))
    assert np.array_equal(np.asarray(sm2.get_tensor_stat('conv0_activation', 'min', 'min')), g.np('b0_use_min_min'))
    assert M.StatisticManagerPerChannel() is sm2                                   # singleton: later calls need no args
    Singleton.reset(M.StatisticManagerPerChannel)


def test_sorted_nicely_and_singleton():
    from cnn_quantization_amd.utils.misc import Singleton, sorted_nicely
    assert sorted_nicely(['conv10_activation', 'conv2_activation', 'conv1_activation']) == \
        ['conv1_activation', 'conv2_activation', 'conv10_activation']

    class A(metaclass=Singleton):
        def __init__(self, v=0):
            self.v = v
    a = A(3)
    assert A() is a and A(9).v == 3
    Singleton.reset(A)
    assert A(9).v == 9
    Singleton.reset(A)
