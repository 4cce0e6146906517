#!/usr/bin/env python3
"""Golden vectors for clip_type == 'mix' (int_quantizer.py:310-323) by RUNNING THE REFERENCE: `-sm use` quantization of
a per-channel activation whose statistics file carries mse_laplace / mse_gaus / mse_lowp columns (the reference's own
collection leaves them NaN; here they are injected into the loaded summary so that all three picks - laplace, gaus,
min/max - occur, NaN included).  Build container only (needs /root/reference); output tests/golden/mix.npz.

    python tests/golden/make_golden_mix.py
"""
import os
import sys
import tempfile
import types

import numpy as np

REF = os.environ.get('CNNQ_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))
os.environ['HOME'] = tempfile.mkdtemp(prefix='cnnq_golden_mix_')
sys.path.insert(0, REF)
sys.modules['int_quantization'] = types.ModuleType('int_quantization')

import torch  # noqa: E402
import pytorch_quantizer.quantization.qtypes.int_quantizer  # noqa: E402,F401

iq = sys.modules['pytorch_quantizer.quantization.qtypes.int_quantizer']
from pytorch_quantizer.quantization.inference import statistic_manager_perchannel as smpc  # noqa: E402

_spy = {}
_orig_entropy = iq.shannon_entropy


def _entropy_spy(t, *a, **k):
    _spy['codes'] = t.clone()
    return _orig_entropy(t, *a, **k)


iq.shannon_entropy = _entropy_spy


class _Logger:
    def log_metric(self, *a, **k):
        pass


def params(**kw):
    p = dict(clipping='no', stats_kind='mean', true_zero=False, kld=False, pcq_weights=False, pcq_act=True,
             bit_alloc_act=False, bit_alloc_weight=False, bit_alloc_rmode='round', bit_alloc_prior='gaus',
             bit_alloc_target_act=None, bit_alloc_target_weight=None, bcorr_act=False, bcorr_weight=False,
             vcorr_weight=False, logger=_Logger(), measure_entropy=True, mtd_quant=False)
    p.update(kw)
    return p


def main():
    g = torch.Generator().manual_seed(909)
    shape = (6, 12, 5, 7)
    C = shape[1]
    d = {}
    smpc.Singleton._instances.pop(smpc.StatisticManagerPerChannel, None)
    sm = smpc.StatisticManagerPerChannel('golden_mix', load_stats=False, stats=['max', 'min', 'std', 'mean', 'kurtosis', 'b', 'std_pos'])
    xs = []
    for k in range(3):
        u = torch.rand(shape, generator=g) - 0.5
        x = (-torch.sign(u) * torch.log1p(-2 * u.abs()) * torch.exp(torch.randn(1, C, 1, 1, generator=g) * 0.7)
             + torch.randn(1, C, 1, 1, generator=g) * 0.3).float()
        xs.append(x)
        d['x%d' % k] = x
        sm.save_tensor_stats(x, 'activation', 'conv0_activation')
    sm.__exit__()
    import pickle
    summ = pickle.load(open(os.path.join(sm.folder, 'golden_mix_statistics_perchannel_summary.pkl'), 'rb'))
    df = summ['conv0_activation']
    d['summary_columns'] = np.array(list(df.columns))
    d['summary_values'] = df.values.astype(np.float32)
    smpc.Singleton._instances.pop(smpc.StatisticManagerPerChannel, None)
    sm2 = smpc.StatisticManagerPerChannel('golden_mix', load_stats=True)
    # the three error columns: every ordering of the three, ties, NaN (a comparison with NaN is False)
    nan = np.nan
    mse_l = np.array([1., 2., 3., 1., 2., 3., 1., 1., nan, 1., nan, 2.], dtype=np.float32)
    mse_g = np.array([2., 1., 2., 3., 3., 1., 1., 2., 1., nan, nan, 2.], dtype=np.float32)
    mse_p = np.array([3., 3., 1., 2., 1., 2., 1., 1., 2., 2., 1., nan], dtype=np.float32)
    d['mse_laplace'], d['mse_gaus'], d['mse_lowp'] = mse_l, mse_g, mse_p
    st = sm2.stats['conv0_activation']
    st['mean_mse_laplace'], st['mean_mse_gaus'], st['mean_mse_lowp'] = mse_l, mse_g, mse_p
    names = []
    for half in (False, True):
        for baa in (False, True):
            q = iq.int_quantizer('int4', params(clipping='mix', bit_alloc_act=baa))
            q.half_range = half
            y = q(xs[0], 'conv0_activation', 'activation', stat_id='conv0_activation')
            nm = 'mix_half%d_baa%d' % (half, baa)
            names.append(nm)
            d[nm + '_y'] = y
            d[nm + '_codes'] = _spy['codes'].view(C, shape[0], shape[2], shape[3]).transpose(0, 1).contiguous().to(torch.int32)
    # per-tensor branch (pcq_act off): scalar statistics; the per-channel file's columns reduced to one value each would
    # not be the reference's per-tensor file, so this branch is pinned with a hand-made single-row table
    d['names'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'mix.npz'), **{k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in d.items()})
    print('mix.npz:', len(d), 'arrays;', names)


if __name__ == '__main__':
    main()
