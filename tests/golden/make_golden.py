#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite use
the committed .npz files.  The reference is imported unmodified with one stub module
(``int_quantization`` - its CUDA extension cannot be built here) exactly as SURVEY.md
Appendix D describes; nothing of the reference's source is stored, only inputs and outputs.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Integer codes: the reference materialises them only as the argument of ``shannon_entropy``
(int_quantizer.py:586-587, :217), so the quantizer is built with measure_entropy=True and a
spy wrapped around that function records its argument.

CPU-only artefact avoided: ``get_alpha_mult`` doubles the caller's ``omega``
in place on CPU because ``.cpu().numpy()`` aliases; on the GPU (the reference's device) it
does not.  The driver hands it a clone so the fixtures encode GPU semantics.
"""
import os
import sys
import tempfile
import types

import numpy as np

REF = os.environ.get('CNNQ_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))

os.environ['HOME'] = tempfile.mkdtemp(prefix='cnnq_golden_home_')
sys.path.insert(0, REF)
sys.modules['int_quantization'] = types.ModuleType('int_quantization')

import torch  # noqa: E402

import pytorch_quantizer.quantization.qtypes.int_quantizer  # noqa: E402,F401

iq = sys.modules['pytorch_quantizer.quantization.qtypes.int_quantizer']
from pytorch_quantizer.quantization.inference import statistic_manager_perchannel as smpc  # noqa: E402

# ---- spy on the integer codes -------------------------------------------------------------
_spy = {}
_orig_entropy = iq.shannon_entropy


def _entropy_spy(t, *a, **k):
    _spy['codes'] = t.clone()
    return _orig_entropy(t, *a, **k)


iq.shannon_entropy = _entropy_spy

# ---- GPU semantics for get_alpha_mult (defensive clone in the driver) -----------------------
_orig_alpha_mult = iq.IntQuantizer.get_alpha_mult


def _alpha_mult_gpu_semantics(omega, sym=True):
    return _orig_alpha_mult(omega.clone(), sym=sym)


iq.IntQuantizer.get_alpha_mult = staticmethod(_alpha_mult_gpu_semantics)


class _Logger:
    def __init__(self):
        self.rows = []

    def log_metric(self, key, value, step=None, meterId=None, weight=1.):
        self.rows.append((key, value, meterId, weight))


def params(**kw):
    p = dict(clipping='no', stats_kind='mean', true_zero=False, kld=False, pcq_weights=False, pcq_act=True,
             bit_alloc_act=False, bit_alloc_weight=False, bit_alloc_rmode='round', bit_alloc_prior='gaus',
             bit_alloc_target_act=None, bit_alloc_target_weight=None, bcorr_act=False, bcorr_weight=False,
             vcorr_weight=False, logger=None, measure_entropy=False, mtd_quant=False)
    p.update(kw)
    return p


def laplace_nchw(gen, shape, mu_scale=0.5):
    """Per-channel Laplace(mu_c, b_c) activations."""
    N, C, H, W = shape
    mu = torch.randn(C, generator=gen) * mu_scale
    b = torch.exp(torch.empty(C).uniform_(math_log(0.05), math_log(2.0), generator=gen))
    u = torch.rand(shape, generator=gen) - 0.5
    x = mu.view(1, C, 1, 1) - b.view(1, C, 1, 1) * torch.sign(u) * torch.log1p(-2 * u.abs())
    return x.float().contiguous()


def math_log(v):
    import math
    return math.log(v)


def npy(v):
    if v is None:
        return np.zeros(0, dtype=np.float32)
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy()
    return np.asarray(v)


def save(name, d):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **{k: npy(v) for k, v in d.items()})
    print('wrote %s (%d arrays, %.1f KB)' % (path, len(d), os.path.getsize(path) / 1024.))


# ================================================================ (i) core Q/DQ, row a4
def gen_core():
    g = torch.Generator().manual_seed(101)
    d = {}
    idx = 0
    for bits in (2, 3, 4, 8):
        C, M = 12, 257
        t = torch.randn(C, M, generator=g) * torch.logspace(-2, 1.5, C).view(C, 1) + torch.linspace(-3, 3, C).view(C, 1)
        t[1] = t[1].abs() + 0.25          # all-positive channel
        t[2] = -t[2].abs() - 0.25         # all-negative channel
        t[3] = 0.75                       # constant channel, delta = 0
        t[4] = 0.                         # all zeros
        t[5, :16] = torch.tensor([float(k) + 0.5 for k in range(16)])   # exact .5 ties after scaling
        t[6, 0] = float('inf')
        t[6, 1] = -float('inf')
        delta = t.max(-1)[0] - t.min(-1)[0]
        offset = t.min(-1)[0].clone()
        delta[5], offset[5] = float(2 ** bits - 1), 0.          # scale exactly 1 -> ties survive
        delta[6], offset[6] = 8., -4.
        delta[7], offset[7] = 1e-9, 0.3                         # scale floored at 1e-8
        delta[8], offset[8] = 5., 2.5                           # zero outside [offset, offset+delta]
        q = iq.int_quantizer('int%d' % bits, params(measure_entropy=True))
        y, ent = q.__gemmlowpQuantize1__(t, delta, offset, measure_entropy=True)
        d['c%d_bits' % idx] = np.int64(bits)
        d['c%d_t' % idx] = t
        d['c%d_delta' % idx] = delta
        d['c%d_offset' % idx] = offset
        d['c%d_y' % idx] = y
        d['c%d_codes' % idx] = _spy['codes']
        d['c%d_entropy' % idx] = ent
        idx += 1
        # with a per-channel bit allocation (0..8 bits, includes 0-bit channels)
        ba = torch.tensor([0., 1., 2., 3., 4., 4., 5., 6., 7., 8., 2., 0.])
        y, ent = q.__gemmlowpQuantize1__(t, delta, offset, bit_alloc=ba, measure_entropy=True)
        d['c%d_bits' % idx] = np.int64(bits)
        d['c%d_t' % idx] = t
        d['c%d_delta' % idx] = delta
        d['c%d_offset' % idx] = offset
        d['c%d_bit_alloc' % idx] = ba
        d['c%d_y' % idx] = y
        d['c%d_codes' % idx] = _spy['codes']
        d['c%d_entropy' % idx] = ent
        idx += 1
    # per-tensor use with 0-dim delta/offset (int_quantizer.py:357)
    q = iq.int_quantizer('int8', params())
    x = torch.randn(2, 3, 5, 7, generator=g) * 2 + 0.3
    y = q.__gemmlowpQuantize1__(x, torch.tensor(float(x.max() - x.min())), torch.tensor(float(x.min())))
    d['pt_x'], d['pt_delta'], d['pt_offset'], d['pt_y'] = x, x.max() - x.min(), x.min(), y
    d['n_cases'] = np.int64(idx)
    save('core_qdq', d)


# ================================================================ (ii) a5 / a6 end to end
ACT_SHAPES = [(4, 8, 7, 7), (2, 64, 14, 14), (3, 16, 5, 9), (5, 6, 1, 3), (2, 20, 12, 12)]
ACT_CFGS = [
    # name, quantizer params, half_range
    ('cfg2_int4', dict(), False),
    ('cfg2_int4_half', dict(), True),
    ('cfg2_int8', dict(_bits=8), False),
    ('cfg2_int4_baa', dict(bit_alloc_act=True), False),
    ('cfg2_int4_baa_half', dict(bit_alloc_act=True), True),
    ('cfg3_laplace', dict(clipping='laplace'), False),
    ('cfg3_laplace_half', dict(clipping='laplace'), True),
    ('cfg3_laplace_baa', dict(clipping='laplace', bit_alloc_act=True), False),
    ('cfg3_laplace_baa_half', dict(clipping='laplace', bit_alloc_act=True), True),
    ('cfg3_laplace_baa_bap', dict(clipping='laplace', bit_alloc_act=True, bit_alloc_prior='laplace'), False),
    ('cfg3_laplace_baa_ceil', dict(clipping='laplace', bit_alloc_act=True, bit_alloc_rmode='ceil'), False),
    ('cfg3_laplace_baa_t53', dict(clipping='laplace', bit_alloc_act=True, bit_alloc_target_act=5.3), False),
    ('cfg3_gaus', dict(clipping='gaus'), False),
    ('cfg3_gaus_half', dict(clipping='gaus'), True),
    ('cfg3_2std', dict(clipping='2std'), False),
    ('cfg3_laplace_int3_baa', dict(clipping='laplace', bit_alloc_act=True, _bits=3), False),
    ('cfg3_laplace_int2', dict(clipping='laplace', _bits=2), True),
]


def gen_act():
    g = torch.Generator().manual_seed(202)
    d = {}
    names = []
    for si, shape in enumerate(ACT_SHAPES):
        x = laplace_nchw(g, shape)
        d['x%d' % si] = x
        for name, kw, half in ACT_CFGS:
            kw = dict(kw)
            bits = kw.pop('_bits', 4)
            q = iq.int_quantizer('int%d' % bits, params(measure_entropy=True, logger=_Logger(), **kw))
            q.half_range = half
            _spy.pop('codes', None)
            y = q(x, 'conv1_activation', 'activation')
            key = '%s_s%d' % (name, si)
            names.append(key)
            d[key + '_y'] = y
            C = shape[1]
            codes = _spy['codes'].view(C, shape[0], shape[2], shape[3]).transpose(0, 1).contiguous()
            d[key + '_codes'] = codes.to(torch.int32)
            d[key + '_entropy'] = np.float32(q.logger.rows[-1][1])
            d[key + '_half'] = np.int64(half)
            d[key + '_bits'] = np.int64(bits)
            # intermediates, recomputed through the reference's own helpers
            st = iq.IntQuantizer.__act_stats_perchannel__(x, ['min', 'max', 'b', 'std'], avg_over_batch=False)
            mean_ab = iq.IntQuantizer.__act_stats_perchannel__(x, ['mean'], avg_over_batch=True)['mean']
            mean_flat = iq.IntQuantizer.__act_stats_perchannel__(x, ['mean'], avg_over_batch=False)['mean']
            for s in st:
                d['s%d_stat_%s' % (si, s)] = st[s]
            d['s%d_stat_mean_avgbatch' % si] = mean_ab
            d['s%d_stat_mean' % si] = mean_flat
            if q.bit_alloc_act and bits <= 4:
                prior = 'std' if q.bit_alloc_prior == 'gaus' else 'b'
                ba = iq.IntQuantizer.get_bits_alloc_fixed_target(st[prior], q.bit_alloc_target_act, q.bit_alloc_round)
                d[key + '_bit_alloc'] = ba
            if q.clipping != 'no':
                alpha = q.get_alpha(x, 'activation', None, q.clipping, per_channel=True)
                rng, off = q.alpha2DeltaOffset(alpha, st['max'], st['min'], mean_ab)
                d[key + '_alpha'] = alpha
                d[key + '_range'] = np.asarray(rng, dtype=np.float32)
                d[key + '_offset'] = np.asarray(off, dtype=np.float32) * np.ones(C, dtype=np.float32)
    d['names'] = np.array(names)
    save('act_pc', d)


# ================================================================ (iii) bit allocation, row a8
def gen_bit_alloc():
    g = torch.Generator().manual_seed(303)
    d = {}
    k = 0
    for C in (8, 64, 256, 1024, 2048):
        for target, rnd in ((4, True), (4, False), (5.3, True), (3, True), (2, False)):
            std = torch.exp(torch.randn(C, generator=g) * 0.8)
            if C >= 64:
                std[3] = 0.   # a dead channel -> log2(0) = -inf -> 0 bits
            ba = iq.IntQuantizer.get_bits_alloc_fixed_target(std, target, rnd)
            d['k%d_std' % k], d['k%d_target' % k], d['k%d_round' % k] = std, np.float64(target), np.int64(rnd)
            d['k%d_bits' % k] = ba
            one = iq.IntQuantizer.get_bits_alloc(std, target, rnd)
            d['k%d_bits_single' % k] = one
            k += 1
    d['n_cases'] = np.int64(k)
    save('bit_alloc', d)


# ================================================================ (iv) tables
def gen_tables():
    q = iq.int_quantizer('int4', params())
    d = dict(omega_table=iq.omega_table, alpha_table=iq.alpha_table)
    for nm in ('alpha_gaus', 'alpha_gaus_positive', 'alpha_laplace', 'alpha_laplace_positive'):
        tab = getattr(q, nm)
        d[nm + '_keys'] = np.array(sorted(tab), dtype=np.int64)
        d[nm + '_vals'] = np.array([tab[k] for k in sorted(tab)], dtype=np.float64)
    save('tables', d)


# ================================================================ (v) mid-tread + entropy, rows a14 a15
def gen_midtread():
    g = torch.Generator().manual_seed(404)
    d = {}
    names = []
    for si, shape in enumerate([(4, 8, 7, 7), (2, 32, 14, 14), (3, 16, 5, 9)]):
        x = laplace_nchw(g, shape)
        d['x%d' % si] = x
        for name, target, half in (('sym_t4', 4, False), ('asym_t4', 4, True), ('sym_t53', 5.3, False),
                                   ('asym_t53', 5.3, True), ('sym_t2', 2, False)):
            lg = _Logger()
            q = iq.int_quantizer('int4', params(clipping='laplace', mtd_quant=True, measure_entropy=True,
                                                logger=lg, bit_alloc_target_act=target))
            q.half_range = half
            y = q(x, 'conv3_activation', 'activation')
            key = '%s_s%d' % (name, si)
            names.append(key)
            C = shape[1]
            d[key + '_y'] = y
            d[key + '_codes'] = _spy['codes'].view(C, shape[0], shape[2], shape[3]).transpose(0, 1).contiguous()
            d[key + '_entropy'] = np.float32(lg.rows[-1][1])
            d[key + '_target'] = np.float64(target)
            d[key + '_half'] = np.int64(half)
            # intermediates
            t = x.transpose(0, 1).contiguous().view(C, -1)
            std = t.std(-1)
            omega = iq.IntQuantizer.get_omega(std, target_bins=(2 ** target)).round()
            d[key + '_omega'] = omega
            d[key + '_alpha_mult'] = np.asarray(iq.IntQuantizer.get_alpha_mult(omega, sym=not half), dtype=np.float64)
    # weights: no clipping, symmetric (int_quantizer.py:147-156)
    for wi, wshape in enumerate([(16, 8, 3, 3), (10, 32)]):
        w = torch.randn(wshape, generator=g) * 0.1
        lg = _Logger()
        q = iq.int_quantizer('int4', params(pcq_weights=True, mtd_quant=True, measure_entropy=True, logger=lg,
                                            bit_alloc_target_weight=4))
        wq = q(w, 'layer.weight', 'weight')
        d['w%d' % wi], d['w%d_y' % wi] = w, wq
        d['w%d_codes' % wi] = _spy['codes'].view(wshape)
        d['w%d_entropy' % wi] = np.float32(lg.rows[-1][1])
    d['names'] = np.array(names)
    save('midtread', d)


# ================================================================ (vi) stats collection, row a16
def gen_collect():
    g = torch.Generator().manual_seed(505)
    d = {}
    for bi, batch_avg in enumerate((False, True)):
        smpc.Singleton._instances.pop(smpc.StatisticManagerPerChannel, None)
        sm = smpc.StatisticManagerPerChannel('golden_arch_%d' % bi, load_stats=False, batch_avg=batch_avg,
                                             stats=['max', 'min', 'std', 'mean', 'kurtosis', 'b', 'std_pos'])
        shape = (4, 6, 5, 7)
        xs = [laplace_nchw(g, shape) for _ in range(3)]
        for k, x in enumerate(xs):
            d['b%d_x%d' % (bi, k)] = x
            sm.save_tensor_stats(x, 'activation', 'conv0_activation')
        sm.save_tensor_stats(torch.randn(4, 10, generator=g), 'activation_linear', 'linear0_activation')      # skipped
        sm.save_tensor_stats(torch.randn(4, 6, 1, 1, generator=g), 'activation', 'conv9_activation')          # skipped
        for s in sm.stats_names:
            d['b%d_%s' % (bi, s)] = sm.stats['conv0_activation'][s]          # [3 batches, C]
        d['b%d_ids' % bi] = np.array(sorted(sm.stats.keys()))
        sm.__exit__()
        import pickle
        path = os.path.join(sm.folder, '%s_statistics_perchannel_summary.pkl' % sm.name)
        summ = pickle.load(open(path, 'rb'))
        df = summ['conv0_activation']
        d['b%d_summary_columns' % bi] = np.array(list(df.columns))
        d['b%d_summary_values' % bi] = df.values.astype(np.float32)
        d['b%d_summary_dtypes' % bi] = np.array([str(t) for t in df.dtypes])
        # -sm use lookups (statistic_manager_perchannel.py:127-133)
        smpc.Singleton._instances.pop(smpc.StatisticManagerPerChannel, None)
        sm2 = smpc.StatisticManagerPerChannel('golden_arch_%d' % bi, load_stats=True)
        d['b%d_use_mean_max' % bi] = np.asarray(sm2.get_tensor_stat('conv0_activation', 'max', 'mean'))
        d['b%d_use_min_min' % bi] = np.asarray(sm2.get_tensor_stat('conv0_activation', 'min', 'min'))
        if bi == 0:
            # full "-sm use" quantization through the stats file: cfg 3 flags, half and full range
            for half in (False, True):
                for nm, kw in (('use_cfg2', dict()), ('use_cfg3', dict(clipping='laplace', bit_alloc_act=True))):
                    q = iq.int_quantizer('int4', params(measure_entropy=True, logger=_Logger(), **kw))
                    q.half_range = half
                    y = q(xs[0], 'conv0_activation', 'activation', stat_id='conv0_activation')
                    d['%s_half%d_y' % (nm, half)] = y
                    C = shape[1]
                    d['%s_half%d_codes' % (nm, half)] = _spy['codes'].view(C, shape[0], shape[2], shape[3]) \
                        .transpose(0, 1).contiguous().to(torch.int32)
    smpc.Singleton._instances.pop(smpc.StatisticManagerPerChannel, None)
    save('collect', d)


# ================================================================ (vii) weights, rows a10 a11
def gen_weights():
    g = torch.Generator().manual_seed(606)
    d = {}
    k = 0
    for wshape in [(16, 8, 3, 3), (32, 16, 1, 1), (10, 64), (24, 3, 7, 7)]:
        w = torch.randn(wshape, generator=g) * torch.exp(torch.randn(wshape[0], generator=g) * 0.5).view(
            (-1,) + (1,) * (len(wshape) - 1)) * 0.05
        for bits, baw, target in ((4, False, None), (4, True, None), (8, False, None), (4, True, 5.3), (3, True, None)):
            q = iq.int_quantizer('int%d' % bits, params(pcq_weights=True, pcq_act=False, bit_alloc_weight=baw,
                                                        bit_alloc_target_weight=target, measure_entropy=True,
                                                        logger=_Logger()))
            wq = q(w, 'm.weight', 'weight')
            d['k%d_w' % k], d['k%d_wq' % k] = w, wq
            d['k%d_codes' % k] = _spy['codes'].view(wshape).to(torch.int32)
            d['k%d_bits' % k], d['k%d_baw' % k] = np.int64(bits), np.int64(baw)
            d['k%d_target' % k] = np.float64(-1 if target is None else target)
            k += 1
    d['n_cases'] = np.int64(k)
    save('weights', d)


if __name__ == '__main__':
    torch.set_num_threads(1)
    gen_core()
    gen_act()
    gen_bit_alloc()
    gen_tables()
    gen_midtread()
    gen_collect()
    gen_weights()
