"""The oracle (oracle/quant_oracle.py) against the vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only.  Everything here is bit-exact: the oracle runs the same
aten ops in the same order as the reference on the same build of torch."""
import numpy as np
import pytest
import torch

from conftest import bits_equal
from oracle import quant_oracle as O

ACT_KW = {
    'cfg2_int4': dict(), 'cfg2_int4_half': dict(), 'cfg2_int8': dict(),
    'cfg2_int4_baa': dict(bit_alloc_act=True), 'cfg2_int4_baa_half': dict(bit_alloc_act=True),
    'cfg3_laplace': dict(clip='laplace'), 'cfg3_laplace_half': dict(clip='laplace'),
    'cfg3_laplace_baa': dict(clip='laplace', bit_alloc_act=True),
    'cfg3_laplace_baa_half': dict(clip='laplace', bit_alloc_act=True),
    'cfg3_laplace_baa_bap': dict(clip='laplace', bit_alloc_act=True, bit_alloc_prior='laplace'),
    'cfg3_laplace_baa_ceil': dict(clip='laplace', bit_alloc_act=True, bit_alloc_round=False),
    'cfg3_laplace_baa_t53': dict(clip='laplace', bit_alloc_act=True, bit_alloc_target=5.3),
    'cfg3_gaus': dict(clip='gaus'), 'cfg3_gaus_half': dict(clip='gaus'), 'cfg3_2std': dict(clip='2std'),
    'cfg3_laplace_int3_baa': dict(clip='laplace', bit_alloc_act=True),
    'cfg3_laplace_int2': dict(clip='laplace'),
}


def test_core_qdq(golden):
    g = golden('core_qdq')
    for i in range(int(g.np('n_cases'))):
        p = 'c%d_' % i
        ba = g.t(p + 'bit_alloc') if (p + 'bit_alloc') in g else None
        y, codes, scale, zp, qmax = O.qdq_core(g.t(p + 't'), g.t(p + 'delta'), g.t(p + 'offset'),
                                               num_bits=int(g.np(p + 'bits')), bit_alloc=ba, return_parts=True)
        assert bits_equal(y, g.np(p + 'y')), i
        assert np.array_equal(codes.int().numpy(), g.np(p + 'codes')), i
        assert bits_equal(O.shannon_entropy(codes.int()), g.np(p + 'entropy'))
    y = O.qdq_core(g.t('pt_x'), g.t('pt_delta'), g.t('pt_offset'), num_bits=8)
    assert bits_equal(y, g.np('pt_y'))


def test_act_per_channel_end_to_end(golden):
    g = golden('act_pc')
    for key in g.np('names'):
        key = str(key)
        name, si = key.rsplit('_s', 1)
        x = g.t('x' + si)
        kw = dict(ACT_KW[name])
        clip = kw.pop('clip', None)
        half = bool(g.np(key + '_half'))
        bits = int(g.np(key + '_bits'))
        if clip is None:
            y, parts = O.act_per_channel_qdq(x, bits, half_range=half, return_parts=True, **kw)
        else:
            y, parts = O.act_clipping_qdq(x, bits, clip_type=clip, half_range=half, return_parts=True, **kw)
            assert bits_equal(parts['alpha'], g.np(key + '_alpha')), key
            assert bits_equal(parts['range'], g.np(key + '_range')), key
        assert bits_equal(y, g.np(key + '_y')), key
        assert np.array_equal(parts['codes'].int().numpy(), g.np(key + '_codes')), key
        if (key + '_bit_alloc') in g:
            assert bits_equal(parts['bit_alloc'], g.np(key + '_bit_alloc')), key


def test_stats(golden):
    g = golden('act_pc')
    for si in range(5):
        x = g.t('x%d' % si)
        st = O.act_stats_perchannel(x, ['min', 'max', 'b', 'std', 'mean'])
        for s in st:
            assert bits_equal(st[s], g.np('s%d_stat_%s' % (si, s)))
        m = O.act_stats_perchannel(x, ['mean'], avg_over_batch=True)['mean']
        assert bits_equal(m, g.np('s%d_stat_mean_avgbatch' % si))


def test_bit_alloc(golden):
    g = golden('bit_alloc')
    for k in range(int(g.np('n_cases'))):
        std, target, rnd = g.t('k%d_std' % k), float(g.np('k%d_target' % k)), bool(g.np('k%d_round' % k))
        if target == int(target):
            target = int(target)
        assert bits_equal(O.bits_alloc_fixed_target(std, target, rnd), g.np('k%d_bits' % k)), k
        assert bits_equal(O.bits_alloc(std, target, rnd), g.np('k%d_bits_single' % k)), k


def test_tables(golden):
    g = golden('tables')
    for nm, tab in (('alpha_gaus', O.ALPHA_GAUS), ('alpha_gaus_positive', O.ALPHA_GAUS_POS),
                    ('alpha_laplace', O.ALPHA_LAPLACE), ('alpha_laplace_positive', O.ALPHA_LAPLACE_POS)):
        assert sorted(tab) == list(g.np(nm + '_keys'))
        assert [tab[k] for k in sorted(tab)] == list(g.np(nm + '_vals'))
    assert g.np('omega_table').shape == (101,) and g.np('alpha_table').shape == (101,)


def test_midtread(golden):
    g, tab = golden('midtread'), golden('tables')
    ot, at = tab.np('omega_table'), tab.np('alpha_table')
    for key in g.np('names'):
        key = str(key)
        name, si = key.rsplit('_s', 1)
        x = g.t('x' + si)
        target = float(g.np(key + '_target'))
        if target == int(target):
            target = int(target)
        half = bool(g.np(key + '_half'))
        y, ent = O.mid_tread_act_per_channel(x, target, half_range=half, omega_table=ot, alpha_table=at,
                                             want_entropy=True)
        assert bits_equal(y, g.np(key + '_y')), key
        assert bits_equal(ent, g.np(key + '_entropy')), key
    for wi in range(2):
        w = g.t('w%d' % wi)
        y, ent = O.mid_tread_weights_per_channel(w, 4, want_entropy=True)
        assert bits_equal(y, g.np('w%d_y' % wi))
        assert bits_equal(ent, g.np('w%d_entropy' % wi))


def test_collect_stats(golden):
    g = golden('collect')
    for bi, batch_avg in enumerate((False, True)):
        for k in range(3):
            st = O.collect_stats_perchannel(g.t('b%d_x%d' % (bi, k)), batch_avg=batch_avg)
            for s in O.COLLECT_STATS:
                assert bits_equal(st[s], g.np('b%d_%s' % (bi, s))[k]), (bi, k, s)
        assert list(g.np('b%d_ids' % bi)) == ['conv0_activation']      # FC and 1x1 inputs are skipped
    assert O.collect_stats_perchannel(torch.zeros(4, 10)) is None
    assert O.collect_stats_perchannel(torch.zeros(4, 6, 1, 1)) is None


def test_collect_stats_at_a_single_launch_shape(golden):
    """round 6: the reference's collection on [16,3,28,28] (tests/golden/make_golden_collect_flat.py), a shape whose statistics
    take ONE launch on the device (k_stats_flat) - the oracle restates it bit for bit here too."""
    g = golden('collect_flat')
    for bi, batch_avg in enumerate((False, True)):
        for k in range(2):
            st = O.collect_stats_perchannel(g.t('x%d' % k), batch_avg=batch_avg)
            for s in O.COLLECT_STATS:
                assert bits_equal(st[s], g.np('b%d_%s' % (bi, s))[k]), (bi, k, s)


def test_weights(golden):
    g = golden('weights')
    for k in range(int(g.np('n_cases'))):
        w = g.t('k%d_w' % k)
        target = float(g.np('k%d_target' % k))
        y, parts = O.weights_per_channel_qdq(w, int(g.np('k%d_bits' % k)), bool(g.np('k%d_baw' % k)),
                                             None if target < 0 else target, True, return_parts=True)
        assert bits_equal(y, g.np('k%d_wq' % k)), k
        assert np.array_equal(parts['codes'].int().numpy(), g.np('k%d_codes' % k)), k


def test_float2gemmlowp_restatement_vs_torch_path():
    """Row a13 has no runnable reference here (CUDA only).  Second pin: the
    reference's own CPU-runnable per-tensor route (qdq_core with 0-dim params, golden-pinned
    above) computes the same thing except for rounding of exact .5 ties and the scale floor."""
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(3, 5, 11, 13, generator=gen) * 1.7 - 0.2
    mn, mx = float(x.min()), float(x.max())
    a = O.float2gemmlowp(x, mx - mn, mn, 8, False, True)
    b = O.qdq_core(x, torch.tensor(mx - mn), torch.tensor(mn), num_bits=8)
    assert torch.equal(a, b)
    # range <= 0 hands back the input object itself (gemmlowp.cu:31-32)
    assert O.float2gemmlowp(x, 0.0, 0.0, 8, False, True) is x
    # ties: roundf goes away from zero, torch.round to even
    t = torch.tensor([0.5, 1.5, 2.5, 3.5, -0.5])
    assert O.float2gemmlowp(t, 255., 0., 8, False, False).tolist() == [1., 2., 3., 4., 0.]
    assert O.qdq_core(t, torch.tensor(255.), torch.tensor(0.), num_bits=8).tolist() == [0., 2., 2., 4., 0.]


@pytest.mark.parametrize('name', ['relu_first', 'full_range', 'one_channel_dead'])
def test_act_bias_correction(golden, name):
    """Row a12 against the reference's Conv2dWithId.forward (tests/golden/make_golden_bca.py)."""
    g = golden('bca')
    got = O.act_bias_correction(g.t(name + '/out'), g.t(name + '/out_q').clone(), bool(g.np(name + '/relu_first')))
    assert bits_equal(got.numpy(), g.np(name + '/corrected'))


def test_nan_golden(golden):
    """nan.npz: NaN / inf activations through the reference's config-2 chain; NaN positions and every other bit
    (the sign / payload bits of a NaN are not semantics: CPU torch itself varies them inside one tensor)."""
    g = golden('nan')
    for i in range(int(g.np('n_cases'))):
        y = O.act_per_channel_qdq(g.t('c%d_x' % i), 4, half_range=bool(g.np('c%d_half' % i))).numpy()
        ref = g.np('c%d_y' % i)
        na, nb = np.isnan(y), np.isnan(ref)
        assert np.array_equal(na, nb) and np.array_equal(y[~na].view(np.uint32), ref[~nb].view(np.uint32)), i
        assert nb[:, 2].all() and not nb[:, 3:].any()


def _mix_inputs(g):
    cols = [str(c) for c in g.np('summary_columns')]
    vals = g.np('summary_values')
    stats = {k: vals[:, cols.index('mean_%s' % k)] for k in ('min', 'max', 'mean', 'b', 'std')}
    mse = {k: g.np('mse_%s' % k) for k in ('laplace', 'gaus', 'lowp')}
    return stats, mse


def test_clipping_mix(golden):
    """clip_type == 'mix' (iq.py:310-323) on the `-sm use` route: the restatement against the reference run on a
    statistics file whose mse_* columns select every branch (tests/golden/make_golden_mix.py)."""
    g = golden('mix')
    stats, mse = _mix_inputs(g)
    picks = set()
    for nm in g.np('names'):
        nm = str(nm)
        half, baa = nm[8] == '1', nm[-1] == '1'
        y, parts = O.act_clipping_mix_qdq(g.t('x0'), 4, stats, mse, half_range=half, bit_alloc_act=baa, return_parts=True)
        assert bits_equal(y, g.np(nm + '_y')), nm
        assert np.array_equal(parts['codes'].numpy().astype(np.int32), g.np(nm + '_codes')), nm
    with np.errstate(invalid='ignore'):
        gaus = mse['gaus'] < mse['laplace']
        lowp = mse['lowp'] < mse['gaus']
    picks = {('lowp' if lo else 'gaus' if ga else 'laplace') for ga, lo in zip(gaus, lowp)}
    assert picks == {'laplace', 'gaus', 'lowp'}                  # the fixture exercises all three
