"""Parity of the HIP path (through the C ABI, cnn_quantization_amd.ops) with the oracle and with
the golden vectors the reference produced.  Needs an MI355X: `pytest -m gpu`.

Tiers (SURVEY.md section 7 "hard parts"):
  * Q/DQ given identical per-channel parameters: bit-exact, codes and dequantized floats;
  * min/max statistics: exact -> config 2 (dynamic min/max) is bit-exact end to end;
  * mean / std / b: sums are accumulated in fp64 in a different order than torch's fp32
    reductions, so they agree to ~1e-6 relative, not bitwise; parameters derived from them may
    differ in the last bit and then move an element that sits exactly on a rounding boundary by
    one code.  End-to-end ACIQ tests therefore bound the fraction of differing codes (<= 2e-4, each
    by one step) and separately prove bit-exactness with the oracle's own parameters.
"""
import numpy as np
import pytest
import torch

from conftest import bits_equal
from oracle import quant_oracle as O

pytestmark = pytest.mark.gpu

RTOL_STAT = 2e-6


@pytest.fixture(scope='module')
def ops():
    from cnn_quantization_amd import ops as _ops
    return _ops


def dev(t):
    return t.to('cuda')


def qp_table(scale, zp, qmax, C):
    qp = torch.empty(3, C)
    qp[0] = scale.reshape(-1).expand(C) if scale.numel() == 1 else scale
    qp[1] = zp.reshape(-1).expand(C) if zp.numel() == 1 else zp
    qp[2] = qmax if not isinstance(qmax, torch.Tensor) else qmax.reshape(-1)
    return qp


# --------------------------------------------------------------------------- a4: core Q/DQ
def test_core_qdq_golden_bit_exact(ops, golden):
    g = golden('core_qdq')
    for i in range(int(g.np('n_cases'))):
        p = 'c%d_' % i
        t = g.t(p + 't')
        C, M = t.shape
        ba = g.t(p + 'bit_alloc') if (p + 'bit_alloc') in g else None
        _, _, scale, zp, qmax = O.qdq_core(t, g.t(p + 'delta'), g.t(p + 'offset'), num_bits=int(g.np(p + 'bits')),
                                           bit_alloc=ba, return_parts=True)
        qp = dev(qp_table(scale, zp, qmax, C))
        for layout in ('1CM', 'weights'):
            x = dev(t.view(1, C, M) if layout == '1CM' else t)
            y, codes = ops.pc_qdq(x, 1, C, M, qp, want_codes=True)
            assert bits_equal(y.cpu().reshape(C, M), g.np(p + 'y')), (i, layout)
            assert np.array_equal(codes.cpu().reshape(C, M).numpy().astype(np.int32), g.np(p + 'codes')), (i, layout)


@pytest.mark.parametrize('shape', [(3, 5, 7, 7), (2, 8, 14, 14), (4, 3, 5, 9), (2, 6, 1, 3), (3, 4, 33, 31),
                                   (2, 2, 40, 36), (1, 1100, 2, 2), (5, 7, 1, 1029), (2, 3, 64, 80)])
@pytest.mark.parametrize('misalign', [0, 1])
def test_qdq_shapes_vs_oracle(ops, shape, misalign):
    """Every load shape (VEC4, VEC4-straddle, VEC1), both block modes, odd sizes, and a base
    pointer that is not 16-byte aligned."""
    gen = torch.Generator().manual_seed(hash(shape) % 1000 + misalign)
    N, C, H, W = shape
    x = torch.randn(shape, generator=gen) * 3
    bits = torch.randint(0, 9, (C,), generator=gen).float()
    mn = x.transpose(0, 1).reshape(C, -1).min(-1)[0]
    mx = x.transpose(0, 1).reshape(C, -1).max(-1)[0]
    t = x.transpose(0, 1).contiguous().view(C, -1)
    y_ref, codes_ref, scale, zp, qmax = O.qdq_core(t, (mx - mn) * 0.7, mn * 0.8, bit_alloc=bits, return_parts=True)
    y_ref = y_ref.view(C, N, H, W).transpose(0, 1).contiguous()
    codes_ref = codes_ref.view(C, N, H, W).transpose(0, 1).contiguous()
    qp = dev(qp_table(scale, zp, qmax, C))
    if misalign:
        buf = torch.empty(x.numel() + 1, device='cuda')
        xd = buf[1:].view(shape)
        xd.copy_(x)
        out = torch.empty(x.numel() + 1, device='cuda')[1:].view(shape)
    else:
        xd, out = dev(x), None
    y, codes = ops.pc_qdq(xd, N, C, H * W, qp, want_codes=True, out=out)
    assert bits_equal(y.cpu(), y_ref)
    assert torch.equal(codes.cpu().float(), codes_ref)


# --------------------------------------------------------------------------- a2: statistics
def test_stats_vs_golden(ops, golden):
    g = golden('act_pc')
    from cnn_quantization_amd._lib import STAT_MIN, STAT_MAX, STAT_MEAN, STAT_STD, STAT_B
    for si in range(5):
        x = g.t('x%d' % si)
        N, C = x.shape[:2]
        st, mom = ops.pc_stats(dev(x), N, C, x.shape[2] * x.shape[3], need_b=True)
        st = st.cpu()
        assert bits_equal(st[STAT_MIN], g.np('s%d_stat_min' % si))
        assert bits_equal(st[STAT_MAX], g.np('s%d_stat_max' % si))
        np.testing.assert_allclose(st[STAT_MEAN], g.np('s%d_stat_mean' % si), rtol=RTOL_STAT, atol=1e-7)
        np.testing.assert_allclose(st[STAT_MEAN], g.np('s%d_stat_mean_avgbatch' % si), rtol=RTOL_STAT, atol=1e-7)
        np.testing.assert_allclose(st[STAT_STD], g.np('s%d_stat_std' % si), rtol=RTOL_STAT)
        np.testing.assert_allclose(st[STAT_B], g.np('s%d_stat_b' % si), rtol=RTOL_STAT)
        assert float(mom[4].cpu()[0]) == N * x.shape[2] * x.shape[3]


@pytest.mark.parametrize('shape', [(6, 5, 28, 28), (3, 70, 7, 7), (4, 9, 3, 5), (2, 3, 70, 66), (7, 1500, 1, 2)])
def test_stats_collect_set_vs_oracle(ops, shape):
    """All seven statistics of smpc.py:45-79 against the oracle."""
    from cnn_quantization_amd import _lib as L
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(shape, generator=gen) * torch.rand(1, shape[1], 1, 1, generator=gen) * 4 + \
        torch.randn(1, shape[1], 1, 1, generator=gen)
    ref = O.collect_stats_perchannel(x)
    N, C = shape[:2]
    st, _ = ops.pc_stats(dev(x), N, C, shape[2] * shape[3], need_b=True, need_kurt=True, need_relu=True)
    st = st.cpu()
    assert bits_equal(st[L.STAT_MIN], ref['min'])
    assert bits_equal(st[L.STAT_MAX], ref['max'])
    for row, name, tol in ((L.STAT_MEAN, 'mean', 5e-6), (L.STAT_STD, 'std', 5e-6), (L.STAT_B, 'b', 5e-6),
                           (L.STAT_STD_POS, 'std_pos', 5e-6)):
        np.testing.assert_allclose(st[row], ref[name], rtol=tol, atol=2e-6, err_msg=name)
    # kurtosis = mean(((x-mean)/std)^4) - 3 amplifies the last-ulp uncertainty of the fp32 mean by ~4*|mean|/std
    np.testing.assert_allclose(st[L.STAT_KURT], ref['kurtosis'], rtol=1e-3, atol=5e-4, err_msg='kurtosis')


# --------------------------------------------------------------------------- a8: bit allocation
def test_bit_alloc_golden(ops, golden):
    from cnn_quantization_amd import _lib as L
    g = golden('bit_alloc')
    for k in range(int(g.np('n_cases'))):
        std = g.t('k%d_std' % k)
        C = std.numel()
        stats = torch.zeros(L.NSTAT, C)
        stats[L.STAT_STD] = std
        stats[L.STAT_MAX] = 1.
        _, diag = ops.pc_params(dev(stats), 4, clip='no', bit_alloc=True, target=float(g.np('k%d_target' % k)),
                                round_mode=bool(g.np('k%d_round' % k)))
        bits = diag[L.DIAG_BITS].cpu().numpy()
        ref = g.np('k%d_bits' % k)
        # a channel exactly on a rounding boundary of log2 may legitimately flip (powf/log2f last ulp);
        # on these seeds none does
        assert np.array_equal(bits, ref), (k, int((bits != ref).sum()))


# --------------------------------------------------------------------------- a5 / a6 end to end
def _run_act(ops, x, bits, half, kw):
    clip = kw.get('clip', 'no')
    return ops.act_qdq_per_channel(dev(x), bits, positive=half, clip=clip, bit_alloc=kw.get('bit_alloc_act', False),
                                   prior_is_b=kw.get('bit_alloc_prior', 'gaus') == 'laplace',
                                   target=kw.get('bit_alloc_target'), round_mode=kw.get('bit_alloc_round', True),
                                   want_codes=True, want_parts=True)


def test_act_cfg2_golden_bit_exact(ops, golden):
    """Config 2 (-pcq_a, dynamic min/max, no clipping): bit-exact end to end."""
    from test_oracle_golden import ACT_KW
    g = golden('act_pc')
    n = 0
    for key in g.np('names'):
        key = str(key)
        name, si = key.rsplit('_s', 1)
        if not name.startswith('cfg2') or 'baa' in name:
            continue
        y, codes, _ = _run_act(ops, g.t('x' + si), int(g.np(key + '_bits')), bool(g.np(key + '_half')), ACT_KW[name])
        assert bits_equal(y.cpu(), g.np(key + '_y')), key
        assert np.array_equal(codes.cpu().numpy().astype(np.int32), g.np(key + '_codes')), key
        n += 1
    assert n == 15


def test_act_cfg3_golden(ops, golden):
    """ACIQ / bit-allocation configs: parameters to ~1e-6, bit allocation identical, codes equal
    except (rarely) elements on a rounding boundary, by one step."""
    from cnn_quantization_amd import _lib as L
    from test_oracle_golden import ACT_KW
    g = golden('act_pc')
    total = diff = 0
    for key in g.np('names'):
        key = str(key)
        name, si = key.rsplit('_s', 1)
        if name.startswith('cfg2') and 'baa' not in name:
            continue
        kw = dict(ACT_KW[name])
        y, codes, parts = _run_act(ops, g.t('x' + si), int(g.np(key + '_bits')), bool(g.np(key + '_half')), kw)
        diag = parts['diag'].cpu()
        if (key + '_bit_alloc') in g:
            assert np.array_equal(diag[L.DIAG_BITS].numpy(), g.np(key + '_bit_alloc')), key
        if (key + '_alpha') in g:
            np.testing.assert_allclose(diag[L.DIAG_ALPHA], g.np(key + '_alpha'), rtol=RTOL_STAT, err_msg=key)
            np.testing.assert_allclose(diag[L.DIAG_DELTA], g.np(key + '_range'), rtol=2 * RTOL_STAT, err_msg=key)
            np.testing.assert_allclose(diag[L.DIAG_OFFSET], g.np(key + '_offset'), rtol=2 * RTOL_STAT, atol=1e-6,
                                       err_msg=key)
        c = codes.cpu().numpy().astype(np.int32)
        ref = g.np(key + '_codes')
        d = np.abs(c - ref)
        assert d.max() <= 1, key
        total += d.size
        diff += int((d != 0).sum())
        np.testing.assert_allclose(y.cpu().numpy(), g.np(key + '_y'), rtol=1e-5,
                                   atol=float(parts['qp'][0].max()) * 1.001, err_msg=key)
    assert diff <= 2e-4 * total, (diff, total)


def test_act_cfg3_bit_exact_given_oracle_stats(ops, golden):
    """Same configs with the REFERENCE's statistics injected: parameters, codes and floats must
    then be bit-exact (proves the parameter kernel and Q/DQ; isolates the summation-order tier)."""
    from cnn_quantization_amd import _lib as L
    from test_oracle_golden import ACT_KW
    g = golden('act_pc')
    n = 0
    for key in g.np('names'):
        key = str(key)
        name, si = key.rsplit('_s', 1)
        x = g.t('x' + si)
        C = x.shape[1]
        st = torch.zeros(L.NSTAT, C)
        st[L.STAT_MIN], st[L.STAT_MAX] = g.t('s%s_stat_min' % si), g.t('s%s_stat_max' % si)
        st[L.STAT_MEAN], st[L.STAT_STD] = g.t('s%s_stat_mean_avgbatch' % si), g.t('s%s_stat_std' % si)
        st[L.STAT_B] = g.t('s%s_stat_b' % si)
        kw = dict(ACT_KW[name])
        y, codes = ops.act_qdq_per_channel(
            dev(x), int(g.np(key + '_bits')), positive=bool(g.np(key + '_half')), clip=kw.get('clip', 'no'),
            bit_alloc=kw.get('bit_alloc_act', False), prior_is_b=kw.get('bit_alloc_prior', 'gaus') == 'laplace',
            target=kw.get('bit_alloc_target'), round_mode=kw.get('bit_alloc_round', True), want_codes=True,
            stats=dev(st))
        assert np.array_equal(codes.cpu().numpy().astype(np.int32), g.np(key + '_codes')), key
        assert bits_equal(y.cpu(), g.np(key + '_y')), key
        n += 1
    assert n == len(g.np('names'))


# --------------------------------------------------------------------------- a10: weights
def test_weights_golden(ops, golden):
    g = golden('weights')
    for k in range(int(g.np('n_cases'))):
        w = g.t('k%d_w' % k)
        target = float(g.np('k%d_target' % k))
        y, codes = ops.act_qdq_per_channel(dev(w), int(g.np('k%d_bits' % k)), bit_alloc=bool(g.np('k%d_baw' % k)),
                                           target=None if target < 0 else target, per_channel_dim=0,
                                           want_codes=True)
        ref = g.np('k%d_codes' % k)
        c = codes.cpu().numpy().astype(np.int32)
        if not bool(g.np('k%d_baw' % k)):
            assert np.array_equal(c, ref), k
            assert bits_equal(y.cpu(), g.np('k%d_wq' % k)), k
        else:   # bit allocation from std: tolerance tier
            assert (c != ref).mean() <= 1e-3, k


# --------------------------------------------------------------------------- a13: per-tensor kernel
@pytest.mark.parametrize('n', [1, 3, 4, 1023, 4096 + 5, 70001])
@pytest.mark.parametrize('etz', [True, False])
def test_float2gemmlowp_vs_oracle(ops, n, etz):
    gen = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=gen) * 2 - 0.3
    x[0] = 0.
    mn, mx = float(x.min()), float(x.max())
    for bits in (8, 4):
        ref = O.float2gemmlowp(x, mx - mn, mn, bits, False, etz)
        ptp = ops.pt_setup('cuda', bits, range_offset=(mx - mn, mn), enforce_true_zero=etz)
        y = ops.pt_qdq(dev(x), ptp)
        assert bits_equal(y.cpu(), ref), (n, etz, bits)
    # exact .5 ties: roundf semantics (half away from zero)
    t = torch.tensor([0.5, 1.5, 2.5, 3.5, -0.5, 254.5, 300.])
    ptp = ops.pt_setup('cuda', 8, range_offset=(255., 0.), enforce_true_zero=False)
    assert ops.pt_qdq(dev(t), ptp).cpu().tolist() == O.float2gemmlowp(t, 255., 0., 8, False, False).tolist()


@pytest.mark.parametrize('tag,shape', [('activation_pooling', (6, 4, 9, 9)), ('activation_classifier', (5, 1000)),
                                       ('activation_linear', (8, 37))])
@pytest.mark.parametrize('half', [False, True])
def test_minmax_per_tensor_vs_oracle(ops, tag, shape, half):
    """iq.py:361-379: dynamic per-tensor min/max on the device (per-sample mean for activation
    tags) feeding the GEMMLOWP kernel."""
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(shape, generator=gen) * 1.5 + 0.2
    avg = 'activation' in tag and 'classifier' not in tag
    ref = O.gemmlowp_minmax_qdq(x, 8, tag=tag, half_range=half)
    y = ops.minmax_qdq_per_tensor(dev(x), 8, avg_over_batch=avg, zero_min=half).cpu()
    if avg:   # batch mean of per-sample extrema: fp32 summation-order tier
        step = float((x.max() - x.min()) / 255)
        assert (y - ref).abs().max() <= step * 1.01
        assert ((y - ref).abs() > 1e-5).float().mean() < 2e-3
    else:
        assert bits_equal(y, ref)


# --------------------------------------------------------------------------- size-independent properties
def test_full_size_properties(ops):
    """A ResNet-50 b512-sized layer slice ([64, 256, 56, 56] = 51 M elements): properties that do
    not need the oracle at this size."""
    from cnn_quantization_amd import _lib as L
    torch.manual_seed(12345)
    x = torch.empty(64, 256, 56, 56, device='cuda').normal_()
    x.mul_(torch.rand(1, 256, 1, 1, device='cuda') * 3 + 0.1)
    y, codes, parts = ops.act_qdq_per_channel(x, 4, want_codes=True, want_parts=True)
    qp, st = parts['qp'], parts['stats']
    # statistics equal torch's own reductions (exact for min/max)
    assert torch.equal(st[L.STAT_MIN], x.amin(dim=(0, 2, 3)))
    assert torch.equal(st[L.STAT_MAX], x.amax(dim=(0, 2, 3)))
    assert int(codes.max()) <= 15
    # each channel uses at most 2^4 levels and the dequantized values are code*scale - zp*scale
    yc = (codes.float() - qp[1].view(1, -1, 1, 1)) * qp[0].view(1, -1, 1, 1)
    assert torch.equal(yc, y)
    # idempotence: quantizing the dequantized tensor with the same parameters changes nothing
    y2 = ops.pc_qdq(y, 64, 256, 56 * 56, qp)
    assert torch.equal(y2, y)
    # error bound: |x - y| <= scale/2 inside the range
    err = (x - y).abs() - 0.5001 * qp[0].view(1, -1, 1, 1)
    assert float(err.max()) <= 0.


# --------------------------------------------------------------------------- a14 / a15: mid-tread + entropy
def test_midtread_golden(ops, golden):
    """Config 5 (-mtq -me): bin allocation, table interpolation, clamp around the quantized mean and
    the entropy of the codes.  omega (rounded bin counts) must match exactly; Delta / clamp bounds
    derive from mean, b, std (fp64-sum tier), so codes may differ on elements at a rounding boundary."""
    from cnn_quantization_amd import _lib as L
    g = golden('midtread')
    total = diff = 0
    for key in g.np('names'):
        key = str(key)
        name, si = key.rsplit('_s', 1)
        x = g.t('x' + si)
        target = float(g.np(key + '_target'))
        half = bool(g.np(key + '_half'))
        y, ent, codes, parts = ops.mid_tread_qdq(dev(x), target, clip=True, sym=not half, want_entropy=True,
                                                 want_codes=True, want_parts=True)
        mt = parts['mt'].cpu()
        assert np.array_equal(mt[L.MT_OMEGA].numpy(), g.np(key + '_omega')), key
        np.testing.assert_allclose(mt[L.MT_ALPHA].numpy(), g.np(key + '_alpha_mult').astype(np.float32), rtol=1e-6,
                                   err_msg=key)
        ref_codes = g.np(key + '_codes')
        c = codes.cpu().numpy()
        d = np.abs(c - ref_codes)
        assert d.max() <= 1.0001, key
        total += d.size
        diff += int((d > 1e-4).sum())
        step = float(mt[L.MT_DELTA][mt[L.MT_OMEGA] > 0].max())
        np.testing.assert_allclose(y.cpu().numpy(), g.np(key + '_y'), rtol=1e-5, atol=step * 1.001, err_msg=key)
        assert abs(float(ent) - float(g.np(key + '_entropy'))) < 2e-3, (key, float(ent), float(g.np(key + '_entropy')))
    assert diff <= 5e-4 * total, (diff, total)
    for wi in range(2):                      # weights: no clipping, symmetric, range = max - min (exact statistics)
        w = g.t('w%d' % wi)
        y, ent, codes = ops.mid_tread_qdq(dev(w), 4, clip=False, sym=True, per_channel_dim=0, group=False,
                                          want_entropy=True, want_codes=True)
        ref = g.np('w%d_codes' % wi)
        assert (np.abs(codes.cpu().numpy() - ref) > 0).mean() < 2e-3
        assert abs(float(ent) - float(g.np('w%d_entropy' % wi))) < 5e-3


def test_midtread_bit_exact_given_oracle_stats(ops, golden):
    """With the oracle's statistics injected the mid-tread parameters, codes and floats are bit-exact."""
    from cnn_quantization_amd import _lib as L
    g, tab = golden('midtread'), golden('tables')
    lib = L.load()
    for key in g.np('names'):
        key = str(key)
        name, si = key.rsplit('_s', 1)
        x = g.t('x' + si)
        N, C = x.shape[:2]
        HW = x.shape[2] * x.shape[3]
        target, half = float(g.np(key + '_target')), bool(g.np(key + '_half'))
        st = O.act_stats_perchannel(x, ['min', 'max', 'mean', 'b', 'std'])
        table = torch.zeros(L.NSTAT, C)
        for row, nm in ((L.STAT_MIN, 'min'), (L.STAT_MAX, 'max'), (L.STAT_MEAN, 'mean'), (L.STAT_B, 'b'),
                        (L.STAT_STD, 'std')):
            table[row] = st[nm]
        table = dev(table)
        tabs = ops._midtread_tables(table.device)
        mt = torch.empty((L.NMT, C), dtype=torch.float32, device='cuda')
        L.check(lib.cnnq_pc_midtread_params(ops._ptr(table), C, target, 1, int(not half), ops._ptr(tabs), 101,
                                            ops._ptr(mt), ops._stream(table)), 'params')
        xd = dev(x)
        y, codes = torch.empty_like(xd), torch.empty_like(xd)
        L.check(lib.cnnq_pc_midtread_qdq(ops._ptr(xd), ops._ptr(y), N, C, HW, ops._ptr(mt), 1, ops._ptr(codes), None,
                                         ops._stream(xd)), 'qdq')
        assert bits_equal(codes.cpu(), g.np(key + '_codes')), key
        assert bits_equal(y.cpu(), g.np(key + '_y')), key


# --------------------------------------------------------------------------- a11 / a12: corrections
@pytest.mark.parametrize('wshape', [(16, 8, 3, 3), (10, 64), (24, 3, 7, 7), (300, 5, 1, 1)])
@pytest.mark.parametrize('vc,bc', [(False, True), (True, False), (True, True)])
def test_weight_correction_vs_oracle(ops, wshape, vc, bc):
    """iqm.py:374-391.  Per-channel means / stds are fp64-sum tier, so agreement is ~1e-6 relative."""
    gen = torch.Generator().manual_seed(21)
    w = torch.randn(wshape, generator=gen) * 0.05 + 0.01
    wq = O.weights_per_channel_qdq(w, 4)
    ref = O.weight_correction(w, wq, vcorr=vc, bcorr=bc)
    out = ops.weight_correction(dev(w), dev(wq), vcorr=vc, bcorr=bc).cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=2e-5, atol=2e-7)
    # the corrected weights have the original per-channel mean (bcorr) / std (vcorr)
    if bc:
        np.testing.assert_allclose(out.view(wshape[0], -1).mean(-1), w.view(wshape[0], -1).mean(-1), atol=2e-6)
    if vc and not bc:
        np.testing.assert_allclose(out.view(wshape[0], -1).std(-1), w.view(wshape[0], -1).std(-1), rtol=1e-4)


@pytest.mark.parametrize('shape', [(4, 8, 7, 7), (3, 20, 14, 14), (2, 5, 33, 31)])
@pytest.mark.parametrize('relu_first', [False, True])
def test_act_bias_correction_vs_oracle(ops, shape, relu_first):
    """iqm.py:188-196."""
    gen = torch.Generator().manual_seed(22)
    x = torch.randn(shape, generator=gen) * 2 + 0.4
    xq = O.act_per_channel_qdq(x, 4, half_range=relu_first)
    ref = O.act_bias_correction(x, xq.clone(), relu_first)
    out = ops.act_bias_correction_(dev(x), dev(xq).clone(), relu_first).cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    assert torch.equal(out == 0, ref == 0)


@pytest.mark.parametrize('name', ['relu_first', 'full_range', 'one_channel_dead'])
def test_act_bias_correction_golden(ops, golden, name):
    """iqm.py:188-196 against tensors recorded from the reference's Conv2dWithId.forward (incl. a channel
    without any positive element: count 0 -> bias = sum / 1e-8).  Sums are fp64 here, fp32 in torch."""
    g = golden('bca')
    ref = g.np(name + '/corrected')
    out = ops.act_bias_correction_(dev(g.t(name + '/out')), dev(g.t(name + '/out_q')).clone(),
                                   bool(g.np(name + '/relu_first'))).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=2e-5, atol=1e-5)
    assert np.array_equal(out == 0, ref == 0)


@pytest.mark.parametrize('shape', [(4, 8, 7, 7), (3, 20, 14, 14), (2, 5, 33, 31), (6, 64, 56, 56), (5, 300, 7, 7)])
@pytest.mark.parametrize('relu_first', [False, True])
@pytest.mark.parametrize('clip', ['no', 'laplace'])
def test_fused_bias_correction_equals_two_step(ops, shape, relu_first, clip):
    """qdq_bias_corrected (statistics pass recomputing q + one fused quantize+correct pass) must give the
    floats of pc_qdq followed by act_bias_correction_ (same sums, same expression)."""
    gen = torch.Generator().manual_seed(5)
    x = dev(torch.randn(shape, generator=gen) * 1.7 + 0.3)
    N, C, HW = shape[0], shape[1], shape[2] * shape[3]
    stats, _ = ops.pc_stats(x, N, C, HW, need_b=True)
    qp, _ = ops.pc_params(stats, 4, relu_first, clip, clip != 'no')
    two = ops.act_bias_correction_(x, ops.pc_qdq(x, N, C, HW, qp), relu_first)
    one = ops.qdq_bias_corrected(x, N, C, HW, qp, relu_first)
    assert bits_equal(one.cpu().numpy(), two.cpu().numpy())
    via = ops.act_qdq_per_channel(x, 4, positive=relu_first, clip=clip, bit_alloc=clip != 'no', stats=stats,
                                  bcorr=relu_first)
    assert bits_equal(via.cpu().numpy(), two.cpu().numpy())


@pytest.mark.parametrize('shape', [(4, 8, 14, 14), (3, 20, 12, 12), (2, 5, 2, 2), (6, 3, 56, 56), (2, 600, 4, 4)])
@pytest.mark.parametrize('bits', [8, 5])
def test_u8_round_trip_equals_fused_qdq(ops, shape, bits):
    """One byte per code as the stored format: codes == the codes of the fused Q/DQ, dequantized == its floats."""
    gen = torch.Generator().manual_seed(3)
    x = dev(torch.randn(shape, generator=gen) * 2 + 0.2)
    N, C, HW = shape[0], shape[1], shape[2] * shape[3]
    stats, _ = ops.pc_stats(x, N, C, HW)
    qp, _ = ops.pc_params(stats, bits, False, 'no', False)
    y, codes = ops.pc_qdq(x, N, C, HW, qp, want_codes=True)
    stored = ops.quantize_u8(x, qp)
    assert torch.equal(stored, codes)
    assert torch.equal(ops.dequantize_u8(stored, qp), y)


@pytest.mark.parametrize('shape,c0,c1', [((4, 16, 14, 14), 0, 8), ((4, 16, 14, 14), 8, 16), ((3, 10, 7, 7), 4, 8),
                                         ((3, 10, 7, 7), 3, 10), ((2, 6, 5, 9), 1, 5), ((2, 64, 56, 56), 16, 48)])
@pytest.mark.parametrize('half', [False, True])
def test_channel_slice_in_place(ops, shape, c0, c1, half):
    """Config 2 on x[:, c0:c1] through the strided entry points (pointer offset + parent sample stride, no
    slice copy) == the same slice quantized as a contiguous tensor; the rest of `out` is untouched."""
    gen = torch.Generator().manual_seed(9)
    x = dev(torch.randn(shape, generator=gen) * 2 + 0.1)
    out = torch.full_like(x, -777.)
    ops.minmax_qdq_channel_slice(x, c0, c1, 4, positive=half, out=out)
    ref = ops.act_qdq_per_channel(x[:, c0:c1].contiguous(), 4, positive=half)
    assert torch.equal(out[:, c0:c1], ref)
    mask = torch.ones(shape[1], dtype=torch.bool)
    mask[c0:c1] = False
    assert bool((out[:, mask.cuda()] == -777.).all())


# --------------------------------------------------------------------------- edge cases, config 2 end to end
@pytest.mark.parametrize('shape', [(1, 3, 5, 5), (7, 1, 9, 9), (2, 4097, 1, 5), (3, 5, 1, 1), (1, 1, 1, 2), (2, 300, 7, 7),
                                   (9, 17, 13, 11), (1, 64, 112, 112)])
@pytest.mark.parametrize('half', [False, True])
def test_cfg2_edge_shapes_bit_exact(ops, shape, half):
    """Single sample, single channel, 1x1 / 1xW planes, more channels than a workgroup can own,
    odd everything: the dynamic min/max path stays bit-exact; degenerate channels (constant, all zero,
    all negative with half range) hit the scale floor exactly like the reference."""
    gen = torch.Generator().manual_seed(sum(shape) + half)
    x = torch.randn(shape, generator=gen) * 2.5 + 0.3
    C = shape[1]
    if C >= 3:
        x[:, 0] = 0.75            # constant channel: delta = 0 -> scale 1e-8
        x[:, 1] = 0.              # all zeros
        x[:, 2] = -x[:, 2].abs() - 0.1   # all negative (half range: max < 0)
    ref, parts = O.act_per_channel_qdq(x, 4, half_range=half, return_parts=True)
    y, codes = ops.act_qdq_per_channel(dev(x), 4, positive=half, want_codes=True)
    assert torch.equal(codes.cpu().float(), parts['codes'])
    assert bits_equal(y.cpu(), ref)


def test_noncontiguous_and_offset_inputs(ops):
    """Views with a storage offset (base pointer not 16-byte aligned) and non-contiguous inputs."""
    gen = torch.Generator().manual_seed(77)
    big = torch.randn(3, 10, 12, 13, generator=gen)
    for x in (big[:, 1:9], big.transpose(2, 3), big[1:]):
        ref = O.act_per_channel_qdq(x.contiguous(), 4)
        y = ops.act_qdq_per_channel(dev(big)[:, 1:9] if x.shape[1] == 8 else (dev(big).transpose(2, 3) if x.shape[2] == 13 else dev(big)[1:]), 4)
        assert bits_equal(y.cpu(), ref)
    flat = torch.randn(1 + 4 * 6 * 49, generator=gen)
    xo = flat[1:].view(4, 6, 7, 7)
    assert bits_equal(ops.act_qdq_per_channel(dev(flat)[1:].view(4, 6, 7, 7), 4).cpu(), O.act_per_channel_qdq(xo.contiguous(), 4))


# --------------------------------------------------------------------------- a6 per-tensor clipping branch
@pytest.mark.parametrize('shape', [(8, 37), (4, 6, 5, 5), (16, 1000)])
@pytest.mark.parametrize('clip', ['laplace', 'gaus', '2std'])
@pytest.mark.parametrize('half', [False, True])
def test_per_tensor_clipping_vs_oracle(ops, shape, clip, half):
    """iq.py:353-357: when -pcq_a does not apply (FC activations, activation_linear) ACIQ clipping uses
    scalar statistics of the whole tensor and delta = range itself.  Statistics tier: the scalar mean /
    b / std differ in the last bits, so codes may move by one step on boundary elements."""
    gen = torch.Generator().manual_seed(len(shape) * 7 + half)
    x = torch.randn(shape, generator=gen) * 1.3 + 0.2
    ref, parts = O.act_clipping_qdq(x, 4, clip_type=clip, half_range=half, pcq_a=False, return_parts=True)
    y, p = ops.act_qdq_per_channel(dev(x), 4, positive=half, clip=clip, whole_tensor=True, want_parts=True)
    from cnn_quantization_amd import _lib as L
    diag = p['diag'].cpu()
    np.testing.assert_allclose(float(diag[L.DIAG_DELTA][0]), float(parts['range']), rtol=3e-6)
    np.testing.assert_allclose(float(diag[L.DIAG_OFFSET][0]), float(parts['offset']), rtol=3e-6, atol=1e-7)
    step = float(p['qp'][0][0])
    d = (y.cpu() - ref).abs()
    assert float(d.max()) <= step * 1.01
    assert float((d > 1e-5).float().mean()) < 2e-3


# --------------------------------------------------------------------------- f3: packed int4 storage
@pytest.mark.parametrize('shape', [(4, 8, 14, 14), (3, 20, 12, 12), (2, 5, 2, 2), (6, 3, 56, 56), (2, 600, 4, 4)])
def test_pack4_round_trip_equals_fused_qdq(ops, shape):
    gen = torch.Generator().manual_seed(shape[1])
    x = dev(torch.randn(shape, generator=gen) * 2 + 0.1)
    y, codes, parts = ops.act_qdq_per_channel(x, 4, want_codes=True, want_parts=True)
    packed = ops.quantize_pack4(x, parts['qp'])
    assert packed.numel() * 2 == x.numel()
    lo, hi = packed & 15, packed >> 4
    c = codes.reshape(-1)
    assert torch.equal(lo, c[0::2]) and torch.equal(hi, c[1::2])
    assert torch.equal(ops.dequantize_pack4(packed, shape, parts['qp']), y)
