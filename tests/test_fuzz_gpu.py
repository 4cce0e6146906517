"""Property-based parity (hypothesis, derandomised so every run draws the same cases): random tensor
geometries - odd H*W (float4 loads straddling channels), single samples, more channels than a
workgroup owns, unaligned base pointers - through the config-2 pipeline and the packed / corrected /
per-tensor variants, against the oracle.  Integer work: bit-exact."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from conftest import bits_equal
from oracle import quant_oracle as O

pytestmark = pytest.mark.gpu
CFG = dict(max_examples=60, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))

shapes = st.tuples(st.integers(1, 6), st.integers(1, 40), st.integers(1, 19), st.integers(1, 19)).filter(
    lambda s: s[2] * s[3] > 1)


def make(shape, seed, offset):
    g = torch.Generator().manual_seed(seed)
    n = int(np.prod(shape))
    base = torch.empty(n + 3).cuda()
    x = torch.randn(shape, generator=g) * (0.2 + 3 * torch.rand(1, shape[1], 1, 1, generator=g)) \
        + torch.randn(1, shape[1], 1, 1, generator=g)
    view = base[offset:offset + n].view(shape)              # offset != 0: base pointer not 16-byte aligned
    view.copy_(x)
    return x, view


@settings(**CFG)
@given(shape=shapes, seed=st.integers(0, 2 ** 16), bits=st.sampled_from([2, 3, 4, 8]), half=st.booleans(),
       offset=st.integers(0, 3))
def test_cfg2_random_geometry_bit_exact(shape, seed, bits, half, offset):
    from cnn_quantization_amd import ops
    x, xd = make(shape, seed, offset)
    ref = O.act_per_channel_qdq(x, bits, half_range=half)
    y, codes = ops.act_qdq_per_channel(xd, bits, positive=half, want_codes=True)
    assert bits_equal(y.cpu().numpy(), ref.numpy())
    assert int(codes.max()) <= 2 ** bits - 1
    again = ops.act_qdq_per_channel(xd, bits, positive=half)   # deterministic
    assert torch.equal(again, y)


@settings(**CFG)
@given(shape=shapes, seed=st.integers(0, 2 ** 16), half=st.booleans())
def test_pack4_and_bias_correction_random_geometry(shape, seed, half):
    from cnn_quantization_amd import ops
    x, xd = make(shape, seed, 0)
    N, C, HW = shape[0], shape[1], shape[2] * shape[3]
    stats, _ = ops.pc_stats(xd, N, C, HW)
    qp, _ = ops.pc_params(stats, 4, half, 'no', False)
    y = ops.pc_qdq(xd, N, C, HW, qp)
    if HW % 4 == 0:
        packed = ops.quantize_pack4(xd, qp)
        assert torch.equal(ops.dequantize_pack4(packed, tuple(shape), qp), y)
    two = ops.act_bias_correction_(xd, y.clone(), half)
    one = ops.qdq_bias_corrected(xd, N, C, HW, qp, half)
    assert bits_equal(one.cpu().numpy(), two.cpu().numpy())
    ref = O.act_bias_correction(x, O.act_per_channel_qdq(x, 4, half_range=half), half)
    # the bias is a difference of two nearly equal channel sums: torch's fp32 reductions (the oracle) carry
    # ~1e-7 * sum|x| of error into it, the fp64 sums here do not - hence the wider absolute allowance
    np.testing.assert_allclose(one.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-4)


@settings(**CFG)
@given(n=st.integers(1, 5000), seed=st.integers(0, 2 ** 16), bits=st.sampled_from([4, 8]), etz=st.booleans(),
       rng=st.floats(0.05, 9.0), off=st.floats(-5.0, 1.0))
def test_float2gemmlowp_random(n, seed, bits, etz, rng, off):
    from cnn_quantization_amd import int_quantization
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, generator=g) * 2
    ref = O.float2gemmlowp(x, rng, off, bits, False, etz)
    out = int_quantization.float2gemmlowp(x.cuda(), rng, off, bits, False, etz, None)
    assert bits_equal(out.cpu(), ref)


@settings(**CFG)
@given(shape=shapes.filter(lambda s: s[0] * s[2] * s[3] >= 4), seed=st.integers(0, 2 ** 16), offset=st.integers(0, 3))
def test_statistics_random_geometry(shape, seed, offset):
    """The seven collect-mode statistics (both reduction passes, relu sums, kurtosis) on random geometry."""
    from cnn_quantization_amd import _lib as L
    from cnn_quantization_amd import ops
    x, xd = make(shape, seed, offset)
    ref = O.collect_stats_perchannel(x)
    st_, _ = ops.pc_stats(xd, shape[0], shape[1], shape[2] * shape[3], need_b=True, need_kurt=True, need_relu=True)
    st_ = st_.cpu()
    assert bits_equal(st_[L.STAT_MIN], ref['min']) and bits_equal(st_[L.STAT_MAX], ref['max'])
    for row, name in ((L.STAT_MEAN, 'mean'), (L.STAT_STD, 'std'), (L.STAT_B, 'b'), (L.STAT_STD_POS, 'std_pos')):
        np.testing.assert_allclose(st_[row], ref[name], rtol=1e-5, atol=4e-6, err_msg=name)
    ok = np.isfinite(ref['kurtosis'].numpy())
    np.testing.assert_allclose(st_[L.STAT_KURT].numpy()[ok], ref['kurtosis'].numpy()[ok], rtol=2e-3, atol=2e-3,
                               err_msg='kurtosis')


@settings(**CFG)
@given(shape=shapes.filter(lambda s: s[0] * s[2] * s[3] >= 8), seed=st.integers(0, 2 ** 16), half=st.booleans(),
       clip=st.sampled_from(['laplace', 'gaus']), ba=st.booleans())
def test_aciq_bit_exact_given_oracle_stats_random_geometry(shape, seed, half, clip, ba):
    """ACIQ clipping (+ bit allocation) with the oracle's statistics injected: parameters, codes and floats
    must then be bit-identical for any geometry."""
    from cnn_quantization_amd import _lib as L
    from cnn_quantization_amd import ops
    x, xd = make(shape, seed, 0)
    C = shape[1]
    st_ = O.act_stats_perchannel(x, ['min', 'max', 'b', 'std'])
    st_['mean'] = O.act_stats_perchannel(x, ['mean'], avg_over_batch=True)['mean']     # iq.py:335
    if not all(bool(torch.isfinite(v).all()) for v in st_.values()) or not bool((st_['b'] > 0).all()):
        return
    table = torch.zeros(L.NSTAT, C)
    for row, nm in ((L.STAT_MIN, 'min'), (L.STAT_MAX, 'max'), (L.STAT_MEAN, 'mean'), (L.STAT_B, 'b'), (L.STAT_STD, 'std')):
        table[row] = st_[nm]
    ref = O.act_clipping_qdq(x, 4, clip, half_range=half, bit_alloc_act=ba)
    y = ops.act_qdq_per_channel(xd, 4, positive=half, clip=clip, bit_alloc=ba, stats=table.cuda())
    assert bits_equal(y.cpu().numpy(), ref.numpy())


@settings(**CFG)
@given(shape=shapes.filter(lambda s: s[0] * s[2] * s[3] >= 8 and s[1] >= 2), seed=st.integers(0, 2 ** 16),
       half=st.booleans(), target=st.sampled_from([3.0, 4.0, 5.0]))
def test_midtread_bit_exact_given_oracle_stats_random_geometry(shape, seed, half, target):
    """Mid-tread quantization with bin allocation (config 5): with the oracle's statistics injected, the bin
    counts, clamp bounds, integer codes and dequantized floats are bit-identical for any geometry."""
    from cnn_quantization_amd import _lib as L
    from cnn_quantization_amd import ops
    from cnn_quantization_amd.qtypes._midtread_tables import ALPHA_TABLE, OMEGA_TABLE
    x, xd = make(shape, seed, 0)
    N, C, HW = shape[0], shape[1], shape[2] * shape[3]
    st_ = O.act_stats_perchannel(x, ['min', 'max', 'mean', 'b', 'std'])
    if not all(bool(torch.isfinite(v).all()) for v in st_.values()) or not bool((st_['std'] > 0).all()):
        return
    table = torch.zeros(L.NSTAT, C)
    for row, nm in ((L.STAT_MIN, 'min'), (L.STAT_MAX, 'max'), (L.STAT_MEAN, 'mean'), (L.STAT_B, 'b'), (L.STAT_STD, 'std')):
        table[row] = st_[nm]
    table = table.cuda()
    lib = L.load()
    mt = torch.empty((L.NMT, C), dtype=torch.float32, device='cuda')
    L.check(lib.cnnq_pc_midtread_params(ops._ptr(table), C, float(target), 1, int(not half),
                                        ops._ptr(ops._midtread_tables(table.device)), 101, ops._ptr(mt),
                                        ops._stream(table)), 'params')
    y, codes = torch.empty_like(xd), torch.empty_like(xd)
    L.check(lib.cnnq_pc_midtread_qdq(ops._ptr(xd), ops._ptr(y), N, C, HW, ops._ptr(mt), 1, ops._ptr(codes), None,
                                     ops._stream(xd)), 'qdq')
    rows = x.transpose(0, 1).contiguous().view(C, -1)
    ref_y, _, parts = O.mid_tread_core(rows, target, True, not half, np.asarray(OMEGA_TABLE), np.asarray(ALPHA_TABLE),
                                       return_parts=True)
    assert np.array_equal(mt[L.MT_OMEGA].cpu().numpy(), parts['omega'].numpy())
    back = lambda t: t.view(C, N, shape[2], shape[3]).transpose(0, 1).contiguous()
    # compared as values, not bit patterns: the SIGN of a zero code is not well defined in the CPU oracle itself -
    # torch.max(-0., +0.) returns the second operand inside its vectorised loop and the first one in the scalar
    # tail of the same tensor (the kernel here always returns the bound, as the vectorised loop does)
    assert np.array_equal(codes.cpu().numpy(), back(parts['codes']).numpy())
    assert np.array_equal(y.cpu().numpy(), back(ref_y).numpy())


@settings(**CFG)
@given(shape=st.tuples(st.integers(1, 48), st.integers(1, 24), st.sampled_from([1, 3, 5, 7])), seed=st.integers(0, 2 ** 16),
       ba=st.booleans(), vc=st.booleans(), bc=st.booleans())
def test_weights_random_geometry(shape, seed, ba, vc, bc):
    """Per-output-channel weight quantization (min/max exact -> bit-exact without bit allocation) and the
    bias / variance correction on random [OFM, IFM, K, K]."""
    from cnn_quantization_amd import ops
    g = torch.Generator().manual_seed(seed)
    w = torch.randn((shape[0], shape[1], shape[2], shape[2]), generator=g) * 0.1
    if w[0].numel() < 2:
        return
    wd = w.cuda()
    ref = O.weights_per_channel_qdq(w, 4, bit_alloc_weight=ba)
    out = ops.act_qdq_per_channel(wd, 4, clip='no', bit_alloc=ba, per_channel_dim=0, group=False)
    if not ba:
        assert bits_equal(out.cpu().numpy(), ref.numpy())
    else:
        assert ((out.cpu() - ref).abs() > 1e-6).float().mean() < 0.02       # std in fp64 vs fp32: rare allocation flips
    if vc or bc:
        c_ref = O.weight_correction(w, ref.clone(), vcorr=vc, bcorr=bc)
        c_out = ops.weight_correction(wd, ref.cuda(), vcorr=vc, bcorr=bc)
        np.testing.assert_allclose(c_out.cpu().numpy(), c_ref.numpy(), rtol=2e-4, atol=2e-6)


@settings(**CFG)
@given(shape=st.tuples(st.integers(1, 70), st.integers(1, 24), st.integers(1, 30), st.integers(1, 30)),
       seed=st.integers(0, 2 ** 20), offset=st.integers(0, 3), wpat=st.integers(0, 8), poff=st.sampled_from([0, 4, 8, 12]))
def test_bit_allocated_packing_random_geometry(shape, seed, offset, wpat, poff):
    """The bit-allocated packed format over random geometries (rows of 1 ... 900 elements, whole float4s or not, short
    and long rows), every width pattern, x / y at every 4-byte misalignment, the packed buffer at every 4-byte offset: the
    library's choice of kernel writes the bytes of the general kernel, decodes to the codes the fused Q/DQ produces, and
    the way back returns its floats - bit for bit; the lean forms are forced wherever they apply."""
    from cnn_quantization_amd import _lib as L
    from cnn_quantization_amd import ops
    N, C, H, W = shape
    x, xv = make(shape, seed, offset)
    g = torch.Generator().manual_seed(seed + 1)
    bits = ((torch.arange(C) + wpat) % 9).float().cuda()
    qp = torch.empty((L.NQP, C))
    qp[L.QP_SCALE] = torch.rand(C, generator=g) * 0.7 + 0.02
    qp[L.QP_ZP] = torch.randint(0, 6, (C,), generator=g).float()
    qp[L.QP_QMAX] = 2. ** bits.cpu() - 1.
    qp = qp.cuda()
    ref = ops.pc_qdq(xv, N, C, H * W, qp)
    ro = ops.packed_layout(bits, H * W)
    cap = ops.packed_capacity(shape)
    pbase = torch.zeros(cap + 16, dtype=torch.uint8, device='cuda')
    a, _ = ops.quantize_packed(xv, qp, bits, form=1)
    used = a.numel()
    for form in (0, 2):
        out = pbase[poff:poff + cap]
        try:
            b, ro_b = ops.quantize_packed(xv, qp, bits, out=out, form=form, rowoff=ro)
        except L.CnnqError:
            assert form == 2                                 # the lean kernel refuses rows of < 8 elements / unaligned whole-float4 rows
            continue
        assert torch.equal(b[:used], a), (shape, form, offset, poff)
    ybase = torch.empty(int(np.prod(shape)) + 3, device='cuda')
    yv = ybase[offset:offset + int(np.prod(shape))].view(shape)
    pk = pbase[poff:poff + cap]
    pk[:used].copy_(a)
    for form in (0, 1, 2):
        try:
            back = ops.dequantize_packed(pk, shape, qp, bits, ro, out=yv, form=form)
        except L.CnnqError:
            assert form == 2
            continue
        assert torch.equal(back, ref), (shape, form, offset, poff)
