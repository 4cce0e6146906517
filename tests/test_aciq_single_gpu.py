"""Config 3 (ACIQ Laplace clipping, dynamic statistics, bit allocation) with pass B, the parameter derivation and the
Q/DQ in ONE launch (cnnq_pc_aciq_qdq_single, csrc/cnnq_aciq.hip.h; iq.py:327-352 -> :227-253, 284-300, 393-407 ->
:409-451, 557-603).  What must hold:

* the reference-generated golden vectors within the statistics tier of the chain (b is an fp64 sum here, an fp32
  reduction in torch): identical bit allocation, alpha / range / offset to 2e-6, codes off by at most one step on at
  most 2e-4 of the elements - through the single launch, on every golden case whose shape has a single-launch plan;
* given the device's own b, everything downstream is the ORACLE's arithmetic bit for bit: alpha, delta / offset,
  scale / zero point / qmax from the oracle's functions on the device's statistics table, y and the codes from
  oracle.qdq_core on the CPU with those parameters;
* determinism of the in-launch sum exchange: the forced recompute path (every member's partial sum recomputed by every
  workgroup and folded in member order) gives the SAME b, bit for bit, as the meeting - on every tile shape - and so
  does a second run; the IEEE-divide flag changes nothing;
* the chain (five launches) and the single launch agree: tables equal except, rarely, the last bit of b.
Needs an MI355X: `pytest -m gpu`."""
import ctypes

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


def describe(N, C, HW):
    from cnn_quantization_amd import _lib
    out = (ctypes.c_int32 * 8)()
    rc = _lib.load().cnnq_pc_group_describe(N, C, HW, out)
    return rc, dict(zip(('A', 'K', 'mode', 'S', 'ncb', 'Gs', 'groups', 'wgs'), list(out)))


def acts(shape, seed, relu=False):
    """Laplace-ish activations with per-channel scale and shift (and a dead channel)."""
    gen = torch.Generator().manual_seed(seed)
    N, C = shape[:2]
    x = torch.empty(shape).exponential_(1.0, generator=gen) * (torch.rand(shape, generator=gen) < 0.5).float().mul_(2).sub_(1)
    x = x * (torch.rand(1, C, 1, 1, generator=gen) * 3 + 0.05) + torch.randn(1, C, 1, 1, generator=gen) * 0.3
    if relu:
        x = x.clamp_(min=0)
    if C > 2:
        x[:, C // 2] = 0.25 if not relu else 0.          # a constant channel: b == 0, scale at its floor
    return x.contiguous()


from _direct import aciq_on_table as oracle_from_device_stats      # noqa: E402  (shared with the sharded tests since round 6)


# flat tiles (56x56, 28x28: one channel per group), row pieces (32x32: cpc = 256), whole channels per workgroup with the
# exchange over the batch splits (14x14), straddling float4s (7x7), two-level departure counters (many members), row
# pieces of one channel (64x64: four workgroups per sample row)
SHAPES = [(40, 6, 56, 56), (33, 5, 28, 28), (24, 3, 32, 32), (70, 12, 14, 14), (130, 24, 7, 7), (8, 4, 112, 112), (64, 37, 14, 14),
          (6, 3, 64, 64)]


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('half', [False, True])
@pytest.mark.parametrize('ba', [False, True])
def test_single_launch_equals_the_oracle_on_its_own_statistics(ops, shape, half, ba):
    from cnn_quantization_amd import _lib as L
    N, C = shape[:2]
    HW = shape[2] * shape[3]
    rc, d = describe(N, C, HW)
    assert rc == 0, d
    x = acts(shape, 7 + C + HW, relu=half)
    xd = x.cuda()
    ops.group_status(xd, clear=True)
    res = ops.aciq_qdq_single(xd, N, C, HW, 4, half, ba, None, True, want_codes=True, want_parts=True)
    assert res is not None, d
    y, codes, parts = res
    st, qp, diag = parts['stats'], parts['qp'], parts['diag']
    # statistics: extrema exact, mean / std / b against fp64
    t64 = x.double().transpose(0, 1).reshape(C, -1)
    assert torch.equal(st[L.STAT_MAX].cpu(), x.amax(dim=(0, 2, 3))) and torch.equal(st[L.STAT_MIN].cpu(), x.amin(dim=(0, 2, 3)))
    m64 = t64.mean(1)
    b64 = (t64 - st[L.STAT_MEAN].cpu().double()[:, None]).abs().mean(1)      # around the device's fp32 mean, as the kernel
    np.testing.assert_allclose(st[L.STAT_MEAN].cpu().double(), m64, rtol=RTOL_STAT, atol=1e-7)
    np.testing.assert_allclose(st[L.STAT_B].cpu().double(), b64, rtol=RTOL_STAT, atol=1e-9)
    ref = oracle_from_device_stats(x, st, diag[L.DIAG_BITS], 4, half, ba)
    assert bits_equal(diag[L.DIAG_ALPHA].cpu(), ref['alpha'])
    assert bits_equal(diag[L.DIAG_DELTA].cpu(), ref['delta']) and bits_equal(diag[L.DIAG_OFFSET].cpu(), ref['offset'])
    assert bits_equal(qp[L.QP_SCALE].cpu(), ref['scale']) and bits_equal(qp[L.QP_ZP].cpu(), ref['zp'])
    assert bits_equal(qp[L.QP_QMAX].cpu(), ref['qmax'])
    assert torch.equal(codes.cpu().float(), ref['codes'])
    assert bits_equal(y.cpu(), ref['y'])
    assert ops.group_status(xd) == 0


@pytest.mark.parametrize('shape', SHAPES)
def test_recompute_path_and_reruns_give_the_same_bits(ops, shape):
    """The sum exchange is deterministic: meeting == forced recompute (flag 1) == IEEE divide (flag 2) == a second run."""
    from cnn_quantization_amd import _lib as L
    N, C = shape[:2]
    HW = shape[2] * shape[3]
    xd = acts(shape, 99 + C, relu=False).cuda()
    ops.group_status(xd, clear=True)
    y0, p0 = ops.aciq_qdq_single(xd, N, C, HW, 4, False, True, None, True, want_parts=True)
    tabs0 = torch.cat([p0['stats'], p0['qp'], p0['diag']]).clone()
    y0 = y0.clone()
    for flags in (0, 1, 2, 3):
        y1, p1 = ops.aciq_qdq_single(xd, N, C, HW, 4, False, True, None, True, want_parts=True, flags=flags)
        assert bits_equal(torch.cat([p1['stats'], p1['qp'], p1['diag']]).cpu(), tabs0.cpu()), flags
        assert bits_equal(y1.cpu(), y0.cpu()), flags
    st = ops.group_status(xd, clear=True)
    assert st == ops.GROUP_TEST_HOOK, st                     # the hook was used (bit 1), no wait expired (bit 0)
    from cnn_quantization_amd import _lib
    nz = ctypes.c_uint64()
    ws = ops._GROUP_WS[(xd.device.index, ops._raw_stream(xd.device.index))]
    assert _lib.load().cnnq_group_ws_at_rest(ws, ctypes.byref(nz)) == 0 and nz.value == 0      # slots and counters re-armed


@pytest.mark.parametrize('shape', SHAPES[:5])
@pytest.mark.parametrize('half', [False, True])
def test_single_launch_against_the_chain(ops, shape, half):
    """Same tables as the five-launch chain (pass B as a launch of its own), except - rarely - the last bit of b; where a
    channel's parameters agree its outputs are bit-identical."""
    from cnn_quantization_amd import _lib as L
    N, C = shape[:2]
    HW = shape[2] * shape[3]
    xd = acts(shape, 5 + HW, relu=half).cuda()
    y1, c1, p1 = ops.aciq_qdq_single(xd, N, C, HW, 4, half, True, None, True, want_codes=True, want_parts=True)
    ops._ACIQ_SINGLE = False
    try:
        y0, c0, p0 = ops.act_qdq_per_channel(xd, 4, positive=half, clip='laplace', bit_alloc=True, want_codes=True, want_parts=True)
    finally:
        ops.reload_switches()
    for r in (L.STAT_MIN, L.STAT_MAX, L.STAT_MEAN, L.STAT_STD):
        assert bits_equal(p1['stats'][r].cpu(), p0['stats'][r].cpu()), r
    assert bits_equal(p1['diag'][L.DIAG_BITS].cpu(), p0['diag'][L.DIAG_BITS].cpu())
    np.testing.assert_allclose(p1['stats'][L.STAT_B].cpu(), p0['stats'][L.STAT_B].cpu(), rtol=2e-7, atol=0)
    same = (p1['qp'] == p0['qp']).all(0) | (torch.isnan(p1['qp']) & torch.isnan(p0['qp'])).all(0)
    assert int(same.sum()) >= C - 1, (int(same.sum()), C)
    assert torch.equal(y1[:, same], y0[:, same]) and torch.equal(c1[:, same], c0[:, same])


def test_golden_cfg3_through_the_single_launch(ops, golden):
    """The reference-generated ACIQ vectors (tests/golden/act_pc.npz), through ops.act_qdq_per_channel - which now routes
    Laplace clipping to the single launch when the shape has a plan: the tier of tests/test_hip_parity.py."""
    from cnn_quantization_amd import _lib as L
    from test_oracle_golden import ACT_KW
    g = golden('act_pc')
    total = diff = n_single = 0
    for key in g.np('names'):
        key = str(key)
        name, si = key.rsplit('_s', 1)
        kw = dict(ACT_KW[name])
        if kw.get('clip') != 'laplace' or kw.get('bit_alloc_prior', 'gaus') == 'laplace':
            continue
        x = g.t('x' + si)
        N, C = x.shape[:2]
        HW = x[0, 0].numel()
        bits, half = int(g.np(key + '_bits')), bool(g.np(key + '_half'))
        ba = bool(kw.get('bit_alloc_act', False))
        res = ops.aciq_qdq_single(x.cuda(), N, C, HW, bits, half, ba, kw.get('bit_alloc_target'), kw.get('bit_alloc_round', True),
                                  want_codes=True, want_parts=True)
        if res is None:
            assert describe(N, C, HW)[0] != 0, key
            continue
        n_single += 1
        y, codes, parts = res
        diag = parts['diag'].cpu()
        if (key + '_bit_alloc') in g:
            assert np.array_equal(diag[L.DIAG_BITS].numpy(), g.np(key + '_bit_alloc')), key
        np.testing.assert_allclose(diag[L.DIAG_ALPHA], g.np(key + '_alpha'), rtol=RTOL_STAT, err_msg=key)
        np.testing.assert_allclose(diag[L.DIAG_DELTA], g.np(key + '_range'), rtol=2 * RTOL_STAT, err_msg=key)
        np.testing.assert_allclose(diag[L.DIAG_OFFSET], g.np(key + '_offset'), rtol=2 * RTOL_STAT, atol=1e-6, err_msg=key)
        c = codes.cpu().numpy().astype(np.int32)
        d = np.abs(c - g.np(key + '_codes'))
        assert d.max() <= 1, key
        total += d.size
        diff += int((d != 0).sum())
        np.testing.assert_allclose(y.cpu().numpy(), g.np(key + '_y'), rtol=1e-5, atol=float(parts['qp'][0].max()) * 1.001, err_msg=key)
    assert n_single >= 4, n_single
    assert diff <= 2e-4 * total, (diff, total)


def test_entropy_from_the_single_launch(ops):
    """-me with config 3: the replica histogram of the single launch gives the entropy of the codes it wrote."""
    shape = (40, 6, 56, 56)
    N, C = shape[:2]
    HW = shape[2] * shape[3]
    xd = acts(shape, 3).cuda()
    y, codes, ent = ops.aciq_qdq_single(xd, N, C, HW, 4, False, True, None, True, want_codes=True, want_entropy=True)
    cnt = torch.bincount(codes.flatten().long(), minlength=256).double()
    pr = cnt[cnt > 0] / codes.numel()
    ref = float(-(pr * torch.log2(pr)).sum())
    assert abs(float(ent) - ref) <= 2e-5 * max(1., ref)


def test_unsupported_configurations_fall_back(ops):
    """No plan / another clipping / the 'laplace' prior: None from the single entry point, the chain through the pipeline."""
    x = acts((5, 6, 1, 3), 1).cuda()                         # C * H*W = 18: no float4 layout at all
    assert ops.aciq_qdq_single(x, 5, 6, 3, 4) is None
    y = ops.act_qdq_per_channel(x, 4, clip='laplace', bit_alloc=True)
    ref = O.act_clipping_qdq(x.cpu(), 4, 'laplace', bit_alloc_act=True)
    assert float(((y.cpu() - ref).abs() > 1e-5).float().mean()) < 1e-3
    x = acts((40, 6, 56, 56), 2).cuda()
    ya = ops.act_qdq_per_channel(x, 4, clip='laplace', bit_alloc=True, prior_is_b=True)       # chain: the prior is b itself
    refa = O.act_clipping_qdq(x.cpu(), 4, 'laplace', bit_alloc_act=True, bit_alloc_prior='laplace')
    assert float(((ya.cpu() - refa).abs() > 1e-5).float().mean()) < 1e-3


@pytest.mark.parametrize('half', [False, True])
def test_nan_and_inf_channels_behave_like_the_chain(ops, half):
    """A NaN activation poisons exactly its channel (statistics, parameters, every output of the channel), +-inf saturate a
    channel's range - in the single launch as in the five-launch chain (whose semantics are pinned to the reference by
    tests/golden/nan.npz): same NaN pattern, same bits elsewhere."""
    shape = (40, 6, 56, 56)
    N, C, HW = 40, 6, 56 * 56
    x = acts(shape, 17, relu=half)
    x[3, 1, 5, 7] = float('nan')
    x[7, 4, 0, 0] = float('inf')
    if not half:
        x[9, 5, 2, 2] = float('-inf')
    xd = x.cuda()
    y1, p1 = ops.aciq_qdq_single(xd, N, C, HW, 4, half, True, None, True, want_parts=True)
    ops._ACIQ_SINGLE = False
    try:
        y0, p0 = ops.act_qdq_per_channel(xd, 4, positive=half, clip='laplace', bit_alloc=True, want_parts=True)
    finally:
        ops.reload_switches()
    assert torch.equal(torch.isnan(y1), torch.isnan(y0))
    assert not bool(torch.isnan(y1[:, 0]).any())
    assert torch.equal(torch.isnan(p1['qp']), torch.isnan(p0['qp'])) and torch.equal(torch.isnan(p1['stats']), torch.isnan(p0['stats']))
    same = ((p1['qp'] == p0['qp']) | (torch.isnan(p1['qp']) & torch.isnan(p0['qp']))).all(0)
    assert int(same.sum()) >= C - 1
    assert torch.equal(torch.nan_to_num(y1[:, same], nan=7.), torch.nan_to_num(y0[:, same], nan=7.))
