"""The seven per-channel statistics of `-sm collect` (statistic_manager_perchannel.py:45-79; BASELINE config 4) from ONE launch
that reads x once (cnnq_pc_stats_single, csrc/cnnq_stats1.hip.h).  What must hold:

* extrema exact; mean / std / std_pos / b to 2e-6 and the kurtosis to 1e-4 of fp64 torch reductions, and of the three-launch
  chain (the same per-element arithmetic and final formulas, another fixed order of the fp64 additions);
* determinism: the forced recompute path == the meeting == a second run, bit for bit; workspace zero at rest;
* NaN semantics of the chain: a NaN poisons exactly its channel;
* short channel rows (14x14, 7x7 with channels straddling the 16-byte loads, rows that are whole multiples of the workgroup)
  through the row-piece tiles (k_stats_group), every tile height; shapes no single-launch plan takes answer None and
  ops.pc_stats takes the chain.
Needs an MI355X: `pytest -m gpu`."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from cnn_quantization_amd import ops as _ops
    return _ops


def acts(shape, seed):
    gen = torch.Generator().manual_seed(seed)
    C = shape[1]
    x = torch.randn(shape, generator=gen) * (torch.rand(1, C, 1, 1, generator=gen) * 3 + 0.05) + torch.randn(1, C, 1, 1, generator=gen)
    if C > 2:
        x[:, C // 2] = 0.25
    return x.contiguous()


def ref64(x):
    from cnn_quantization_amd import _lib as L
    C = x.shape[1]
    t = x.cuda().double().transpose(0, 1).reshape(C, -1)
    mean, std = t.mean(1), t.std(1, unbiased=True)
    out = torch.zeros(L.NSTAT, C, dtype=torch.float64, device='cuda')
    out[L.STAT_MIN], out[L.STAT_MAX], out[L.STAT_MEAN], out[L.STAT_STD] = t.min(1)[0], t.max(1)[0], mean, std
    m32, s32 = mean.float().double(), std.float().double()
    out[L.STAT_B] = (t - m32[:, None]).abs().mean(1)
    out[L.STAT_KURT] = (((t - m32[:, None]) / s32[:, None]) ** 4).mean(1) - 3
    out[L.STAT_STD_POS] = t.clamp(min=0).std(1, unbiased=True)
    return out.cpu()


SHAPES = [(40, 6, 56, 56), (33, 5, 28, 28), (8, 4, 112, 112), (300, 3, 56, 56), (64, 7, 56, 56), (16, 3, 28, 28), (300, 2, 112, 112)]
# row-piece tiles: whole channels per workgroup (14x14), channels straddling the loads (7x7: one channel per element), a row
# that is a whole multiple of the workgroup (64x64: column blocks of one channel), ragged sample splits, and the tile heights
# the planner picks at full size (K = 16: [512,160,14,14] and - straddling rows take at most K = 16 since round 6, their K = 32
# instances spilled - [512,640,7,7]; K = 32 with eight rows in LDS: [512,320,14,14])
GROUP_SHAPES = [(12, 24, 14, 14), (70, 12, 14, 14), (64, 37, 14, 14), (40, 44, 7, 7), (33, 12, 7, 7), (6, 3, 64, 64), (300, 8, 7, 7), (130, 20, 14, 14)]
BIG_GROUP_SHAPES = [(512, 160, 14, 14), (512, 320, 14, 14), (512, 640, 7, 7)]


@pytest.mark.parametrize('shape', SHAPES + GROUP_SHAPES)
@pytest.mark.parametrize('need', [(True, True, True), (True, False, False), (False, False, False), (False, False, True)])
def test_single_launch_statistics(ops, shape, need):
    check_statistics(ops, shape, need)


@pytest.mark.parametrize('shape', BIG_GROUP_SHAPES)
@pytest.mark.parametrize('need', [(True, True, True), (True, False, False)])
def test_single_launch_statistics_full_tile_heights(ops, shape, need):
    check_statistics(ops, shape, need)


def test_reference_generated_collection_through_the_single_launch(ops, golden, tmp_path, monkeypatch):
    """k_stats_flat against vectors of the REFERENCE (statistic_manager_perchannel.py:45-79 run on [16,3,28,28],
    tests/golden/make_golden_collect_flat.py), through the route that takes the single launch: ops.pc_stats_single itself and the
    product's StatisticManagerPerChannel.save_tensor_stats (which must end up in it: cnnq_pc_stats_auto).  Extrema bit for bit;
    mean / std / b / std_pos within the fp64-sum tier of test_hip_parity.test_stats_collect_set_vs_oracle; kurtosis as there."""
    from cnn_quantization_amd import _lib as L
    from cnn_quantization_amd.inference import statistic_manager_perchannel as smpc
    g = golden('collect_flat')
    rows = (('max', L.STAT_MAX), ('min', L.STAT_MIN), ('std', L.STAT_STD), ('mean', L.STAT_MEAN), ('kurtosis', L.STAT_KURT),
            ('b', L.STAT_B), ('std_pos', L.STAT_STD_POS))

    def close(name, got, want, exact=True):
        if name in ('max', 'min') and exact:
            assert bits_equal(got, want), name
        elif name in ('max', 'min'):        # batch_avg: the mean over the batch of the per-sample extrema (fp64 here, fp32 there)
            np.testing.assert_allclose(got, want, rtol=2e-6, err_msg=name)
        elif name == 'kurtosis':
            np.testing.assert_allclose(got, want, rtol=1e-3, atol=5e-4, err_msg=name)
        else:
            np.testing.assert_allclose(got, want, rtol=5e-6, atol=2e-6, err_msg=name)
    for k in range(2):
        xd = g.t('x%d' % k).cuda()
        N, C, H, W = xd.shape
        res = ops.pc_stats_single(xd, N, C, H * W, True, True, True, flags=0)        # flags 0: the product's own routing
        assert res is not None, 'no single-launch plan for the golden shape'
        st = res[0].cpu()
        for name, row in rows:
            close(name, st[row].numpy(), g.np('b0_%s' % name)[k])
    assert ops.group_status(xd) == 0
    # ... and through the manager (same pickle rows as the reference's manager holds), batch_avg off and on
    monkeypatch.setenv('HOME', str(tmp_path))
    for bi, batch_avg in enumerate((False, True)):
        smpc.Singleton._instances.pop(smpc.StatisticManagerPerChannel, None)
        sm = smpc.StatisticManagerPerChannel('flat_%d' % bi, load_stats=False, batch_avg=batch_avg,
                                             stats=['max', 'min', 'std', 'mean', 'kurtosis', 'b', 'std_pos'])
        for k in range(2):
            sm.save_tensor_stats(g.t('x%d' % k).cuda(), 'activation', 'conv0_activation')
        for name, _ in rows:
            got = sm.stats['conv0_activation'][name]
            assert got.shape == (2, 3)
            for k in range(2):
                close(name, got[k], g.np('b%d_%s' % (bi, name))[k], exact=not batch_avg)
    smpc.Singleton._instances.pop(smpc.StatisticManagerPerChannel, None)


def check_statistics(ops, shape, need):
    from cnn_quantization_amd import _lib as L
    need_b, need_kurt, need_relu = need
    N, C = shape[:2]
    HW = shape[2] * shape[3]
    x = acts(shape, 3 + C + HW)
    xd = x.cuda()
    ops.group_status(xd, clear=True)
    res = ops.pc_stats_single(xd, N, C, HW, need_b, need_kurt, need_relu)
    assert res is not None
    st, mom = res
    ref = ref64(x)
    live = torch.arange(C) != (C // 2 if C > 2 else -1)          # the constant channel: std == 0, kurtosis 0 / 0
    assert bits_equal(st[L.STAT_MIN].cpu(), ref[L.STAT_MIN].float()) and bits_equal(st[L.STAT_MAX].cpu(), ref[L.STAT_MAX].float())
    np.testing.assert_allclose(st[L.STAT_MEAN].cpu().double(), ref[L.STAT_MEAN], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(st[L.STAT_STD].cpu().double()[live], ref[L.STAT_STD][live], rtol=2e-6)
    if need_b or need_kurt:
        np.testing.assert_allclose(st[L.STAT_B].cpu().double(), ref[L.STAT_B], rtol=3e-6, atol=1e-7)
    else:
        assert float(st[L.STAT_B].abs().max()) == 0.
    if need_kurt:
        np.testing.assert_allclose(st[L.STAT_KURT].cpu().double()[live], ref[L.STAT_KURT][live], rtol=2e-4, atol=2e-4)
    if need_relu:
        np.testing.assert_allclose(st[L.STAT_STD_POS].cpu().double(), ref[L.STAT_STD_POS], rtol=3e-6, atol=1e-7)
    # the merged moment record: count exact, sums to fp64 rounding
    t = xd.double().transpose(0, 1).reshape(C, -1)
    assert torch.equal(mom[L.MOM_COUNT].cpu(), torch.full((C,), float(N * HW), dtype=torch.float64))
    np.testing.assert_allclose(mom[L.MOM_SUM].cpu(), t.sum(1).cpu(), rtol=1e-7, atol=1e-3)      # fp32 4-sums accumulated in fp64 (Mom::add4)
    np.testing.assert_allclose(mom[L.MOM_SUMSQ].cpu(), (t * t).sum(1).cpu(), rtol=1e-7)
    # ... and the chain on the same tensor
    ops._ACIQ_SINGLE = False
    try:
        st0, mom0 = ops.pc_stats(xd, N, C, HW, need_b=need_b, need_kurt=need_kurt, need_relu=need_relu)
    finally:
        ops.reload_switches()
    assert bits_equal(st[:2].cpu(), st0[:2].cpu())
    # (every row but the kurtosis at the tier of two summation orders; the kurtosis moves by ~4 ulp(std) * (kurt + 3) with the last
    #  bit of the fp32 std it divides by - the single launch sums squares in fp32 groups of 8, the chain in groups of 4)
    rows = [r for r in range(L.NSTAT) if r != L.STAT_KURT]
    np.testing.assert_allclose(st.cpu()[rows][:, live], st0.cpu()[rows][:, live], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(st.cpu()[L.STAT_KURT][live], st0.cpu()[L.STAT_KURT][live], rtol=2e-5, atol=2e-5)
    assert ops.group_status(xd) == 0


@pytest.mark.parametrize('shape', SHAPES[:4] + GROUP_SHAPES + BIG_GROUP_SHAPES[1:])
def test_recompute_path_and_reruns_give_the_same_bits(ops, shape):
    N, C = shape[:2]
    HW = shape[2] * shape[3]
    xd = acts(shape, 21 + C).cuda()
    ops.group_status(xd, clear=True)
    st0, mom0 = ops.pc_stats_single(xd, N, C, HW, True, True, True)
    st0, mom0 = st0.clone(), mom0.clone()
    for flags in (8, 9, 8, 9):
        st1, mom1 = ops.pc_stats_single(xd, N, C, HW, True, True, True, flags=flags)
        # (the constant channel's kurtosis is 0 * inf: a NaN whose sign / payload is not part of the contract)
        assert bits_equal(torch.nan_to_num(st1, nan=77.).cpu(), torch.nan_to_num(st0, nan=77.).cpu()), flags
        assert torch.equal(torch.isnan(st1), torch.isnan(st0))
        assert torch.equal(mom1.cpu().view(torch.int64), mom0.cpu().view(torch.int64)), flags
    assert ops.group_status(xd, clear=True) == ops.GROUP_TEST_HOOK
    from cnn_quantization_amd import _lib
    nz = ctypes.c_uint64()
    ws = ops._GROUP_WS[(xd.device.index, ops._raw_stream(xd.device.index))]
    assert _lib.load().cnnq_group_ws_at_rest(ws, ctypes.byref(nz)) == 0 and nz.value == 0
    # the other single-launch kernels share the slot region: config 2 right behind it on the same workspace
    y = ops.act_qdq_per_channel(xd, 4)
    yc = ops.minmax_qdq_fused(xd, N, C, HW, 4, False, chain=True)
    assert torch.equal(y, yc)


@pytest.mark.parametrize('shape', [(40, 6, 56, 56), (40, 6, 14, 14), (40, 8, 7, 7)])
def test_nan_poisons_its_channel_only(ops, shape):
    from cnn_quantization_amd import _lib as L
    x = acts(shape, 9)
    x[3, 2, 5, 3] = float('nan')
    xd = x.cuda()
    HW = shape[2] * shape[3]
    st, _ = ops.pc_stats_single(xd, shape[0], shape[1], HW, True, True, True)
    st = st.cpu()
    for row in (L.STAT_MIN, L.STAT_MAX, L.STAT_MEAN, L.STAT_STD, L.STAT_B, L.STAT_KURT):
        assert bool(torch.isnan(st[row, 2])), row
    keep = torch.tensor([c for c in range(shape[1]) if c not in (2, shape[1] // 2)])     # (the constant channel's kurtosis is 0 / 0)
    assert not bool(torch.isnan(st[:, keep]).any())
    ops._ACIQ_SINGLE = False                                   # the chain's pattern (relu(NaN) = 0: its std_pos stays finite)
    try:
        st0, _ = ops.pc_stats(xd, shape[0], shape[1], HW, need_b=True, need_kurt=True, need_relu=True)
    finally:
        ops.reload_switches()
    assert torch.equal(torch.isnan(st), torch.isnan(st0.cpu()))


def test_shapes_without_a_single_launch_plan(ops):
    x = acts((12, 5, 7, 7), 1).cuda()                            # neither H*W nor C*H*W a multiple of 4: no 16-byte tiling
    assert ops.pc_stats_single(x, 12, 5, 49, True, True, True) is None
    st, _ = ops.pc_stats(x, 12, 5, 49, need_b=True)              # the chain
    assert torch.equal(st[1].cpu(), x.cpu().amax(dim=(0, 2, 3)))
