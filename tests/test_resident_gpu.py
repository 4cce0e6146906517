"""The register-resident single-launch form of config 2 (cnnq_pc_minmax_qdq_resident, csrc/cnnq_resident.hip.h):
same bits as the reference-pinned golden vectors, the oracle and the three-launch chain on every tile shape
(T = 256 / 512 / 1024 lanes, K = 8 / 16 / 32 samples per lane, plain and straddling float4 columns, ragged last
channel blocks and sample counts), NaN / inf semantics of torch.min / torch.max, HIP-graph replay, and the
BASELINE-sized layers of the batch-64 shard.  Needs an MI355X: `pytest -m gpu`."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import bits_equal
from oracle import quant_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from cnn_quantization_amd import ops as _ops
    return _ops


def describe(N, C, HW):
    from cnn_quantization_amd import _lib
    out = (ctypes.c_int32 * 8)()
    rc = _lib.load().cnnq_pc_resident_describe(N, C, HW, out)
    return rc, dict(zip(('A', 'T', 'K', 'k', 'CL', 'RL', 'wgs'), list(out)))


def classic(ops, x, bits, half):
    """The three-launch chain (statistics pass, parameters, Q/DQ pass): chain=True keeps it off the resident path."""
    N, C = x.shape[:2]
    y, codes, parts = ops.minmax_qdq_fused(x, N, C, x[0, 0].numel(), bits, half, want_codes=True, want_parts=True, chain=True)
    return y, parts


def test_resident_golden_bit_exact(ops, golden):
    """The 15 config-2 cases recorded from the reference: dequantized floats bit for bit, min / max / scale / zero
    point bit for bit (codes follow from those: y = (code - zp) * scale)."""
    from cnn_quantization_amd import _lib as L
    g = golden('act_pc')
    n = 0
    for key in g.np('names'):
        key = str(key)
        name, si = key.rsplit('_s', 1)
        if not name.startswith('cfg2') or 'baa' in name:
            continue
        x = g.t('x' + si).cuda()
        N, C = x.shape[:2]
        bits, half = int(g.np(key + '_bits')), bool(g.np(key + '_half'))
        res = ops.minmax_qdq_resident(x, N, C, x[0, 0].numel(), bits, half, want_parts=True)
        if describe(N, C, x[0, 0].numel())[0] != 0:      # [3,16,5,9]: H*W = 45, no resident kernel
            assert res is None
            continue
        assert res is not None, key
        y, parts = res
        assert bits_equal(y.cpu(), g.np(key + '_y')), key
        assert bits_equal(parts['stats'][L.STAT_MAX].cpu(), g.np('s%s_stat_max' % si)), key
        assert bits_equal(parts['stats'][L.STAT_MIN].cpu(), g.np('s%s_stat_min' % si)), key
        yc, pc = classic(ops, x, bits, half)
        assert torch.equal(parts['qp'], pc['qp']), key
        n += 1
    assert n == 12


SHAPES = [
    (3, 8, 7, 7), (5, 12, 7, 7), (4, 6, 1, 2),   # straddling float4s: blocks of 4 (2) channels
    (70, 40, 7, 7),      # straddle, K = 16, ragged samples
    (64, 2048, 7, 7), (160, 36, 7, 7),
    (2, 8, 14, 14), (37, 24, 14, 14), (64, 256, 14, 14), (80, 6, 14, 14), (160, 3, 14, 14),
    (5, 3, 28, 28), (33, 16, 28, 28), (64, 7, 28, 28),
    (9, 4, 56, 56), (8, 5, 56, 56), (16, 3, 56, 56),    # T = 1024
    (130, 2, 8, 8), (1, 1100, 2, 2), (4, 260, 1, 4), (2, 3, 64, 48), (1, 64, 3, 3 * 4), (600, 2, 4, 4), (1, 5, 32, 128),
]


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('half', [False, True])
def test_resident_equals_chain_and_oracle(ops, shape, half):
    gen = torch.Generator().manual_seed(sum(shape) + int(half))
    N, C, H, W = shape
    x = torch.randn(shape, generator=gen) * (torch.rand(1, C, 1, 1, generator=gen) * 4 + 0.05) + \
        torch.randn(1, C, 1, 1, generator=gen)
    rc, d = describe(N, C, H * W)
    xd = x.cuda()
    ref = O.act_per_channel_qdq(x, 4, half_range=half)
    res = ops.minmax_qdq_resident(xd, N, C, H * W, 4, half, want_parts=True)
    if rc != 0:
        assert res is None
        pytest.skip('no resident kernel for %r' % (shape,))
    y, parts = res
    assert bits_equal(y.cpu(), ref), (shape, d)
    yc, pc = classic(ops, xd, 4, half)
    assert torch.equal(yc, y) and torch.equal(pc['qp'], parts['qp'])
    assert torch.equal(parts['stats'][0], xd.amin(dim=(0, 2, 3))) and torch.equal(parts['stats'][1], xd.amax(dim=(0, 2, 3)))


def test_resident_plans_cover_the_tile_shapes():
    """The shapes above reach every instantiated (A, T, K) combination that the ResNet-50 batch-64 shard uses,
    and the batch-512 layers have no resident kernel (they take the chain)."""
    seen = set()
    for shape in SHAPES + [(64, 512, 28, 28), (64, 1024, 14, 14), (64, 512, 7, 7)]:
        rc, d = describe(shape[0], shape[1], shape[2] * shape[3])
        if rc == 0:
            seen.add((d['A'], d['T'], d['K']))
            assert d['RL'] * d['K'] >= shape[0] and d['CL'] * d['RL'] <= d['T']
    assert {(1, 256, 8), (1, 256, 16), (1, 256, 32), (1, 512, 32), (1, 1024, 8), (1, 1024, 16), (4, 256, 8), (4, 256, 16),
            (4, 256, 32)} <= seen, seen
    for hw, C in ((112, 64), (56, 256), (28, 512), (14, 1024), (7, 2048)):
        assert describe(512, C, hw * hw)[0] == -3
    assert describe(64, 64, 112 * 112)[0] == -3 and describe(64, 256, 56 * 56)[0] == -3


def test_resident_unsupported_shapes_take_the_chain(ops):
    """Unaligned base pointers and H*W that neither is a multiple of 4 nor straddles cleanly: no resident kernel,
    act_qdq_per_channel still answers (three-launch chain) with the oracle's bits."""
    x = torch.randn(3, 5, 5, 9)
    assert ops.minmax_qdq_resident(x.cuda(), 3, 5, 45, 4) is None
    buf = torch.empty(2 * 8 * 16 + 1, device='cuda')
    xu = buf[1:].view(2, 8, 4, 4)
    xu.copy_(torch.randn(2, 8, 4, 4))
    assert ops.minmax_qdq_resident(xu, 2, 8, 16, 4) is None
    big = torch.randn(512, 8, 14, 14)                      # 100 K elements per channel: more than a register tile
    assert ops.minmax_qdq_resident(big.cuda(), 512, 8, 196, 4) is None
    for t in (x.cuda(), xu, big.cuda()):
        assert bits_equal(ops.act_qdq_per_channel(t, 4).cpu(), O.act_per_channel_qdq(t.cpu(), 4))


def test_resident_nan_inf_follow_torch(ops):
    """torch.min / torch.max propagate NaN (iq.py:416,423): a NaN poisons exactly its channel; +-inf give an
    infinite range.  Same bits as the oracle (the reference's own op chain on CPU)."""
    gen = torch.Generator().manual_seed(3)
    def same(a, b):
        # NaN sign / payload bits are not semantics (CPU torch itself returns 0xffc00000 in the scalar tail of a
        # vectorised loop and 0x7fc00000 inside it): NaN exactly where the reference has NaN, same bits elsewhere
        a, b = a.numpy(), b.numpy()
        na, nb = np.isnan(a), np.isnan(b)
        return np.array_equal(na, nb) and np.array_equal(a[~na].view(np.uint32), b[~nb].view(np.uint32))

    for shape in ((6, 8, 14, 14), (5, 12, 7, 7), (12, 3, 56, 56), (64, 5, 28, 28)):
        x = torch.randn(shape, generator=gen)
        x[1, 2, 3, 4] = float('nan')
        x[0, 0, 0, 0] = float('inf')
        x[2, 1, 1, 1] = float('-inf')
        for half in (False, True):
            ref = O.act_per_channel_qdq(x, 4, half_range=half)
            y = ops.minmax_qdq_resident(x.cuda(), shape[0], shape[1], shape[2] * shape[3], 4, half)
            assert same(y.cpu(), ref), (shape, half)
            assert bool(torch.isnan(y[:, 2]).all()) and not bool(torch.isnan(y[:, 3:]).any())


def test_resident_replays_from_a_graph(ops):
    """No workspace, no state: 20 graph replays on changing data give the chain's bits."""
    torch.manual_seed(7)
    x = torch.randn(64, 128, 28, 28, device='cuda') * 2
    assert describe(64, 128, 784)[0] == 0
    y = torch.empty_like(x)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.minmax_qdq_resident(x, 64, 128, 784, 4, False, out=y)
        graph = torch.cuda.CUDAGraph()
        y.zero_()
        with torch.cuda.graph(graph, stream=side):
            ops.minmax_qdq_resident(x, 64, 128, 784, 4, False, out=y)
        for i in range(20):
            x.mul_(1.01)
            graph.replay()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref, _ = classic(ops, x, 4, False)
    assert torch.equal(y, ref)


@pytest.mark.parametrize('shape,half', [((64, 512, 28, 28), False), ((64, 1024, 14, 14), False), ((64, 2048, 7, 7), False),
                                        ((64, 256, 14, 14), True), ((64, 512, 7, 7), True), ((64, 128, 28, 28), True)])
def test_resident_batch64_shard_layers(ops, shape, half):
    """The ResNet-50 layers of the batch-64 shard (512 / 8 GPUs) that run resident: properties that need no
    oracle, plus equality with the three-launch chain."""
    from cnn_quantization_amd import _lib as L
    N, C, H, W = shape
    torch.manual_seed(12345)
    x = torch.empty(shape, device='cuda').normal_()
    x.mul_(torch.rand(1, C, 1, 1, device='cuda') * 3 + 0.1)
    y, parts = ops.minmax_qdq_resident(x, N, C, H * W, 4, half, want_parts=True)
    st, qp = parts['stats'], parts['qp']
    assert torch.equal(st[L.STAT_MAX], x.amax(dim=(0, 2, 3)))
    assert torch.equal(st[L.STAT_MIN], x.amin(dim=(0, 2, 3)))
    sc, zp = qp[0].view(1, C, 1, 1), qp[1].view(1, C, 1, 1)
    codes = torch.round(y / sc + zp)
    assert float(codes.min()) >= 0 and float(codes.max()) <= 15
    assert torch.equal((codes - zp) * sc, y)
    y2 = ops.pc_qdq(y, N, C, H * W, qp)            # idempotence under the same parameters
    assert torch.equal(y2, y)
    yc, pc = classic(ops, x, 4, half)
    assert torch.equal(pc['qp'], qp)
    assert torch.equal(yc, y)
