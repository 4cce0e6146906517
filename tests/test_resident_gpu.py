"""The register-resident single-launch form of config 2 (cnnq_pc_minmax_qdq_resident, csrc/cnnq_resident.hip.h):
same bits as the reference-pinned golden vectors and as the three-launch chain, on every tile shape, with the
workgroup exchange exercised (groups of 2 .. 208 workgroups), re-armed across launches, replayed from a HIP graph,
and with the bounded-wait fallback forced.  Needs an MI355X: `pytest -m gpu`."""
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
    return rc, dict(zip(('A', 'K', 'mode', 'S', 'ncb', 'Gs', 'groups', 'wgs'), list(out)))


def classic(ops, x, bits, half):
    """The three-launch chain (statistics pass, parameters, Q/DQ pass): want_codes keeps it off the resident path."""
    N, C = x.shape[:2]
    y, codes, parts = ops.minmax_qdq_fused(x, N, C, x[0, 0].numel(), bits, half, want_codes=True, want_parts=True)
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
        for flags in (0, 1):
            res = ops.minmax_qdq_resident(x, N, C, x[0, 0].numel(), bits, half, want_parts=True, flags=flags)
            if describe(N, C, x[0, 0].numel())[0] != 0:      # [3,16,5,9]: H*W = 45, no resident kernel
                assert res is None
                break
            assert res is not None, key
            y, parts = res
            assert bits_equal(y.cpu(), g.np(key + '_y')), (key, flags)
            assert bits_equal(parts['stats'][L.STAT_MAX].cpu(), g.np('s%s_stat_max' % si)), key
            if not half:
                assert bits_equal(parts['stats'][L.STAT_MIN].cpu(), g.np('s%s_stat_min' % si)), key
            yc, pc = classic(ops, x, bits, half)
            assert torch.equal(parts['qp'], pc['qp']), key
        else:
            n += 1
    assert n == 12


SHAPES = [
    (3, 5, 7, 7),        # straddling float4s, one column block
    (70, 40, 7, 7),      # straddle, several column blocks and batch splits
    (2, 8, 14, 14), (37, 24, 14, 14), (64, 256, 14, 14),
    (5, 3, 28, 28), (33, 16, 28, 28),
    (9, 4, 56, 56),      # a channel row wider than a workgroup: column slices x batch splits
    (66, 3, 112, 112), (130, 2, 40, 36),
    (1, 1100, 2, 2), (4, 260, 1, 4), (2, 3, 64, 80), (1, 64, 3, 3 * 4), (600, 2, 8, 8),
]


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('half', [False, True])
def test_resident_equals_chain_and_oracle(ops, shape, half):
    gen = torch.Generator().manual_seed(sum(shape) + int(half))
    N, C, H, W = shape
    x = torch.randn(shape, generator=gen) * (torch.rand(1, C, 1, 1, generator=gen) * 4 + 0.05) + \
        torch.randn(1, C, 1, 1, generator=gen)
    rc, d = describe(N, C, H * W)
    assert rc == 0
    xd = x.cuda()
    ref = O.act_per_channel_qdq(x, 4, half_range=half)
    for flags in (0, 1, 0):
        y = ops.minmax_qdq_resident(xd, N, C, H * W, 4, half, flags=flags)
        assert y is not None
        assert bits_equal(y.cpu(), ref), (shape, d, flags)
    yc, _ = classic(ops, xd, 4, half)
    assert torch.equal(yc, y)
    # the forced recompute path reports itself; nothing else may have timed out
    st = ops.resident_status(xd)
    assert st in (0, 1)


def test_resident_unsupported_shapes_take_the_chain(ops):
    """Unaligned base pointers and H*W that neither is a multiple of 4 nor straddles cleanly: no resident kernel,
    act_qdq_per_channel still answers (three-launch chain) with the oracle's bits."""
    x = torch.randn(3, 5, 5, 9)
    assert ops.minmax_qdq_resident(x.cuda(), 3, 5, 45, 4) is None
    buf = torch.empty(2 * 8 * 16 + 1, device='cuda')
    xu = buf[1:].view(2, 8, 4, 4)
    xu.copy_(torch.randn(2, 8, 4, 4))
    assert ops.minmax_qdq_resident(xu, 2, 8, 16, 4) is None
    for t in (x.cuda(), xu):
        assert bits_equal(ops.act_qdq_per_channel(t, 4).cpu(), O.act_per_channel_qdq(t.cpu(), 4))


def test_resident_nan_inf_follow_torch(ops):
    """torch.min / torch.max propagate NaN (iq.py:416,423): a NaN poisons exactly its channel; +-inf give an
    infinite range.  Same bits as the oracle (the reference's own op chain on CPU)."""
    gen = torch.Generator().manual_seed(3)
    for shape in ((6, 8, 14, 14), (5, 12, 7, 7), (40, 3, 56, 56)):
        x = torch.randn(shape, generator=gen)
        x[1, 2, 3, 4] = float('nan')
        x[0, 0, 0, 0] = float('inf')
        x[2, 1, 1, 1] = float('-inf')
        for half in (False, True):
            ref = O.act_per_channel_qdq(x, 4, half_range=half)
            y = ops.minmax_qdq_resident(x.cuda(), shape[0], shape[1], shape[2] * shape[3], 4, half)
            assert bits_equal(y.cpu(), ref), (shape, half)
            assert bool(torch.isnan(y[:, 2]).all()) and not bool(torch.isnan(y[:, 3:]).any())


def test_resident_rearms_and_replays_from_a_graph(ops):
    """The exchange workspace is zeroed once; 50 launches in a row and 20 graph replays all give the same bits."""
    torch.manual_seed(7)
    x = torch.randn(64, 128, 28, 28, device='cuda') * 2
    rc, d = describe(64, 128, 784)
    assert rc == 0 and d['Gs'] > 1
    ref, _ = classic(ops, x, 4, False)
    y = torch.empty_like(x)
    for _ in range(50):
        ops.minmax_qdq_resident(x, 64, 128, 784, 4, False, out=y)
    assert torch.equal(y, ref)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.minmax_qdq_resident(x, 64, 128, 784, 4, False, out=y)      # allocates this stream's workspace
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
    assert ops.resident_status(x) == 0


@pytest.mark.parametrize('shape,half', [((512, 64, 112, 112), True), ((512, 256, 56, 56), False),
                                        ((512, 2048, 7, 7), False), ((64, 64, 112, 112), True)])
def test_resident_full_size_properties(ops, shape, half):
    """BASELINE-sized layers (1.64 GB: groups of 208 / 64 workgroups, 13-16 K workgroups per launch): properties
    that need no oracle, plus equality with the three-launch chain."""
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
    del codes
    y2 = ops.pc_qdq(y, N, C, H * W, qp)            # idempotence under the same parameters
    assert torch.equal(y2, y)
    del y2
    yc, pc = classic(ops, x, 4, half)
    assert torch.equal(pc['qp'], qp)
    assert torch.equal(yc, y)
    assert ops.resident_status(x) == 0
