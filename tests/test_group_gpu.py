"""The group-exchange single-launch form of config 2 (cnnq_pc_minmax_qdq_group, csrc/cnnq_group.hip.h): same bits
as the reference-pinned golden vectors, the oracle and the three-launch chain on every tile shape, with the workgroup
exchange exercised - groups of 2 .. 208 workgroups, one- and two-level arrival counters, a workspace zeroed ONCE and
shared by launches of different geometry, rotating inputs (no launch may see the previous launch's extrema), HIP-graph
replay, the bounded-wait fallback forced, NaN / inf semantics, and the BASELINE-sized 1.64 GB layers.
Needs an MI355X: `pytest -m gpu`."""
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
    rc = _lib.load().cnnq_pc_group_describe(N, C, HW, out)
    return rc, dict(zip(('A', 'K', 'mode', 'S', 'ncb', 'Gs', 'groups', 'wgs'), list(out)))


def classic(ops, x, bits, half):
    """The three-launch chain: chain=True keeps it off the single-launch paths."""
    N, C = x.shape[:2]
    y, codes, parts = ops.minmax_qdq_fused(x, N, C, x[0, 0].numel(), bits, half, want_codes=True, want_parts=True, chain=True)
    return y, parts


def test_group_golden_bit_exact(ops, golden):
    """The config-2 cases recorded from the reference (12 with a float4 layout): dequantized floats, min / max,
    scale / zero point bit for bit - with the exchange, and with the recompute path forced."""
    from cnn_quantization_amd import _lib as L
    g = golden('act_pc')
    n = 0
    ops.group_status(torch.empty(1, device='cuda'), clear=True)
    for key in g.np('names'):
        key = str(key)
        name, si = key.rsplit('_s', 1)
        if not name.startswith('cfg2') or 'baa' in name:
            continue
        x = g.t('x' + si).cuda()
        N, C = x.shape[:2]
        bits, half = int(g.np(key + '_bits')), bool(g.np(key + '_half'))
        if describe(N, C, x[0, 0].numel())[0] != 0:      # [3,16,5,9]: H*W = 45
            assert ops.minmax_qdq_group(x, N, C, x[0, 0].numel(), bits, half) is None
            continue
        for flags in (0, 1):
            y, parts = ops.minmax_qdq_group(x, N, C, x[0, 0].numel(), bits, half, want_parts=True, flags=flags)
            assert bits_equal(y.cpu(), g.np(key + '_y')), (key, flags)
            assert bits_equal(parts['stats'][L.STAT_MAX].cpu(), g.np('s%s_stat_max' % si)), key
            assert bits_equal(parts['stats'][L.STAT_MIN].cpu(), g.np('s%s_stat_min' % si)), key
        n += 1
    assert n == 12
    # the forced recompute reported itself (bit 1), no wait expired (bit 0)
    assert ops.group_status(x, clear=True) == ops.GROUP_TEST_HOOK


SHAPES = [
    # flat tiles (k_mmq_flat: a channel row that is not a multiple of 256 float4): ragged last tiles, rows shorter and
    # longer than a workgroup, one- and two-level arrival
    (40, 6, 56, 56), (300, 3, 56, 56), (130, 4, 28, 28), (512, 2, 28, 28), (20, 3, 112, 112), (33, 5, 24, 24),
    (17, 3, 40, 52),
    (3, 8, 7, 7), (70, 40, 7, 7), (300, 24, 7, 7),          # straddling float4s; several column blocks and batch splits
    (2, 8, 14, 14), (37, 24, 14, 14), (64, 256, 14, 14), (200, 12, 14, 14),
    (5, 3, 28, 28), (33, 16, 28, 28), (130, 20, 28, 28),
    (9, 4, 56, 56), (70, 6, 56, 56),                         # a channel row wider than a workgroup: slices x splits
    (66, 3, 112, 112), (260, 2, 112, 112),                   # two-level arrival (more than 16 members per group)
    (130, 2, 40, 36), (1, 1100, 2, 2), (4, 260, 1, 4), (2, 3, 64, 80), (600, 2, 8, 8),
]


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('half', [False, True])
def test_group_equals_chain_and_oracle(ops, shape, half):
    gen = torch.Generator().manual_seed(sum(shape) + int(half))
    N, C, H, W = shape
    rc, d = describe(N, C, H * W)
    assert rc == 0
    yc = None
    ops.group_status(torch.empty(1, device='cuda'), clear=True)
    for rnd in range(3):          # a different tensor every launch, the same (never re-zeroed) workspace
        x = torch.randn(shape, generator=gen) * (torch.rand(1, C, 1, 1, generator=gen) * 4 + 0.05) + \
            torch.randn(1, C, 1, 1, generator=gen)
        xd = x.cuda()
        y, parts = ops.minmax_qdq_group(xd, N, C, H * W, 4, half, want_parts=True, flags=1 if rnd == 1 else 0)
        assert torch.equal(parts['stats'][0], xd.amin(dim=(0, 2, 3))) and torch.equal(parts['stats'][1], xd.amax(dim=(0, 2, 3)))
        yc, pc = classic(ops, xd, 4, half)
        assert torch.equal(yc, y) and torch.equal(pc['qp'], parts['qp']), (shape, d, rnd)
    assert bits_equal(y.cpu(), O.act_per_channel_qdq(x, 4, half_range=half)), (shape, d)
    # the forced recompute of round 1 reported itself (bit 1); no wait ever expired (bit 0)
    assert ops.group_status(xd, clear=True) == ops.GROUP_TEST_HOOK


def test_group_plans_cover_one_and_two_level_arrival():
    """mode 3 = flat tiles (k_mmq_flat: 56x56 / 28x28 / 112x112), 1 / 2 = row pieces / whole channels (k_mmq_group)"""
    levels = set()
    for shape in SHAPES + [(512, 64, 112, 112), (512, 256, 56, 56), (512, 2048, 7, 7)]:
        rc, d = describe(shape[0], shape[1], shape[2] * shape[3])
        assert rc == 0
        levels.add((d['mode'], 1 if d['Gs'] <= 16 else 2))
        if d['mode'] == 3:
            assert (d['Gs'] - 1) * 256 * d['K'] < shape[0] * shape[2] * shape[3] // 4 <= d['Gs'] * 256 * d['K']
        else:
            assert d['S'] * d['K'] >= shape[0]
        assert d['Gs'] <= 512
    assert levels >= {(3, 1), (3, 2), (2, 1), (2, 2)}, levels
    assert describe(512, 256, 3136)[1]['mode'] == 3 and describe(512, 1024, 196)[1]['mode'] == 2
    assert describe(512, 64, 112 * 112)[1]['Gs'] == 196          # 512 * 3136 float4 / (256 * 32)


def test_group_nan_inf_follow_torch(ops):
    """torch.min / torch.max propagate NaN (iq.py:416,423): a NaN poisons exactly its channel, also through the
    exchange; +-inf give an infinite range.  NaN positions and all other bits as the oracle."""
    def same(a, b):
        a, b = a.numpy(), b.numpy()
        na, nb = np.isnan(a), np.isnan(b)
        return np.array_equal(na, nb) and np.array_equal(a[~na].view(np.uint32), b[~nb].view(np.uint32))

    gen = torch.Generator().manual_seed(3)
    for shape in ((70, 8, 14, 14), (40, 12, 7, 7), (40, 3, 56, 56), (130, 5, 28, 28)):
        x = torch.randn(shape, generator=gen)
        x[1, 2, 3, 4] = float('nan')
        x[0, 0, 0, 0] = float('inf')
        x[2, 1, 1, 1] = float('-inf')
        for half in (False, True):
            ref = O.act_per_channel_qdq(x, 4, half_range=half)
            y = ops.minmax_qdq_group(x.cuda(), shape[0], shape[1], shape[2] * shape[3], 4, half)
            assert same(y.cpu(), ref), (shape, half)
            assert bool(torch.isnan(y[:, 2]).all()) and not bool(torch.isnan(y[:, 3:]).any())


def test_group_rearms_and_replays_from_a_graph(ops):
    """The exchange workspace is zeroed once; 50 launches in a row and 20 graph replays give the chain's bits."""
    torch.manual_seed(7)
    x = torch.randn(64, 128, 56, 56, device='cuda') * 2
    rc, d = describe(64, 128, 3136)
    assert rc == 0 and d['Gs'] > 1
    y = torch.empty_like(x)
    for _ in range(50):
        x.mul_(1.001)
        ops.minmax_qdq_group(x, 64, 128, 3136, 4, False, out=y)
    ref, _ = classic(ops, x, 4, False)
    assert torch.equal(y, ref)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.minmax_qdq_group(x, 64, 128, 3136, 4, False, out=y)      # allocates this stream's workspace
        graph = torch.cuda.CUDAGraph()
        y.zero_()
        with torch.cuda.graph(graph, stream=side):
            ops.minmax_qdq_group(x, 64, 128, 3136, 4, False, out=y)
        for i in range(20):
            x.mul_(1.01)
            graph.replay()
        st = ops.group_status(x)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref, _ = classic(ops, x, 4, False)
    assert torch.equal(y, ref)
    assert st == 0 and ops.group_status(x) == 0


@pytest.mark.parametrize('slots', [1, 0])
def test_workspace_is_zero_at_rest(slots):
    """Counters and slots are zero whenever no launch is in flight - after launches of every tile shape on ONE workspace,
    normal and with the recompute path forced - so no launch can meet another launch's arrivals: with the slot meeting
    (round 4) a stale non-zero slot WOULD be taken for a member's pair.  Runs in a process of its own (a workspace of its
    own, nothing else in flight); flags bit 5 selects the counter meeting of round 2."""
    import os
    import subprocess
    import sys
    code = r"""
import ctypes, sys, torch
sys.path.insert(0, %r)
from cnn_quantization_amd import _lib
lib = _lib.load()
ws = ctypes.c_void_p()
_lib.check(lib.cnnq_group_ws_alloc(18 << 20, ctypes.byref(ws)), 'alloc')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
MEET = %d
shapes = [(40, 6, 56, 56), (300, 3, 56, 56), (66, 3, 112, 112), (260, 2, 112, 112), (70, 40, 7, 7), (64, 256, 14, 14),
          (5, 3, 28, 28), (130, 20, 28, 28), (1, 1100, 2, 2), (600, 2, 8, 8), (64, 512, 7, 7), (512, 16, 56, 56)]
n = ctypes.c_uint64(0)
for flags in (0 | MEET, 1 | MEET, 0 | MEET):
    for (N, C, H, W) in shapes:
        x = torch.randn(N, C, H, W, device='cuda') * 3
        y = torch.empty_like(x)
        qp = torch.empty(3, C, device='cuda')
        mm = torch.empty(2, C, device='cuda')
        rc = lib.cnnq_pc_minmax_qdq_group(x.data_ptr(), y.data_ptr(), N, C, H * W, 4, 0, ws, qp.data_ptr(), mm.data_ptr(), flags, st)
        assert rc == 0, (rc, N, C, H, W)
        _lib.check(lib.cnnq_group_ws_at_rest(ws, ctypes.byref(n)), 'at_rest')
        assert n.value == 0, ('words not zero at rest', n.value, (N, C, H, W), flags)
        assert torch.equal(mm[0], x.amin(dim=(0, 2, 3))) and torch.equal(mm[1], x.amax(dim=(0, 2, 3))), (N, C, H, W)
s = ctypes.c_uint32(0)
_lib.check(lib.cnnq_group_ws_status(ws, ctypes.byref(s)), 'status')
assert s.value == 2, s.value          # the test hook reported itself; no wait expired
print('ok')
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 0 if slots else 32)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stderr[-3000:]


@pytest.mark.parametrize('shape,half', [((512, 64, 112, 112), True), ((512, 256, 56, 56), False),
                                        ((512, 2048, 7, 7), False), ((64, 64, 112, 112), True),
                                        ((512, 512, 28, 28), False), ((64, 64, 56, 56), True),
                                        ((64, 1024, 14, 14), False)])
def test_group_full_size_properties(ops, shape, half):
    """BASELINE-sized layers (1.64 GB: groups of 208 / 64 workgroups, 13-16 K workgroups per launch): properties
    that need no oracle, plus equality with the three-launch chain - twice, on different data."""
    from cnn_quantization_amd import _lib as L
    N, C, H, W = shape
    torch.manual_seed(12345)
    x = torch.empty(shape, device='cuda').normal_()
    ops.group_status(x, clear=True)
    for rnd in range(2):
        x.mul_(torch.rand(1, C, 1, 1, device='cuda') * 3 + 0.1)
        y, parts = ops.minmax_qdq_group(x, N, C, H * W, 4, half, want_parts=True)
        st, qp = parts['stats'], parts['qp']
        assert torch.equal(st[L.STAT_MAX], x.amax(dim=(0, 2, 3)))
        assert torch.equal(st[L.STAT_MIN], x.amin(dim=(0, 2, 3)))
        sc, zp = qp[0].view(1, C, 1, 1), qp[1].view(1, C, 1, 1)
        codes = torch.round(y / sc + zp)
        assert float(codes.min()) >= 0 and float(codes.max()) <= 15
        assert torch.equal((codes - zp) * sc, y)
        del codes
        yc, pc = classic(ops, x, 4, half)
        assert torch.equal(pc['qp'], qp)
        assert torch.equal(yc, y)
        del yc, y
    assert ops.group_status(x) == 0          # no wait expired on an idle GPU
