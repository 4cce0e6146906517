"""Codes, code histogram / entropy and packed 4-bit codes straight out of the single-launch kernels
(cnnq_pc_minmax_qdq_single: k_mmq_whole / k_mmq_group / k_mmq_flat with OUT = 1 / 2; VERDICT r2 missing #2, #3):
  * golden: the reference's integer codes and their Shannon entropy (act_pc.npz `*_codes`, `*_entropy`;
    iq.py:586-587, utils/entropy.py:6-17) bit for bit / to fp32 rounding, with the single launch proven to have run;
  * every tile shape: codes and y equal the three-launch chain's, the entropy equals -sum p log2 p of the very
    codes returned, the replica histogram is zero again afterwards (the next call starts clean);
  * packed storage: dequantize_pack4(minmax_quantize_pack4(x)) == act_qdq_per_channel(x) bit for bit, and the packed
    bytes equal those of the separate quantize+pack pass fed with the same parameters;
  * BASELINE-sized layers.
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


def entropy_of(codes):
    _, counts = torch.unique(codes.flatten(), return_counts=True)
    p = counts.double() / codes.numel()
    return float(-(p * torch.log2(p)).sum())


def test_golden_codes_and_entropy_from_the_single_launch(ops, golden):
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
        res = ops.minmax_qdq_single(x, N, C, x[0, 0].numel(), bits, half, want_codes=True, want_entropy=True)
        if res is None:                       # [3,16,5,9]: H*W = 45 has no single-launch kernel
            assert x[0, 0].numel() % 4 != 0 and (C * x[0, 0].numel()) % 4 != 0 or x[0, 0].numel() == 45
            continue
        y, codes, ent = res
        assert bits_equal(y.cpu(), g.np(key + '_y')), key
        assert np.array_equal(codes.cpu().numpy().astype(np.int32), g.np(key + '_codes')), key
        ref = float(g.np(key + '_entropy'))
        assert abs(float(ent) - ref) <= 2e-5 * max(1., abs(ref)), (key, float(ent), ref)
        n += 1
    assert n >= 11


SHAPES = [
    (40, 6, 56, 56), (300, 3, 56, 56), (130, 4, 28, 28), (20, 3, 112, 112), (17, 3, 40, 52),       # flat tiles
    (70, 40, 7, 7), (300, 24, 7, 7), (37, 24, 14, 14), (200, 12, 14, 14), (64, 256, 14, 14),          # group (A = 4, 1)
    (8, 32, 14, 14), (8, 64, 7, 7), (4, 16, 28, 28), (64, 40, 7, 7),                                   # whole channels
    (130, 2, 40, 36), (600, 2, 8, 8),
    (40, 3, 4, 131), (64, 6, 28, 28), (96, 5, 56, 56),      # flat tiles: rows of odd / 4-float4 / 8-float4 groups (packed store widths)
]


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('bits,half', [(4, False), (4, True), (8, False), (2, True)])
def test_single_equals_chain(ops, shape, bits, half):
    gen = torch.Generator().manual_seed(sum(shape) + bits + int(half))
    N, C, H, W = shape
    x = (torch.randn(shape, generator=gen) * (torch.rand(1, C, 1, 1, generator=gen) * 4 + 0.05)
         + torch.randn(1, C, 1, 1, generator=gen)).cuda()
    yc, cc, pc = ops.minmax_qdq_fused(x, N, C, H * W, bits, half, want_codes=True, want_parts=True, chain=True)
    for rnd in range(2):                      # twice: the histogram tables must come back zero
        res = ops.minmax_qdq_single(x, N, C, H * W, bits, half, want_codes=True, want_entropy=True, want_parts=True)
        assert res is not None, shape
        y, codes, ent, parts = res
        assert torch.equal(y, yc) and torch.equal(codes, cc) and torch.equal(parts['qp'], pc['qp'])
        ref = entropy_of(codes)
        assert abs(float(ent) - ref) <= 2e-5 * max(1., ref), (shape, float(ent), ref)
    st = ops._raw_stream(x.device.index)
    assert int(ops._hist_replicas(x, st).abs().sum()) == 0
    # entropy alone, codes alone
    y2, ent2 = ops.minmax_qdq_single(x, N, C, H * W, bits, half, want_entropy=True)
    assert torch.equal(y2, yc) and float(ent2) == float(ent)
    y3, codes3 = ops.minmax_qdq_single(x, N, C, H * W, bits, half, want_codes=True)
    assert torch.equal(y3, yc) and torch.equal(codes3, cc)
    assert ops.group_status(x) == 0


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('bits,half', [(4, False), (4, True), (3, False)])
def test_packed_from_the_single_launch(ops, shape, bits, half):
    gen = torch.Generator().manual_seed(sum(shape) * 3 + bits + int(half))
    N, C, H, W = shape
    x = (torch.randn(shape, generator=gen) * (torch.rand(1, C, 1, 1, generator=gen) * 2 + 0.1)).cuda()
    res = ops.minmax_quantize_pack4(x, bits, half)
    assert res is not None, shape
    packed, qp = res
    y = ops.act_qdq_per_channel(x, bits, positive=half)
    # decode on the torch side: even element in the low nibble; (code - zp) * scale (iq.py:591-592)
    nib = torch.stack([packed & 15, packed >> 4], dim=1).flatten().view(shape).float()
    assert torch.equal((nib - qp[1].view(1, C, 1, 1)) * qp[0].view(1, C, 1, 1), y)
    _, parts = ops.minmax_qdq_fused(x, N, C, H * W, bits, half, want_parts=True, chain=True)
    assert torch.equal(parts['qp'], qp)
    if (H * W) % 4 == 0:
        # the product's own decoder, and the bytes of the separate quantize+pack pass with the same parameters
        assert torch.equal(ops.dequantize_pack4(packed, x.shape, qp), y)
        assert torch.equal(ops.quantize_pack4(x, qp), packed)
    # a packed buffer that is only 2-byte aligned (the ABI's requirement): the narrow stores, the same bytes
    big = torch.zeros(packed.numel() + 16, dtype=torch.uint8, device='cuda')
    p2, _ = ops.minmax_quantize_pack4(x, bits, half, out=big[2:2 + packed.numel()])
    assert torch.equal(p2, packed) and int(big[:2].sum()) == 0 and int(big[2 + packed.numel():].sum()) == 0
    with pytest.raises(Exception):
        ops.minmax_quantize_pack4(x, 8, half)


@pytest.mark.parametrize('shape,half', [((512, 256, 56, 56), False), ((512, 512, 28, 28), True), ((512, 2048, 7, 7), False),
                                        ((512, 1024, 14, 14), False)])
def test_single_outputs_full_size(ops, shape, half):
    """b512 layers: codes within [0, 15] and consistent with y, histogram total == numel (through the entropy of a
    two-valued tensor being what it must), entropy == chunked count of the returned codes, packed round trip."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    N, C, H, W = shape
    x = bench.laplace_activation(shape, 99, torch.device('cuda'))
    if half:
        x.clamp_(min=-0.1)
    y, codes, ent, parts = ops.minmax_qdq_single(x, N, C, H * W, 4, half, want_codes=True, want_entropy=True, want_parts=True)
    qp = parts['qp']
    sc, zp = qp[0].view(1, C, 1, 1), qp[1].view(1, C, 1, 1)
    assert int(codes.max()) <= 15
    assert torch.equal((codes.float() - zp) * sc, y)
    assert torch.equal(y, ops.act_qdq_per_channel(x, 4, positive=half))
    counts = torch.zeros(16, dtype=torch.int64, device='cuda')
    for n0 in range(0, N, 64):
        counts += torch.bincount(codes[n0:n0 + 64].flatten().long(), minlength=16)
    assert int(counts.sum()) == x.numel()
    p = counts[counts > 0].double() / x.numel()
    ref = float(-(p * torch.log2(p)).sum())
    assert abs(float(ent) - ref) <= 2e-5 * max(1., ref), (float(ent), ref)
    del codes
    packed, qp2 = ops.minmax_quantize_pack4(x, 4, half)
    assert torch.equal(qp2, qp)
    if (H * W) % 4 == 0:
        assert torch.equal(ops.dequantize_pack4(packed, x.shape, qp2), y)
    else:
        for n0 in range(0, N, 64):
            pk = packed[n0 * C * H * W // 2:(n0 + 64) * C * H * W // 2]
            nib = torch.stack([pk & 15, pk >> 4], dim=1).flatten().view(-1, C, H, W).float()
            assert torch.equal((nib - zp) * sc, y[n0:n0 + 64])
    assert ops.group_status(x) == 0


def test_out_must_not_overlap_the_input(ops):
    """`out` overlapping x is refused (the kernels' pointers are __restrict__, the cold path of the exchange re-reads
    x); two views of one arena that do not overlap are accepted."""
    arena = torch.randn(2 * 8 * 16 * 14 * 14 + 64, device='cuda')
    n = 8 * 16 * 14 * 14
    x = arena[:n].view(8, 16, 14, 14)
    with pytest.raises(Exception):
        ops.act_qdq_per_channel(x, 4, out=x)
    with pytest.raises(Exception):
        ops.act_qdq_per_channel(x, 4, out=arena[16:16 + n].view(8, 16, 14, 14))
    y = ops.act_qdq_per_channel(x, 4, out=arena[n:2 * n].view(8, 16, 14, 14))
    assert torch.equal(y, ops.act_qdq_per_channel(x.clone(), 4))
