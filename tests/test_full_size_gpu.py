"""BASELINE-sized correctness of configs 1, 3 and 5 (only configs 2 and 4 ran at b512 size in the
GPU suite).  At these sizes the oracle does not finish in seconds, so each test checks (a) the small per-channel
tables against the oracle's own functions fed with the device's statistics (exact: iq.py's parameter arithmetic is a
deterministic function of those vectors), (b) the statistics against fp64 torch reductions, and (c) size-independent
properties of the element-wise result: codes inside their range, y == (code - zp) * scale bit for bit, histogram
totals, entropy against a chunked count of the very codes the kernel stored.  Needs an MI355X: `pytest -m gpu`."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import quant_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from cnn_quantization_amd import ops as _ops
    return _ops


def channel_stats_f64(x, chunk=32):
    """mean, unbiased std, b = E|x - mean| per channel of NCHW x, accumulated in fp64 (channels in chunks)."""
    N, C = x.shape[:2]
    mean = torch.empty(C, dtype=torch.float64, device=x.device)
    std = torch.empty_like(mean)
    b = torch.empty_like(mean)
    for c0 in range(0, C, chunk):
        t = x[:, c0:c0 + chunk].double().transpose(0, 1).reshape(min(chunk, C - c0), -1)
        m = t.mean(1)
        mean[c0:c0 + chunk] = m
        std[c0:c0 + chunk] = t.std(1, unbiased=True)
        b[c0:c0 + chunk] = (t - m[:, None]).abs().mean(1)
        del t
    return mean, std, b


@pytest.mark.parametrize('shape,half', [((512, 256, 56, 56), False), ((512, 128, 28, 28), True)])
def test_config3_aciq_bit_allocation_full_size(ops, shape, half):
    """-c laplace -baa at b512 (iq.py:327-352, 381-407, 557-603)."""
    from cnn_quantization_amd import _lib as L
    N, C = shape[:2]
    x = bench.laplace_activation(shape, 4242, torch.device('cuda'))
    y, codes, parts = ops.act_qdq_per_channel(x, 4, positive=half, clip='laplace', bit_alloc=True, want_codes=True,
                                              want_parts=True)
    st, qp, diag = parts['stats'], parts['qp'], parts['diag']
    # (b) statistics: extrema exact, moments to fp64-vs-fp32 tolerance
    assert torch.equal(st[L.STAT_MAX], x.amax(dim=(0, 2, 3))) and torch.equal(st[L.STAT_MIN], x.amin(dim=(0, 2, 3)))
    mean, std, b = channel_stats_f64(x)
    for row, ref in ((L.STAT_MEAN, mean), (L.STAT_STD, std), (L.STAT_B, b)):
        err = ((st[row].double() - ref).abs() / ref.abs().clamp(min=1e-3)).max()
        assert float(err) < 5e-6, (row, float(err))
    # (a) bit allocation = the reference's iteration on the device's own prior (std: bit_alloc_prior 'gaus')
    bits = diag[L.DIAG_BITS].cpu()
    assert torch.equal(bits, O.bits_alloc_fixed_target(st[L.STAT_STD].cpu(), 4, True))
    assert float(bits.min()) >= 0 and float(bits.max()) <= 8 and torch.equal(bits, bits.round())
    assert abs(float(bits.mean()) - 4.0) <= 0.011            # iq.py:403: the loop stops within 0.01 of the target
    # (a) alpha, delta / offset, scale / zero point / qmax from the oracle's functions on the device's statistics
    alpha = O.alpha_laplace(st[L.STAT_B].cpu(), 4, half, bits)
    delta, offset = O.alpha_to_delta_offset(alpha, st[L.STAT_MAX].cpu(), st[L.STAT_MIN].cpu(), st[L.STAT_MEAN].cpu(), half)
    delta, offset = torch.as_tensor(delta, dtype=torch.float32), torch.as_tensor(offset, dtype=torch.float32)
    max_ = offset + delta                                    # iq.py:351 then :443 (two fp32 roundings)
    _, _, scale, zp, qmax = O.qdq_core(torch.zeros(C, 1), max_ - offset, offset, bit_alloc=bits, return_parts=True)
    assert torch.equal(qp[L.QP_SCALE].cpu(), scale) and torch.equal(qp[L.QP_ZP].cpu(), zp)
    assert torch.equal(qp[L.QP_QMAX].cpu(), qmax)
    # (c) element-wise properties
    sc, z, qm = (qp[r].view(1, C, 1, 1) for r in (L.QP_SCALE, L.QP_ZP, L.QP_QMAX))
    cf = codes.float()
    assert bool((cf <= qm).all()) and int(codes.min()) >= 0
    assert torch.equal((cf - z) * sc, y)
    # the codes are the quantization of x: re-derive them with torch's own ops (IEEE divide, add, clamp, round)
    q = torch.round(torch.minimum((x / sc + z), qm).clamp_(min=0.))
    assert torch.equal(q, cf)


def test_config5_midtread_entropy_full_size(ops):
    """VGG-16's largest layer at b512, [512, 64, 224, 224] = 6.6 GB (-mtq -me; iq.py:185-225, utils/entropy.py)."""
    from cnn_quantization_amd import _lib as L
    shape = (512, 64, 224, 224)
    N, C = shape[:2]
    x = bench.laplace_activation(shape, 777, torch.device('cuda')).clamp_(min=0)      # fused-ReLU archs: force_positive
    y, ent, codes, parts = ops.mid_tread_qdq(x, 4, clip=True, sym=False, want_entropy=True, want_codes=True,
                                             want_parts=True)
    mt, hist = parts['mt'], parts['hist']
    delta, cmin, cmax = (mt[r].view(1, C, 1, 1) for r in (L.MT_DELTA, L.MT_CMIN, L.MT_CMAX))
    assert bool((codes >= cmin).all()) and bool((codes <= cmax).all())
    assert torch.equal(codes * delta, y)
    # codes are round(x / Delta) clamped (iq.py:196-215), bit for bit
    q = torch.maximum(torch.minimum(torch.round(x / delta), cmax), cmin)
    assert torch.equal(q, codes)
    del q, y
    assert int(hist[:-1].sum()) == x.numel()                 # every element counted exactly once
    # entropy of the stored codes, counted in chunks: integers by bincount, the few non-integer clamp values by unique
    top = int(codes.max().item()) + 2
    counts = torch.zeros(top, dtype=torch.int64, device='cuda')
    odd = {}
    for n0 in range(0, N, 16):
        c = codes[n0:n0 + 16].flatten()
        isint = c == c.round()
        counts += torch.bincount(c[isint].long(), minlength=top)
        vals, cnt = torch.unique(c[~isint], return_counts=True)
        for v, k in zip(vals.tolist(), cnt.tolist()):
            odd[v] = odd.get(v, 0) + k
        del c, isint
    allc = torch.cat([counts[counts > 0].double(), torch.tensor(list(odd.values()), dtype=torch.float64, device='cuda')])
    assert int(allc.sum()) == x.numel()
    p = allc / x.numel()
    ref = float(-(p * torch.log2(p)).sum())
    assert abs(float(ent) - ref) < 2e-4 * max(1.0, ref), (float(ent), ref)


def test_config1_per_tensor_full_size(ops):
    """BASELINE config 1 on exactly [32, 64, 112, 112] against the oracle on the host (it fits: 103 MB):
    iq.py:361-379 + kernels/gemmlowp.cu:8-45."""
    x = bench.laplace_activation((32, 64, 112, 112), 1, torch.device('cuda'))
    xc = x.cpu()
    # global min / max (classifier-style tags): exact statistics, so every bit must match
    ref = O.gemmlowp_minmax_qdq(xc, 8, tag='activation_classifier')
    y = ops.minmax_qdq_per_tensor(x, 8, avg_over_batch=False).cpu()
    assert torch.equal(y, ref)
    # batch mean of the per-sample extrema (conv activations): fp32 summation-order tier on the two scalars
    ref = O.gemmlowp_minmax_qdq(xc, 8, tag='activation')
    y = ops.minmax_qdq_per_tensor(x, 8, avg_over_batch=True).cpu()
    step = float((xc.max() - xc.min()) / 255)
    assert float((y - ref).abs().max()) <= step * 1.01
    assert float(((y - ref).abs() > 1e-5).float().mean()) < 2e-3
    # ... and bit-exact once the oracle is handed the device's own range / offset
    rows = ops.tensor_row_stats(x, 32)
    mn, mx = rows[0].mean(), rows[1].mean()
    delta, offset = float(mx - mn), float(mn)
    ref2 = O.float2gemmlowp(xc, delta, offset, 8, False, bool((offset + delta) > 0 and offset < 0))
    frac = float((y != ref2).float().mean())
    assert frac < 2e-3, frac          # identical unless the device's mean of 32 extrema rounds differently from torch's


@pytest.mark.parametrize('shape', [(512, 256, 56, 56), (512, 512, 28, 28), (512, 1024, 14, 14), (512, 2048, 7, 7)])
def test_packed_storage_full_size(ops, shape):
    """The bit-allocated packed storage at b512 size: the lean kernels (one channel per wave; 7x7: ragged rows) write the
    bytes of the general kernel, and the round trip returns the fused Q/DQ's floats bit for bit, in both kernel forms."""
    from cnn_quantization_amd import _lib as L
    N, C, H, W = shape
    x = bench.laplace_activation(shape, 5, torch.device('cuda'))
    y, parts = ops.act_qdq_per_channel(x, 4, clip='laplace', bit_alloc=True, want_parts=True)
    qp, bits = parts['qp'], parts['diag'][L.DIAG_BITS].contiguous()
    assert 3.5 <= float(bits.mean()) <= 4.5 and int(bits.max()) > int(bits.min())      # several widths in one tensor
    a, ro = ops.quantize_packed(x, qp, bits, form=1)
    b, ro_b = ops.quantize_packed(x, qp, bits, form=2)
    assert torch.equal(ro, ro_b) and torch.equal(a, b)
    del a
    for form in (1, 2):
        back = ops.dequantize_packed(b, shape, qp, bits, ro, form=form)
        assert torch.equal(back, y), form
        del back
