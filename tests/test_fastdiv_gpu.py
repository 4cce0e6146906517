"""The divide-free exact quotient inside the single-launch kernels (qdq1_fast, csrc/cnnq_qdq.hip.h) against the hardware
divide, on tensors built to sit on its edges:
  * channels inside the domain whose values lie a few ulps around the rounding ties of the quotient and of the code,
    channels of zeros / denormals / tiny numbers, a constant channel (the 1e-8 scale floor), a channel far from zero
    (zero point ~ -15000), a channel of signed zeros;
  * channels OUTSIDE the domain (|x| > 2^70, inf, NaN) that must route their workgroups to the hardware divide.
y, codes and parameters must equal, bit for bit, (a) the three-launch chain, whose Q/DQ pass divides, (b) the same
single launch with the divide forced (flags bit 1), (c) the CPU oracle.  Needs an MI355X: `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from conftest import bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from cnn_quantization_amd import ops as _ops
    return _ops


def edge_tensor(shape, seed, out_of_domain):
    N, C, H, W = shape
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g) * (torch.rand(1, C, 1, 1, generator=g) * 3 + 0.1)
    kinds = ['ties', 'tiny', 'const', 'far', 'zeros', 'denorm', 'plain', 'halfties']
    if out_of_domain:
        kinds += ['huge', 'inf', 'nan', 'hugescale']
    for c in range(C):
        kind = kinds[c % len(kinds)]
        v = x[:, c]
        if kind == 'ties':            # values a few ulps around scale * (k + 1/2), scale = (max - min) / 15 with min = 0, max = 15 s
            s = float(torch.rand(1, generator=g)) * 2 + 0.01
            k = torch.randint(0, 15, v.shape, generator=g).float() + 0.5
            t = (k.double() * s).float()
            t = (t.view(torch.int32) + torch.randint(-3, 4, v.shape, generator=g, dtype=torch.int32)).view(torch.float32)
            v.copy_(t)
            v[0, 0, 0] = 0.
            v[0, 0, 1] = 15 * s
        elif kind == 'halfties':      # the same around integers
            s = float(torch.rand(1, generator=g)) * 0.5 + 0.001
            t = (torch.randint(1, 16, v.shape, generator=g).double() * s).float()
            t = (t.view(torch.int32) + torch.randint(-2, 3, v.shape, generator=g, dtype=torch.int32)).view(torch.float32)
            v.copy_(t)
            v[0, 0, 0] = 0.
            v[0, 0, 1] = 15 * s
        elif kind == 'tiny':
            e = torch.randint(-140, -60, v.shape, generator=g).float()
            v.copy_(torch.rand(v.shape, generator=g) * e.exp2() * (torch.randint(0, 2, v.shape, generator=g).float() * 2 - 1))
            v[0, 0, 0] = 1.0
        elif kind == 'const':
            v.fill_(1000.0)
        elif kind == 'far':
            v.copy_(1000.0 + torch.rand(v.shape, generator=g))
        elif kind == 'zeros':
            v.copy_(torch.where(torch.rand(v.shape, generator=g) < 0.5, torch.tensor(0.0), torch.tensor(-0.0)))
            v[0, 0, 0] = 0.75
        elif kind == 'denorm':
            v.copy_(torch.randint(0, 1 << 23, v.shape, generator=g, dtype=torch.int32).view(torch.float32))
            v[0, 0, 0] = -1e-38
        elif kind == 'huge':
            v.mul_(1e30)
        elif kind == 'inf':
            v[N // 2, H // 2, 0] = float('inf')
        elif kind == 'nan':
            v[N // 3, H // 3, 1] = float('nan')
        elif kind == 'hugescale':
            v.mul_(1e12)                      # |x| inside 2^70 but the scale above 2^30
    return x


SHAPES = [
    (40, 8, 56, 56), (130, 12, 28, 28), (20, 8, 112, 112),        # k_mmq_flat
    (70, 40, 7, 7), (37, 24, 14, 14), (64, 64, 14, 14),           # k_mmq_group A = 4 / 1
    (8, 32, 14, 14), (8, 64, 7, 7), (4, 16, 28, 28),              # k_mmq_whole
]


@pytest.mark.parametrize('out_of_domain', [False, True])
@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('bits,half', [(4, False), (4, True), (8, False)])
def test_edges_equal_the_divide(ops, shape, bits, half, out_of_domain):
    from oracle import quant_oracle as O
    N, C, H, W = shape
    x = edge_tensor(shape, sum(shape) + bits, out_of_domain)
    xd = x.cuda()
    yc, cc, pc = ops.minmax_qdq_fused(xd, N, C, H * W, bits, half, want_codes=True, want_parts=True, chain=True)
    res = ops.minmax_qdq_single(xd, N, C, H * W, bits, half, want_codes=True, want_parts=True)
    assert res is not None
    y, codes, parts = res
    assert bits_equal(y.cpu(), yc.cpu().numpy()) and torch.equal(codes, cc)
    qa, qb = parts['qp'].cpu().numpy().view(np.uint32), pc['qp'].cpu().numpy().view(np.uint32)
    assert np.array_equal(qa, qb), [(i, hex(qa[i]), hex(qb[i])) for i in zip(*np.nonzero(qa != qb))]
    ref, rp = O.act_per_channel_qdq(x, bits, half_range=half, return_parts=True)
    ref = np.asarray(ref, dtype=np.float32)
    # the parameter table against the oracle's: scale and zero point bit for bit (the zero point of a channel whose
    # minimum is +-0 is +0, iq.py:570-572 - the compiler once turned it into -0), NaN where the oracle has NaN
    from cnn_quantization_amd import _lib as L
    for row, key in ((L.QP_SCALE, 'scale'), (L.QP_ZP, 'zero_point')):
        want = np.asarray(rp[key], dtype=np.float32).reshape(-1)
        got = parts['qp'][row].cpu().numpy()
        nn = np.isnan(want)
        assert np.array_equal(np.isnan(got), nn) and bits_equal(np.where(nn, 0, got), np.where(nn, 0, want)), key
    yh = y.cpu().numpy()
    na = np.isnan(ref)                      # NaN where the oracle has NaN (the payloads of x86 and gfx950 differ), bits elsewhere
    assert np.array_equal(np.isnan(yh), na) and bits_equal(np.where(na, 0, yh), np.where(na, 0, ref))
    r = ops.minmax_qdq_group(xd, N, C, H * W, bits, half, flags=2)
    if r is not None:                       # the same kernel, every channel through the hardware divide
        assert bits_equal(r.cpu(), yc.cpu().numpy())
    assert ops.group_status(xd) == 0
