"""The code histogram of the mid-tread path (config 5, -me: utils/entropy.py:8-15 over the integer codes): the entropy
the device computes from its windowed LDS / replica histogram must equal -sum p log2 p over torch.unique of the very
codes the kernel stored - inside the window (the normal case), with codes beyond the window (many bins per channel:
the global bins and the flag word), negative clamp bounds (symmetric ranges), non-integer clamp values, the unclipped
weight form, and ragged shapes.  Needs an MI355X: `pytest -m gpu`."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from cnn_quantization_amd import ops as _ops
    return _ops


def entropy_of(codes):
    """utils/entropy.py:8-15 on the device codes (fp64)."""
    _, counts = torch.unique(codes.flatten(), return_counts=True)
    p = counts.double() / codes.numel()
    return float(-(p * torch.log2(p)).sum())


CASES = [
    # shape, target bits, sym, scale spread
    ((64, 64, 56, 56), 4.0, False, 1.0),      # the benchmark's regime: every code inside the window
    ((16, 32, 28, 28), 4.0, True, 1.0),       # symmetric: negative clamp bounds, window starts below zero
    ((8, 16, 14, 14), 8.0, False, 1.0),       # ~256 bins per channel: codes beyond the 128-code window
    ((8, 16, 14, 14), 9.0, True, 4.0),        # ~512 bins, wide spread of channel scales
    ((3, 5, 7, 9), 3.0, False, 2.0),          # rows that are not whole float4s
    ((1, 2, 4, 4), 2.0, True, 1.0),           # tiny
]


@pytest.mark.parametrize('shape,target,sym,spread', CASES)
def test_entropy_matches_unique_of_device_codes(ops, shape, target, sym, spread):
    from cnn_quantization_amd import _lib as L
    g = torch.Generator(device='cuda').manual_seed(int(target * 100) + shape[1])
    C = shape[1]
    scale = (1 + spread * torch.arange(C, device='cuda').view(1, C, 1, 1) / max(C - 1, 1))
    x = torch.empty(shape, device='cuda').exponential_(generator=g) * scale
    x = x * torch.where(torch.rand(shape, device='cuda', generator=g) < 0.5, -1.0, 1.0)
    if not sym:
        x = x.clamp(min=0)
    y, ent, codes, parts = ops.mid_tread_qdq(x, target, clip=True, sym=sym, want_entropy=True, want_codes=True,
                                             want_parts=True)
    ref = entropy_of(codes)
    assert abs(float(ent) - ref) < 2e-4 * max(1.0, ref), (float(ent), ref)
    hist = parts['hist']
    # every element was counted exactly once, wherever its count went
    w0, nb = int(parts['mt'][L.MT_WSTART][0]), L.MT_HIST_BINS
    assert int(hist[:-1].sum()) == x.numel()
    window_codes = (codes >= w0) & (codes < w0 + L.MT_HIST_WINDOW) & (codes == codes.round())
    reps = hist[nb + 2 + 2 * C:-1].view(L.MT_HIST_REPLICAS, L.MT_HIST_WINDOW).sum(0)
    assert int(reps.sum()) == int(window_codes.sum())
    assert (int(hist[-1]) != 0) == bool(int(hist[:nb + 2].sum()) != 0)      # the flag word tells the truth
    # spot check: the count of the most frequent integer code
    vals, counts = torch.unique(codes[window_codes], return_counts=True)
    top = int(counts.argmax())
    assert int(reps[int(vals[top]) - w0]) == int(counts[top])


def test_entropy_unclipped_weights(ops):
    """clip=False (weights, per output channel): the window is centred on zero."""
    w = torch.randn(96, 32, 3, 3, device='cuda') * 0.05
    y, ent, codes = ops.mid_tread_qdq(w, 4, clip=False, sym=True, per_channel_dim=0, group=False, want_entropy=True,
                                      want_codes=True)
    ref = entropy_of(codes)
    assert abs(float(ent) - ref) < 2e-4 * max(1.0, ref)
    assert math.isfinite(float(ent))


def test_entropy_zero_code_outside_the_window(ops):
    """A symmetric clip with ~512 bins per channel puts code 0 beyond the 128-code window (it starts at the smallest
    clamp bound, about -256): the register-counted zeros go to the global bins and must raise the flag word, or the
    mode of the distribution is missing from the entropy.  Mostly-zero tensor: nothing else raises it."""
    g = torch.Generator(device='cuda').manual_seed(11)
    x = torch.zeros(8, 16, 14, 14, device='cuda')
    mask = torch.rand(x.shape, device='cuda', generator=g) < 0.02
    x[mask] = torch.randn(int(mask.sum()), device='cuda', generator=g) * 0.01
    y, ent, codes, parts = ops.mid_tread_qdq(x, 9.0, clip=True, sym=True, want_entropy=True, want_codes=True, want_parts=True)
    from cnn_quantization_amd import _lib as L
    assert int(parts['mt'][L.MT_WSTART][0]) + L.MT_HIST_WINDOW <= 0          # code 0 really is outside the window
    ref = entropy_of(codes)
    assert ref > 0.05
    assert abs(float(ent) - ref) < 2e-4 * max(1.0, ref), (float(ent), ref)
    assert int(parts['hist'][:-1].sum()) == x.numel()
