"""Config 1 in one launch (cnnq_pt_minmax_qdq_fused, k_pt_fused: two sweeps with a meeting of the workgroups in between;
opt-in - it measured slower than the chain) against the four-launch chain - bit for bit on ragged rows, one row, thousands of rows, both statistics
modes, half range, power-of-two scales, NaN / inf - the workspace re-arming itself across launches and geometries,
interleaved with the group kernels on the same workspace, and against the oracle (iq.py:361-379 +
kernels/gemmlowp.cu:8-45).  Needs an MI355X: `pytest -m gpu`."""
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


SHAPES = [(32, 64, 28, 28), (6, 4, 9, 12), (1, 3, 64, 64), (5, 1000), (700, 4, 2, 2), (3, 4100), (4096, 8), (16, 16, 56, 56),
          (2, 4, 130, 130)]


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('avg,half,int_exp', [(True, False, False), (False, False, False), (True, True, False),
                                              (False, False, True)])
def test_fused_equals_chain(ops, shape, avg, half, int_exp):
    gen = torch.Generator().manual_seed(len(shape) * 100 + shape[0])
    ops.group_status(torch.empty(1, device='cuda'), clear=True)
    for rnd in range(3):                      # the region re-arms itself
        x = (torch.randn(shape, generator=gen) * (1 + rnd) + 0.3 * rnd).cuda()
        a = ops.minmax_qdq_per_tensor(x, 8, avg_over_batch=avg, zero_min=half, int_exp=int_exp, fused=True)
        b = ops.minmax_qdq_per_tensor(x, 8, avg_over_batch=avg, zero_min=half, int_exp=int_exp, fused=False)
        assert torch.equal(a, b), (shape, rnd)
        assert ops.act_qdq_per_channel(torch.randn(40, 6, 56, 56, device='cuda'), 4) is not None   # shares the workspace
    assert ops.group_status(x) == 0


def test_fused_vs_oracle_and_nan(ops):
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(5, 1000, generator=gen) * 1.5 + 0.2
    assert bits_equal(ops.minmax_qdq_per_tensor(x.cuda(), 8, avg_over_batch=False, fused=True).cpu(),
                      O.gemmlowp_minmax_qdq(x, 8, tag='activation_classifier'))
    # NaN: torch.min / torch.max propagate it (iq.py:515-528) - the scale is NaN and so is everything; same as the chain
    for avg in (True, False):
        xn = torch.randn(6, 4, 8, 8, generator=gen)
        xn[2, 1, 3, 3] = float('nan')
        xn[0, 0, 0, 0] = float('inf')
        a = ops.minmax_qdq_per_tensor(xn.cuda(), 8, avg_over_batch=avg, fused=True).cpu().numpy()
        b = ops.minmax_qdq_per_tensor(xn.cuda(), 8, avg_over_batch=avg, fused=False).cpu().numpy()
        assert np.array_equal(np.isnan(a), np.isnan(b)) and bool(np.isnan(a).all())
    # a constant tensor: range 0 -> the input comes back (kernels/int_quantization.cpp: range <= 0)
    c = torch.full((4, 3, 8, 8), 1.25, device='cuda')
    assert torch.equal(ops.minmax_qdq_per_tensor(c, 8, avg_over_batch=True, fused=True), c)


def test_fused_full_size_equals_chain(ops):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    for shape, avg in (((32, 64, 112, 112), True), ((512, 64, 56, 56), True), ((512, 1000), False)):
        x = bench.laplace_activation(shape, 5, torch.device('cuda')) if len(shape) == 4 else torch.randn(shape, device='cuda')
        a = ops.minmax_qdq_per_tensor(x, 8, avg_over_batch=avg, fused=True)
        b = ops.minmax_qdq_per_tensor(x, 8, avg_over_batch=avg, fused=False)
        assert torch.equal(a, b), shape
    assert ops.group_status(x) == 0
