"""The two halves of config 2 around the cross-rank exchange, each ONE launch (multi-GPU hot path):
cnnq_pc_minmax_local_auto (k_minmax_group: the last workgroup of a channel group to arrive folds the group's pairs) and
cnnq_pc_gathered_qdq (k_qdq<GATH>: every workgroup derives its channels' parameters from the W gathered records) -
bit for bit against the multi-launch forms they replace and against torch, on every tile shape, with NaN, with the
exchange workspace shared with the other group kernels, and replayed.  Needs an MI355X: `pytest -m gpu`."""
import ctypes

import pytest
import torch

from conftest import bits_equal

pytestmark = pytest.mark.gpu

SHAPES = [
    (64, 64, 112, 112),     # mode 1, two-level arrival (26 members)
    (64, 256, 56, 56),      # mode 1, several column slices
    (40, 6, 56, 56),        # few channels, ragged batch split
    (64, 1024, 14, 14),     # mode 2: several whole channels per tile
    (33, 24, 14, 14),       # ragged
    (8, 64, 7, 7),          # float4s straddle channels (A = 4)
    (1, 32, 28, 28),        # one sample
    (3, 16, 5, 9),          # no float4 layout: the two-launch fallback
    (512, 64, 56, 56),      # BASELINE-sized layer (> 384 MB: non-temporal loads)
]


@pytest.fixture(scope='module')
def env():
    from cnn_quantization_amd import _lib as L
    from cnn_quantization_amd import ops
    return L, L.load(), ops


def local_auto(env, x, use_gws=True):
    L, lib, ops = env
    N, C = x.shape[:2]
    HW = x[0, 0].numel()
    G = max(lib.cnnq_pc_groups(N, C, HW, 1), lib.cnnq_pc_groups(N, C, HW, 0))
    pmm = torch.empty((G, 2, C), dtype=torch.float32, device=x.device)
    out = torch.full((2, C), 7.0, dtype=torch.float32, device=x.device)
    gws = ops._group_workspace(x) if use_gws else None
    L.check(lib.cnnq_pc_minmax_local_auto(ops._ptr(x), N, C, HW, ops._ptr(pmm), gws, ops.GROUP_WS_BYTES if gws is not None else 0,
                                          ops._ptr(out), ops._stream(x)), 'cnnq_pc_minmax_local_auto')
    return out


@pytest.mark.parametrize('shape', SHAPES)
def test_local_extrema_one_launch(env, shape):
    L, lib, ops = env
    g = torch.Generator(device='cuda').manual_seed(shape[0] * 7 + shape[1])
    x = torch.randn(shape, device='cuda', generator=g) * 3
    C = shape[1]
    ref = torch.stack([x.transpose(0, 1).reshape(C, -1).min(1)[0], x.transpose(0, 1).reshape(C, -1).max(1)[0]])
    for rep in range(3):                      # the counters re-arm themselves
        out = local_auto(env, x)
        assert bits_equal(out.cpu(), ref.cpu()), (shape, rep)
    assert bits_equal(local_auto(env, x, use_gws=False).cpu(), ref.cpu())      # the two-launch form
    assert ops.group_status(x) & ops.GROUP_WAIT_EXPIRED == 0


def test_local_extrema_nan_and_shared_workspace(env):
    """NaN poisons its channel only; launches of different geometry and of the other group kernels interleave on one
    workspace."""
    L, lib, ops = env
    x = torch.randn(64, 64, 56, 56, device='cuda')
    x[7, 5, 3, 3] = float('nan')
    z = torch.randn(64, 256, 28, 28, device='cuda')
    for _ in range(3):
        out = local_auto(env, x).cpu()
        yq = ops.minmax_qdq_group(z, 64, 256, 784, 4)
        zo = local_auto(env, z)
        assert torch.isnan(out[0, 5]) and torch.isnan(out[1, 5])
        keep = torch.arange(64) != 5
        t = x[:, keep].transpose(0, 1).reshape(63, -1)
        assert bits_equal(out[0, keep], t.min(1)[0].cpu()) and bits_equal(out[1, keep], t.max(1)[0].cpu())
        tz = z.transpose(0, 1).reshape(256, -1)
        assert bits_equal(zo[0].cpu(), tz.min(1)[0].cpu()) and bits_equal(zo[1].cpu(), tz.max(1)[0].cpu())
        assert yq is not None


@pytest.mark.parametrize('shape', SHAPES[:8])
@pytest.mark.parametrize('W', [1, 3])
@pytest.mark.parametrize('half', [False, True])
def test_gathered_qdq_one_launch(env, shape, W, half):
    """Parameters derived inside the Q/DQ launch == k_minmax_params + k_qdq, and the table it leaves behind."""
    L, lib, ops = env
    g = torch.Generator(device='cuda').manual_seed(shape[1] + W)
    x = torch.randn(shape, device='cuda', generator=g) * 2
    if half:
        x = x.abs()
    N, C = shape[:2]
    HW = x[0, 0].numel()
    t = x.transpose(0, 1).reshape(C, -1)
    # W "ranks": this shard's record plus records that widen some channels' ranges
    rec = torch.stack([t.min(1)[0], t.max(1)[0]]).unsqueeze(0).repeat(W, 1, 1).contiguous()
    for r in range(1, W):
        rec[r, 0] -= torch.rand(C, device='cuda') * (r % 2)
        rec[r, 1] += torch.rand(C, device='cuda')
    st = ops._stream(x)
    qp_ref = torch.empty((L.NQP, C), dtype=torch.float32, device='cuda')
    L.check(lib.cnnq_pc_minmax_params(ops._ptr(rec), W, C, 4, int(half), ops._ptr(qp_ref), st), 'params')
    y_ref = torch.empty_like(x)
    L.check(lib.cnnq_pc_qdq(ops._ptr(x), ops._ptr(y_ref), N, C, HW, ops._ptr(qp_ref), None, None, 1, st), 'qdq')
    y = torch.empty_like(x)
    qp = torch.full((L.NQP, C), -1.0, dtype=torch.float32, device='cuda')
    L.check(lib.cnnq_pc_gathered_qdq(ops._ptr(x), ops._ptr(y), N, C, HW, ops._ptr(rec), W, 4, int(half), ops._ptr(qp), st),
            'gathered_qdq')
    assert bits_equal(y.cpu(), y_ref.cpu()), (shape, W, half)
    assert bits_equal(qp.cpu(), qp_ref.cpu()), (shape, W, half)
    y2 = torch.empty_like(x)                                   # the table is optional
    L.check(lib.cnnq_pc_gathered_qdq(ops._ptr(x), ops._ptr(y2), N, C, HW, ops._ptr(rec), W, 4, int(half), None, st), 'gathered_qdq')
    assert bits_equal(y2.cpu(), y_ref.cpu())


# ---- random geometries (hypothesis, derandomised): odd H*W, single samples, unaligned base pointers, W ranks
from hypothesis import HealthCheck, given, settings  # noqa: E402
from hypothesis import strategies as hst  # noqa: E402

_CFG = dict(max_examples=60, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
_shapes = hst.tuples(hst.integers(1, 70), hst.integers(1, 40), hst.integers(1, 19), hst.integers(1, 19)).filter(
    lambda s: s[2] * s[3] > 1)


@settings(**_CFG)
@given(shape=_shapes, seed=hst.integers(0, 2 ** 16), offset=hst.integers(0, 3), W=hst.integers(1, 4), half=hst.booleans(),
       bits=hst.sampled_from([2, 4, 8]))
def test_exchange_halves_random_geometry(shape, seed, offset, W, half, bits):
    """local extrema (one launch where the geometry has a group plan) == torch; the shard quantized with W gathered
    records == the oracle on the concatenation of W such shards (the other ranks' data only enters through their
    records: here scaled copies of this shard)."""
    from cnn_quantization_amd import _lib as L
    from cnn_quantization_amd import ops
    from oracle import quant_oracle as O
    lib = L.load()
    g = torch.Generator().manual_seed(seed)
    n = shape[0] * shape[1] * shape[2] * shape[3]
    x = torch.randn(shape, generator=g) * (0.2 + 3 * torch.rand(1, shape[1], 1, 1, generator=g))
    if half:
        x = x.abs()
    base = torch.empty(n + 3, device='cuda')
    xd = base[offset:offset + n].view(shape)                 # offset != 0: base pointer not 16-byte aligned
    xd.copy_(x)
    N, C, HW = shape[0], shape[1], shape[2] * shape[3]
    out = local_auto((L, lib, ops), xd)
    t = x.transpose(0, 1).reshape(C, -1)
    assert bits_equal(out.cpu(), torch.stack([t.min(1)[0], t.max(1)[0]]))
    scales = [1.0, 1.5, 0.25, 2.0][:W]
    shards = [x * s_ for s_ in scales]                       # rank r holds shard r; this process is rank 0
    rec = torch.stack([torch.stack([(sh.transpose(0, 1).reshape(C, -1)).min(1)[0], (sh.transpose(0, 1).reshape(C, -1)).max(1)[0]])
                       for sh in shards]).cuda().contiguous()
    y = torch.empty_like(xd)
    L.check(lib.cnnq_pc_gathered_qdq(ops._ptr(xd), ops._ptr(y), N, C, HW, ops._ptr(rec), W, bits, int(half), None,
                                     ops._stream(xd)), 'gathered_qdq')
    ref = O.act_per_channel_qdq(torch.cat(shards), bits, half_range=half)[:N]
    assert bits_equal(y.cpu().numpy(), ref.numpy())
