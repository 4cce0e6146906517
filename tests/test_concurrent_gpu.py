"""The in-launch exchange beside other work on the GPU: a group's workgroups must be resident
together or each member burns a bounded wait and recomputes its extrema - slow but correct.  These tests run the
single-launch kernels while a second stream keeps the CUs busy, assert the bits, and REPORT (never hide) expired
waits through the status word.  Needs an MI355X: `pytest -m gpu`."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from cnn_quantization_amd import ops as _ops
    return _ops


@pytest.mark.parametrize('shape', [(64, 256, 56, 56), (64, 512, 14, 14), (256, 64, 112, 112)])
def test_exchange_beside_a_busy_stream(ops, shape):
    N, C, H, W = shape
    g = torch.Generator(device='cuda').manual_seed(C)
    x = torch.randn(shape, device='cuda', generator=g) * 2
    ref = ops.minmax_qdq_fused(x, N, C, H * W, 4, False, chain=True)
    ops.group_status(x, clear=True)
    # the hog: large GEMMs back to back on a side stream - every CU has matrix work queued for the whole test
    a = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    outs = []
    with torch.cuda.stream(side):
        for _ in range(40):
            c = a @ b
    for _ in range(8):                      # the exchange kernels, dispatched while the GEMMs hold the CUs
        outs.append(ops.act_qdq_per_channel(x, 4))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for y in outs:
        assert torch.equal(y, ref)
    st = ops.group_status(x, clear=True)
    # bit 0 would mean some group was not resident together and paid a 20 ms wait: legal, but it has to be visible
    print('group_status beside a busy stream, shape %s: %d' % (shape, st))
    assert st & ~ops.GROUP_WAIT_EXPIRED == 0
    del c


def test_exchange_on_two_streams_at_once(ops):
    """Two streams, each with its own exchange workspace, running the group kernels concurrently: their groups compete
    for the same CUs."""
    x1 = torch.randn(128, 128, 56, 56, device='cuda')
    x2 = torch.randn(128, 256, 28, 28, device='cuda') * 3
    r1 = ops.minmax_qdq_fused(x1, 128, 128, 3136, 4, False, chain=True)
    r2 = ops.minmax_qdq_fused(x2, 128, 256, 784, 4, True, chain=True)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    s1.wait_stream(torch.cuda.current_stream())
    s2.wait_stream(torch.cuda.current_stream())
    o1, o2 = [], []
    for _ in range(10):
        with torch.cuda.stream(s1):
            o1.append(ops.act_qdq_per_channel(x1, 4))
        with torch.cuda.stream(s2):
            o2.append(ops.act_qdq_per_channel(x2, 4, positive=True))
    torch.cuda.synchronize()
    assert all(torch.equal(y, r1) for y in o1) and all(torch.equal(y, r2) for y in o2)
    for s, x in ((s1, x1), (s2, x2)):
        with torch.cuda.stream(s):
            st = ops.group_status(x, clear=True)
        print('group_status of concurrent stream: %d' % st)
        assert st & ~ops.GROUP_WAIT_EXPIRED == 0
