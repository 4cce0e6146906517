"""HIP-graph safety: the library only enqueues kernels on the caller's stream (no allocation, no
synchronisation, no host reads), so a whole per-channel Q/DQ sequence can be captured once and
replayed - the way a serving stack would remove launch overhead for small layers."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_capture_and_replay_matches_eager():
    from cnn_quantization_amd import ops
    torch.manual_seed(5)
    static_x = torch.randn(16, 32, 14, 14, device='cuda')
    eager = {}
    for name, kw in (('cfg2', dict()), ('cfg3', dict(clip='laplace', bit_alloc=True))):
        eager[name] = ops.act_qdq_per_channel(static_x, 4, **kw).clone()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):                                     # warm-up on the side stream
            ops.act_qdq_per_channel(static_x, 4)
            ops.act_qdq_per_channel(static_x, 4, clip='laplace', bit_alloc=True)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out2 = ops.act_qdq_per_channel(static_x, 4)
        out3 = ops.act_qdq_per_channel(static_x, 4, clip='laplace', bit_alloc=True)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out2, eager['cfg2']) and torch.equal(out3, eager['cfg3'])
    # new data in the static input, replay only
    new = torch.randn_like(static_x) * 3 + 1
    static_x.copy_(new)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out2, ops.act_qdq_per_channel(new, 4))
    assert torch.equal(out3, ops.act_qdq_per_channel(new, 4, clip='laplace', bit_alloc=True))
