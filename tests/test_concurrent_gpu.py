"""The in-launch exchange beside other work on the GPU: a group's workgroups must be resident
together or each member burns a bounded wait and recomputes its extrema - slow but correct.  These tests run the
single-launch kernels while a second stream keeps the CUs busy, assert the bits, and REPORT (never hide) expired
waits through the status word.  Needs an MI355X: `pytest -m gpu`."""
import json
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import quant_oracle as O  # noqa: E402  (the checker: tests may use it)

pytestmark = pytest.mark.gpu
ARTIFACT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'concurrent_status.json')


def record(key, status):
    """`pytest -q` drops what a passing test prints: the status words go to gpurun_out/concurrent_status.json instead
    (0 = no bounded wait expired; bit 0 = some group was not resident together and recomputed its extrema)."""
    try:
        os.makedirs(os.path.dirname(ARTIFACT), exist_ok=True)
        d = {}
        if os.path.exists(ARTIFACT):
            with open(ARTIFACT) as f:
                d = json.load(f)
        d[key] = int(status)
        with open(ARTIFACT, 'w') as f:
            json.dump(d, f, indent=1, sort_keys=True)
    except OSError:
        pass


@pytest.fixture(scope='module')
def ops():
    from cnn_quantization_amd import ops as _ops
    return _ops


@pytest.mark.parametrize('shape', [(64, 256, 56, 56), (64, 512, 14, 14), (256, 64, 112, 112)])
def test_exchange_beside_a_busy_stream(ops, shape):
    N, C, H, W = shape
    g = torch.Generator(device='cuda').manual_seed(C)
    x = torch.randn(shape, device='cuda', generator=g) * 2
    ref = O.act_per_channel_qdq(x.cpu(), 4).cuda()             # the oracle on the whole tensor, not the HIP chain
    ops.group_status(x, clear=True)
    # without the hog nothing may expire: the status word stays 0
    for _ in range(4):
        assert torch.equal(ops.act_qdq_per_channel(x, 4), ref)
    st0 = ops.group_status(x, clear=True)
    record('alone %s' % (shape,), st0)
    assert st0 == 0
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
    record('beside 40 GEMMs of 8192^3 %s' % (shape,), st)
    assert st & ~ops.GROUP_WAIT_EXPIRED == 0
    del c


def test_exchange_on_two_streams_at_once(ops):
    """Two streams, each with its own exchange workspace, running the group kernels concurrently: their groups compete
    for the same CUs."""
    x1 = torch.randn(128, 128, 56, 56, device='cuda')
    x2 = torch.randn(128, 256, 28, 28, device='cuda') * 3
    r1 = O.act_per_channel_qdq(x1.cpu(), 4).cuda()
    r2 = O.act_per_channel_qdq(x2.cpu(), 4, half_range=True).cuda()
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
        record('two streams at once, stream of %s' % (tuple(x.shape),), st)
        assert st & ~ops.GROUP_WAIT_EXPIRED == 0


@pytest.mark.parametrize('batch', [64, 512])
def test_two_processes_share_the_gpu(batch):
    """Beside another PROCESS at bench size: two bench.py jobs (batch 64, and the full batch 512: 45 GB
    each) on the one GPU at the same time.  Kernels of two processes need not run together, so a group's members may be kept off the CUs by the other process's
    workgroups: waits may expire (bit 0 of the status word, reported in the line and recorded), the bits may not change."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--batch', str(batch), '--steps', '6', '--warmup', '2', '--no-cpu-baseline',
           '--no-other-configs']
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=root, env=env) for _ in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    for i, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, se[-2000:]
        d = json.loads([l for l in so.splitlines() if l.strip()][-1])
        record('two processes at batch %d, process %d' % (batch, i), d['group_status'])
        assert d['verified'] is True
        assert d['group_status'] & ~1 == 0          # nothing but "a wait expired" may ever be raised here
