#!/usr/bin/env python3
"""bench.py - throughput of the per-channel int4 quantize/dequantize hot path on MI355X.

Workload (BASELINE.json configs[1]): every conv activation of one ResNet-50 forward at batch 512
(53 tensors, 5.69 G fp32 elements), quantized per channel to int4 with
dynamic min/max statistics (`-pcq_a --qtype int4`; half-range layers clamp the minimum to 0).
One "step" = the whole set once.  Inputs are synthetic per-channel Laplace activations generated
on the device (seed 12345), resident in HBM before the timed region; all inputs and outputs are
distinct buffers, so nothing is re-read from a cache across layers.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU (inference_sim.py:196-200): ONE batch of 512 is sharded over the N ranks,
512/N samples per GPU (`"scaling": "strong"`; `--scaling weak` gives every rank its own 512);
per-channel statistics are made global with one small all_gather per tensor (RCCL over xGMI),
the Q/DQ itself needs no communication.

Per tensor the product entry point (ops.act_qdq_per_channel) picks a SINGLE launch that reads x once
(4 B read + 4 B write per element, the tile stays in registers between statistics and Q/DQ):
  * k_mmq_whole when a channel's batch population fits one workgroup (small H*W at batch <= 64),
  * k_mmq_group otherwise: the workgroups that share a channel exchange {min, max} inside the launch;
and only for shapes neither supports (unaligned, odd H*W) the three-launch chain
k_minmax (4 B read) -> k_minmax_params -> k_qdq (4 B read + 4 B write).

Prints ONE JSON line (rank 0): `value` = elements/s of the whole job; `roofline` = the dominant
kernel's algorithmic bytes / its measured launch time (HIP events on the launch stream) against
the 8 TB/s HBM3E peak; `cpu_baseline` = the oracle (CPU restatement of the reference's op chain)
timed on this host; `other_configs` = BASELINE configs 1, 3, 4, 5 timed the same way; `verified`
= the outputs of the timed steps were checked after the timed region.
"""
import argparse
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# (C, H=W, half_range, layers) - ResNet-50 v1.5 conv outputs, SURVEY.md Appendix B
RESNET50_CONV_OUTPUTS = [
    (64, 112, True, 1), (256, 56, False, 4), (128, 56, True, 1), (512, 28, False, 5), (64, 56, True, 6),
    (256, 28, True, 1), (1024, 14, False, 7), (128, 28, True, 7), (512, 14, True, 1), (2048, 7, False, 4),
    (256, 14, True, 11), (512, 7, True, 5),
]
VGG16_CONV_OUTPUTS = [(64, 224, 2), (128, 112, 2), (256, 56, 3), (512, 28, 3), (512, 14, 3)]   # (C, H=W, layers)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_QDQ = 8                  # fused Q/DQ pass, and the resident single launch: 4 B read + 4 B write per element
BYTES_STATS = 4                # statistics pass: one read
BYTES_PATH = 12                # dynamic min/max + Q/DQ as SURVEY.md 8(d3) accounts it


def laplace_activation(shape, gen_seed, device):
    """Per-channel Laplace(mu_c, b_c), mu_c ~ N(0, 0.5), b_c ~ LogUniform(0.05, 2)."""
    g = torch.Generator(device=device).manual_seed(gen_seed)
    N, C, H, W = shape
    mu = torch.randn(C, generator=g, device=device) * 0.5
    b = torch.exp(torch.empty(C, device=device).uniform_(math.log(0.05), math.log(2.0), generator=g))
    x = torch.empty(shape, device=device, dtype=torch.float32)
    chunk = max(1, (1 << 26) // (C * H * W))       # bound the temporaries
    for n0 in range(0, N, chunk):
        # |u| < 0.5 strictly: rand() == 0 would give log1p(-1) = -inf (about 6 samples per 100 M)
        u = (torch.rand((min(chunk, N - n0), C, H, W), generator=g, device=device) - 0.5) * (1 - 2 ** -20)
        x[n0:n0 + chunk] = mu.view(1, C, 1, 1) - b.view(1, C, 1, 1) * torch.sign(u) * torch.log1p(-2 * u.abs())
    return x


def build_workload(batch, device, seed=12345):
    layers = []
    for (C, hw, half, count) in RESNET50_CONV_OUTPUTS:
        for _ in range(count):
            x = laplace_activation((batch, C, hw, hw), seed, device)
            layers.append(dict(x=x, y=torch.empty_like(x), half=half, N=batch, C=C, HW=hw * hw))
            seed += 1
    return layers


def run_step(ops, layers, group):
    """One pass of the hot path over the 53 activations (the product entry point the quantizer
    uses: IntQuantizer.gemmlowpQuantizeActivationPerChannel -> ops.act_qdq_per_channel)."""
    for L in layers:
        ops.act_qdq_per_channel(L['x'], 4, positive=L['half'], group=group, out=L['y'])


from bench_roofline import roofline_objects  # noqa: E402  (the `roofline` keys: live HIP-event timing per kernel class)


def verify_outputs(ops, layers, group=None, world=1):
    """After the timed region: the buffers the timed steps wrote are checked on the largest tensor and on the
    largest resident-kernel tensor - extrema against torch's own reductions (with several ranks: all-reduced over
    the ranks, i.e. the extrema of the GLOBAL batch, through typed MIN / MAX all-reduces that the product path
    never uses), codes within [0, 15], y equal to (code - zp) * scale bit for bit, |x - y| <= scale / 2 inside
    the range, and equal to a second run that also returns its parameters.  Every rank must call it."""
    from cnn_quantization_amd import _lib as Lb
    import ctypes
    lib = Lb.load()
    d = (ctypes.c_int32 * 8)()
    big = max(layers, key=lambda L: L['x'].numel())
    res = [L for L in layers if lib.cnnq_pc_resident_describe(L['N'], L['C'], L['HW'], d) == 0]
    picks = [big] + ([max(res, key=lambda L: L['x'].numel())] if res else [])
    ok = True
    for L in picks:
        x, y, C = L['x'], L['y'], L['C']
        y2, parts = ops.act_qdq_per_channel(x, 4, positive=L['half'], want_parts=True, group=group)
        qp, st = parts['qp'], parts['stats']
        sc, zp = qp[0].view(1, C, 1, 1), qp[1].view(1, C, 1, 1)
        ok = ok and bool(torch.equal(y2, y))
        mx, mn = x.amax(dim=(0, 2, 3)), x.amin(dim=(0, 2, 3))
        if world > 1:
            on_dev = dist.get_backend(group) == 'nccl'
            mxr, mnr = (mx, mn) if on_dev else (mx.cpu(), mn.cpu())
            dist.all_reduce(mxr, op=dist.ReduceOp.MAX, group=group)
            dist.all_reduce(mnr, op=dist.ReduceOp.MIN, group=group)
            mx, mn = mxr.to(x.device), mnr.to(x.device)
        ok = ok and bool(torch.equal(st[1], mx)) and bool(torch.equal(st[0], mn))
        del y2
        codes = torch.round(y / sc + zp)
        ok = ok and float(codes.min()) >= 0 and float(codes.max()) <= 15
        ok = ok and bool(torch.equal((codes - zp) * sc, y))
        del codes
        err = (x - y).abs_().sub_(0.5001 * sc)
        if L['half']:
            err = err.masked_fill_(x < 0, 0.)          # below the half range everything clamps to 0
        ok = ok and float(err.max()) <= 0.
        del err
    return bool(ok)


def timed_best(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def other_configs(ops, device, batch):
    """BASELINE.json configs 1, 3, 4, 5 and config 2's optional outputs, timed and verified: bench_other.py."""
    import bench_other
    return bench_other.other_configs(ops, device, batch)


def box_id(index=0):
    """The GPU's unique id (what `rocm-smi --showuniqueid` prints; torch exposes it as the ASCII of the device uuid): the
    boxes of the pool differ by several percent on this workload, so every bench line and every profiles/ file names
    the box it comes from."""
    try:
        u = str(torch.cuda.get_device_properties(index).uuid).replace('-', '')
        return 'gpu-' + bytes.fromhex(u).decode('ascii')
    except Exception:                                        # noqa: BLE001 - an identifier, never a reason to fail
        return 'unknown'


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(batch_sample=8, reps=5):
    """The oracle (op-for-op CPU restatement of the reference's chain, iq.py:409-451 / 557-603) on the host of
    this box: the same per-channel Laplace inputs as the GPU leg at a small batch, thread counts 1 / 8 / 32 / 64
    (capped at the host's CPUs; torch's intra-op pool with all 256 hardware threads of the GPU box runs this chain of
    small ops 400x SLOWER than one thread - measured 1.3 M elements/s - so "all" is not in the sweep), median of
    `reps` passes each after a warm-up pass; plus BASELINE config 1 on exactly [32,64,112,112]."""
    from oracle import quant_oracle as O
    ncpu = os.cpu_count() or 1
    xs, seed, elems = [], 12345, 0
    for (C, hw, half, count) in RESNET50_CONV_OUTPUTS:
        x = laplace_activation((batch_sample, C, hw, hw), seed, torch.device('cpu'))
        xs.append((x, half, count))
        elems += x.numel() * count
        seed += count
    x1 = laplace_activation((32, 64, 112, 112), 1, torch.device('cpu'))

    def cfg2():
        for x, half, count in xs:
            for _ in range(count):
                O.act_per_channel_qdq(x, 4, half_range=half)

    def cfg1():
        O.gemmlowp_minmax_qdq(x1, 8, tag='activation')

    def median_time(fn):
        t0 = time.perf_counter()
        fn()
        if time.perf_counter() - t0 > 4.:          # a pathological thread count: one more pass is enough to say so
            t0 = time.perf_counter()
            fn()
            return time.perf_counter() - t0
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    saved = torch.get_num_threads()
    sweep, sweep1 = {}, {}
    for th in sorted({1, min(8, ncpu), min(32, ncpu), min(64, ncpu)}):
        torch.set_num_threads(th)
        sweep[th] = elems / median_time(cfg2)
        sweep1[th] = x1.numel() / median_time(cfg1)
    torch.set_num_threads(saved)
    best = max(sweep, key=sweep.get)
    return dict(value=sweep[best], unit='elements/s', cores=best, kind='port', cpu=cpu_model(), host_cpus=ncpu,
                one_thread=sweep[1], by_threads={str(k): v for k, v in sweep.items()},
                config1={'value': max(sweep1.values()), 'cores': max(sweep1, key=sweep1.get), 'one_thread': sweep1[1],
                         'by_threads': {str(k): v for k, v in sweep1.items()},
                         'sample': 'oracle.gemmlowp_minmax_qdq (per-tensor int8) on one [32,64,112,112] tensor'},
                sample='oracle.act_per_channel_qdq over the 53 ResNet-50 conv outputs at batch %d (%.1f M elements per '
                       'pass), the GPU leg\'s per-channel Laplace inputs, median of %d passes per thread count after '
                       'a warm-up pass; value = the best thread count' % (batch_sample, elems / 1e6, reps))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=512, help='GLOBAL batch of the forward (BASELINE config: 512)')
    ap.add_argument('--scaling', choices=('strong', 'weak'), default='strong',
                    help='strong: the one batch is sharded, batch/N per GPU (SURVEY 8 e1); weak: every GPU its own batch')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true')
    ap.add_argument('--graph', action='store_true',
                    help='replay the step from a HIP graph (also the sharded step, its exchange included: every rank captures '
                         'and replays the same launches)')
    ap.add_argument('--sustained-secs', type=float, default=2.0,
                    help='after the timed K steps: keep stepping for about this long and report that rate too (0: skip)')
    ap.add_argument('--force-exchange', action='store_true',
                    help='single GPU: run the multi-GPU launch sequence with a 1-rank RCCL group (a self all_gather per '
                         'tensor) - measures what the collective costs on this box')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` started as a plain process: start the N ranks ourselves (one per GPU, the
        # launcher the contract names); rank 0 of the children prints the one JSON line on the inherited stdout
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    # stdout carries exactly ONE line, the JSON: libraries that print to fd 1 (RCCL's version banner at communicator
    # creation, gloo's connection notes) are sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # one rank per GPU; CNNQ_BENCH_BACKEND=gloo (test rigs with fewer GPUs than ranks) lets several
    # ranks share a device so the multi-rank code path can be exercised on a 1-GPU box
    backend = os.environ.get('CNNQ_BENCH_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    group = None
    if world > 1 or args.force_exchange:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29577')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.force_exchange:
        os.environ['CNNQ_FORCE_EXCHANGE'] = '1'
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

    from cnn_quantization_amd import ops, _lib
    from cnn_quantization_amd import distributed as D
    _lib.load()                                         # fail loudly if the HIP library is missing
    # The library's default for a sharded run is 'auto' since round 6 (with one GPU per rank the in-launch exchange, verified at
    # first use; distributed.xrank_checkpoint is the recovery).  This program adds what a benchmark owes on top: both routes are
    # probed before the warm-up and the faster one is kept, and if a wait for a peer expires all ranks drop to the collective
    # together and the job is timed again (below).
    if args.scaling == 'strong':
        n0, n1 = D.shard_batch(args.batch, rank, world)
        per_rank = n1 - n0
    else:
        per_rank = args.batch
    layers = build_workload(per_rank, device, seed=12345 + 1000 * rank)    # every rank its own samples
    elems = sum(L['x'].numel() for L in layers)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    eager_step = lambda: run_step(ops, layers, group)       # noqa: E731
    launch_form = 'eager'
    if args.graph:
        # everything runs on ONE side stream - the exchange windows and the workspaces are bound to the stream of their
        # first launch, and a capture needs a stream other than the default one
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(side)

    def make_step():
        """the step as the timed region runs it: eager, or a graph captured with the exchange route in force NOW"""
        nonlocal launch_form
        if not args.graph:
            return eager_step
        eager_step()                                   # workspaces (and the exchange's sequence word) exist before the capture
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=torch.cuda.current_stream()):
            eager_step()
        launch_form = 'hip graph replay'
        return graph.replay

    step = eager_step
    def region(warm, steps):
        # every rank reaches both barriers whatever happens in between: a wait of the in-launch exchange that expired
        # surfaces as CnnqError at that exchange's next host check, on one rank first
        ok = True
        try:
            for _ in range(warm):
                step()
        except _lib.CnnqError:
            ok = False
        barrier()
        t0 = time.perf_counter()
        if ok:
            try:
                for _ in range(steps):
                    step()
            except _lib.CnnqError:
                ok = False
        barrier()
        return time.perf_counter() - t0, ok

    def agreed(flag, secs=()):
        """group-wide AND of a flag and MAX of some times: every rank takes the same decision"""
        if world == 1:
            return flag, list(secs)
        t = torch.tensor([0. if flag else 1.] + list(secs), device=device if backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0].item()) == 0., [float(v) for v in t[1:].tolist()]

    xrank_info = None
    exchanging = world > 1 or args.force_exchange
    if exchanging and D.xrank_mode() == 'auto' and D.xrank_exchange(group) is not None:
        # untimed, before the warm-up: which exchange is faster on THIS machine?  Two steps through the in-launch exchange
        # (verified at first use, just now), two through the collective; the slower one is not used for the timed steps
        tx, okx = region(1, 2)
        okx = okx and bool(D.xrank_exchange(group).healthy())
        ops._XRANK_ON = False
        ops.release_plans()
        tc, okc = region(1, 2)
        ops._XRANK_ON = True
        ops.release_plans()
        okx, (tx, tc) = agreed(okx and okc, (tx, tc))
        xrank_info = {'probe_ms_in_launch': tx / 2 * 1e3, 'probe_ms_collective': tc / 2 * 1e3}
        if not okx or tc < tx:
            D.disable_xrank(group)
    step = make_step()
    dt, ok = region(args.warmup, args.steps)
    if exchanging and D.xrank_exchange(group) is not None:
        # the job ran through the in-launch exchange (CNNQ_XRANK=auto / 1).  If a peer wait expired anywhere, every rank
        # drops to the collective together and the job is timed again
        ok, _ = agreed(ok and bool(D.xrank_exchange(group).healthy()))
        xrank_info = dict(xrank_info or {}, used=ok, healthy=ok, fell_back=not ok)
        if not ok:
            D.disable_xrank(group)
            ops.release_plans()
            step = make_step()                           # a graph captured with the closed exchange is never replayed again
            dt, ok = region(args.warmup, args.steps)
    elif xrank_info is not None:
        xrank_info.update(used=False, healthy=True, fell_back=False)      # probed, and the collective was faster (or a probe failed)
    assert ok, 'the timed region failed outside the in-launch exchange'
    total_elems = elems
    per_rank_ms = [dt * 1e3 / args.steps]
    if world > 1:
        cdev = device if backend == 'nccl' else 'cpu'
        t = torch.tensor([dt, float(elems)], device=cdev, dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        mine = torch.zeros(world, device=cdev, dtype=torch.float64)
        mine[rank] = dt * 1e3 / args.steps
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        per_rank_ms = [float(v) for v in mine.tolist()]
        dt, total_elems = float(tmax[0].item()), int(t[1].item())
    ms_per_step = dt * 1e3 / args.steps
    # the same step for about two more seconds: the 20-step region of a fresh box flatters by 2-4 % (the part warms up)
    sustained = None
    if args.sustained_secs > 0:
        ksus = max(args.steps, int(math.ceil(args.sustained_secs / (dt / args.steps))))
        dts, oks = region(0, ksus)
        if world > 1:
            tt = torch.tensor([dts], device=device if backend == 'nccl' else 'cpu', dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dts = float(tt[0].item())
        if oks:
            sustained = {'steps': ksus, 'seconds': dts, 'ms_per_step': dts * 1e3 / ksus, 'value': total_elems * ksus / dts,
                         'path_frac_hbm_peak_8B': total_elems / world * ksus / dts * BYTES_QDQ / 1e9 / HBM_PEAK_GBS,
                         'note': 'the timed step repeated for ~%.0f s right after the timed region (max over ranks); '
                                 'path_frac_hbm_peak_8B prices every element at the 8 bytes the single launch moves (what the line\'s '
                                 'path_frac_hbm_peak does for the timed steps on the single-launch routes)' % args.sustained_secs}
    if world == 1 and not args.force_exchange:
        exchange_name = 'none (1 GPU)'
    elif D.xrank_exchange(group) is not None:
        xr = D.xrank_exchange(group)
        exchange_name = ('in-launch exchange of the per-channel {min, max} through hipIpc windows (CNNQ_XRANK=%s, verified against ' % D.xrank_mode() +
                         'the collective; x is read once)%s' % (' (forced on a 1-rank group)' if args.force_exchange else ''))
        if not xr.healthy():
            exchange_name += ' - UNHEALTHY: a wait for a peer expired, results of this run are invalid'
    else:
        from cnn_quantization_amd import rccl
        direct = backend == 'nccl' and rccl.direct_comm(group) is not None
        exchange_name = '%s all_gather of the per-channel {min, max} records, one per tensor%s%s' % (
            'RCCL' if backend == 'nccl' else backend,
            ' (ncclAllGather enqueued directly on the compute stream)' if direct else ' (through torch.distributed)',
            ' (forced on a 1-rank group)' if args.force_exchange else '')
    value = total_elems * args.steps / dt
    rccl_ranks = 0
    if (world > 1 or args.force_exchange) and backend == 'nccl':
        from cnn_quantization_amd import rccl
        rccl_ranks = rccl.comm_ranks(group)          # what ncclCommCount says about the communicator actually used

    verified = verify_outputs(ops, layers, group, world)
    group_status = ops.group_status(layers[0]['x'])      # 0: no bounded wait of the in-launch exchange expired
    if world > 1:
        v = torch.tensor([1 if verified else 0], device=device if backend == 'nccl' else 'cpu', dtype=torch.int32)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        verified = bool(int(v.item()))
    single_route = (world == 1 and not args.force_exchange) or D.xrank_exchange(group) is not None
    bytes_moved = BYTES_QDQ if single_route else BYTES_PATH
    dominant, objs = roofline_objects(layers, per_rank, world, single_launch=single_route)
    out = {
        'metric': 'activation elements/sec (and % HBM peak) for per-channel int4 Q/DQ, ResNet-50 b512',
        'value': value, 'unit': 'elements/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'ms_per_step_by_rank': per_rank_ms, 'rccl_ranks': rccl_ranks,
        'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'ResNet-50 b%d conv activations (53 tensors, %.2f G elements in the job, %.2f G per GPU), '
                               'per-channel int4 dynamic min/max quantize+dequantize (-pcq_a --qtype int4)' % (
                                   args.batch * (world if args.scaling == 'weak' else 1), total_elems / 1e9, elems / 1e9),
                   'per_gpu_batch': per_rank, 'global_batch': args.batch * (world if args.scaling == 'weak' else 1),
                   'parallelism': 'batch-sharded dp%d, %s' % (world, 'per-channel extrema exchanged inside the launch' if (xrank_info or {}).get('used') else 'per-channel stats all_gather'),
                   'exchange': exchange_name, 'launch': launch_form},
        'verified': verified, 'group_status': group_status, 'xrank': None, 'box': box_id(dev_index), 'sustained': sustained,
        'path_gbs': value / world * bytes_moved / 1e9,
        'path_frac_hbm_peak': value / world * bytes_moved / 1e9 / HBM_PEAK_GBS,
        'path_bytes_moved_per_element': bytes_moved,
        'path_equiv_12B': value / world * BYTES_PATH / 1e9 / HBM_PEAK_GBS,
        'path_note': 'path_frac_hbm_peak prices every element of the step at the bytes this route reads + writes (8: one read, one '
                     'write - the single launch, also through the in-launch exchange; 12 through the collective, whose two passes '
                     'read x twice): achieved bandwidth per GPU over the whole step, launch gaps included.  path_equiv_12B is SURVEY '
                     '8(d3)\'s accounting (statistics read + Q/DQ read + write), an equivalence kept for comparison with rounds 1-5, '
                     'where it was printed as path_frac_hbm_peak: it can exceed what a 12-byte path could reach',
        'roofline': objs[dominant],
        'roofline_other_kernels': {k: v for k, v in objs.items() if k != dominant},
    }
    out['xrank'] = xrank_info       # None: the collective from the start (1 GPU, ranks sharing a GPU, CNNQ_XRANK=0, not verified)
    shard = None
    if (world > 1 or args.force_exchange) and not args.no_other_configs:
        # BASELINE configs 3 / 4 / 5 at the shard (round 6): collective - every rank takes part, rank 0 prints.  The legs run
        # through the exchange the headline ended on (in-launch: the ranks' sums meet inside the single launch; else the chain
        # around the collective) and recover like the product does (distributed.xrank_checkpoint per leg)
        import bench_other
        shard = bench_other.shard_configs(ops, device, per_rank, group, world, rank, exchange_name)
    if rank == 0:
        if shard is not None:
            out['other_configs'] = shard
        if world == 1 and not args.force_exchange:
            if not args.no_other_configs:
                out['other_configs'] = other_configs(ops, device, args.batch)
            if not args.no_cpu_baseline:
                out['cpu_baseline'] = cpu_baseline()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if world > 1 or args.force_exchange:
        from cnn_quantization_amd import rccl
        if D.xrank_exchange(group) is not None:
            D.xrank_exchange(group).close()
        rccl.close_all()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
