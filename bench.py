#!/usr/bin/env python3
"""bench.py - throughput of the per-channel int4 quantize/dequantize hot path on MI355X.

Workload (BASELINE.json configs[1]): every conv activation of one ResNet-50 forward at batch 512
(53 tensors, 5.69 G fp32 elements), quantized per channel to int4 with
dynamic min/max statistics (`-pcq_a --qtype int4`; half-range layers clamp the minimum to 0).
One "step" = the whole set once: per tensor  statistics pass (4 B/elem) -> parameters ->
fused Q/DQ (4 B read + 4 B write per element).  Inputs are synthetic per-channel Laplace
activations generated on the device (seed 12345) and are resident in HBM before the timed
region; all 53 inputs and 53 outputs are distinct buffers (45.6 GB), so nothing is re-read from
a cache across layers.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: one process per GPU, each rank holds a batch-512 shard of a global batch of 512*N
(weak scaling); per-channel statistics are made global with one all_gather of fp64 moment records
per tensor (RCCL over xGMI), the Q/DQ itself needs no communication.

Prints ONE JSON line (rank 0).  `value` = elements/s of the whole job; `roofline` = the Q/DQ
kernel's algorithmic bytes / its measured launch time against the 8 TB/s HBM3E peak;
`cpu_baseline` = the oracle (CPU restatement of the reference's op chain) timed on this host.
"""
import argparse
import json
import math
import os
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
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_QDQ = 8                  # fused Q/DQ: 4 B read + 4 B write per element
BYTES_STATS = 4                # statistics pass: one read
BYTES_PATH = 12                # dynamic min/max + Q/DQ, SURVEY.md 8(d3)


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


def time_kernel_classes(layers):
    """Device time per kernel class, measured live with HIP events recorded on the launch stream
    between the launches of ONE pass that issues exactly the sequence cnnq_pc_minmax_qdq issues
    (so cache state is the real one), one event per launch boundary so that each class is the
    duration of that kernel alone, as rocprofv3 --kernel-trace reports it.  Returns
    {class: (seconds, launches)}."""
    import ctypes
    from cnn_quantization_amd import _lib
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    recs = []
    for L in layers:
        x, y, N, C, HW = L['x'], L['y'], L['N'], L['C'], L['HW']
        G = lib.cnnq_pc_groups(N, C, HW, 1)
        pmm = torch.empty((G, 2, C), dtype=torch.float32, device=x.device)
        qp = torch.empty((3, C), dtype=torch.float32, device=x.device)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        _lib.check(lib.cnnq_pc_minmax(x.data_ptr(), N, C, HW, pmm.data_ptr(), st), 'minmax')
        e[1].record()
        _lib.check(lib.cnnq_pc_minmax_params(pmm.data_ptr(), G, C, 4, int(L['half']), qp.data_ptr(), st), 'params')
        e[2].record()
        _lib.check(lib.cnnq_pc_qdq(x.data_ptr(), y.data_ptr(), N, C, HW, qp.data_ptr(), None, None, 1, st), 'qdq')
        e[3].record()
        recs.append(e)
    torch.cuda.synchronize()
    t_mm = sum(e[0].elapsed_time(e[1]) for e in recs) * 1e-3
    t_p = sum(e[1].elapsed_time(e[2]) for e in recs) * 1e-3
    t_q = sum(e[2].elapsed_time(e[3]) for e in recs) * 1e-3
    return {'k_minmax': (t_mm, len(recs)), 'k_minmax_params': (t_p, len(recs)), 'k_qdq': (t_q, len(recs))}


def cpu_baseline(batch_sample=32, reps=3):
    """The oracle (op-for-op CPU restatement of iq.py:409-451) on the same layer set at a small
    batch: ~10-30 s of CPU work on all host cores."""
    from oracle import quant_oracle as O
    threads = torch.get_num_threads()
    xs = []
    g = torch.Generator().manual_seed(12345)
    elems = 0
    for (C, hw, half, count) in RESNET50_CONV_OUTPUTS:
        x = torch.randn((batch_sample, C, hw, hw), generator=g)
        xs.append((x, half, count))
        elems += x.numel() * count
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        for x, half, count in xs:
            for _ in range(count):
                O.act_per_channel_qdq(x, 4, half_range=half)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return dict(value=elems / best, unit='elements/s', cores=threads, kind='port',
                sample='oracle.act_per_channel_qdq over the 53 ResNet-50 conv outputs at batch %d '
                       '(%.1f M elements), best of %d, torch threads=%d' % (batch_sample, elems / 1e6, reps, threads))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=512, help='per-GPU batch (BASELINE config: 512)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

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
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

    from cnn_quantization_amd import ops, _lib
    _lib.load()                                         # fail loudly if the HIP library is missing
    layers = build_workload(args.batch, device, seed=12345 + 1000 * rank)    # every rank its own batch shard
    elems = sum(L['x'].numel() for L in layers)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_step(ops, layers, group)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step(ops, layers, group)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device if backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt * 1e3 / args.steps
    from cnn_quantization_amd import distributed as D
    exchange_name = 'none (1 GPU)' if world == 1 else (
        'peer-to-peer stores over xGMI (CNNQ_P2P_EXCHANGE=1, verified against the collective)'
        if D.p2p_exchange(group) is not None else '%s all_gather' % ('RCCL' if backend == 'nccl' else backend))
    if world > 1 and D.p2p_exchange(group) is not None and not D.p2p_exchange(group).healthy():
        exchange_name += ' - UNHEALTHY: a wait timed out, results of this run are invalid'
    value = elems * world * args.steps / dt

    # roofline of the dominant kernel (fused Q/DQ), measured live with HIP events on its stream
    time_kernel_classes(layers)                       # warm
    kc = [time_kernel_classes(layers) for _ in range(3)]
    t_qdq = min(k['k_qdq'][0] for k in kc)
    t_stats = min(k['k_minmax'][0] for k in kc)
    n_launch = kc[0]['k_qdq'][1]
    qdq_gbs = elems * BYTES_QDQ / t_qdq / 1e9
    stats_gbs = elems * BYTES_STATS / t_stats / 1e9
    out = {
        'metric': 'activation elements/sec (and % HBM peak) for per-channel int4 Q/DQ, ResNet-50 b512',
        'value': value, 'unit': 'elements/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'ResNet-50 b%d conv activations (53 tensors, %.2f G elements per GPU), per-channel '
                               'int4 dynamic min/max quantize+dequantize (-pcq_a --qtype int4)' % (args.batch, elems / 1e9),
                   'per_gpu_batch': args.batch, 'global_batch': args.batch * world,
                   'parallelism': 'batch-sharded dp%d, per-channel stats all_gather' % world,
                   'exchange': exchange_name},
        'path_gbs_algorithmic': value / world * BYTES_PATH / 1e9,
        'path_frac_hbm_peak': value / world * BYTES_PATH / 1e9 / HBM_PEAK_GBS,
        'roofline': {'bound': 'hbm', 'kernel': 'k_qdq (fused per-channel Q/DQ, 8 algorithmic B/elem)',
                     'achieved': qdq_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': qdq_gbs / HBM_PEAK_GBS,
                     'traffic': 864.3e6 if args.batch == 512 else None, 'traffic_unit': 'bytes per launch',
                     'traffic_source': 'rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes, same '
                                       'command; committed in profiles/r01_pmc_summary.md (not re-measured live)',
                     'launches_per_step': n_launch, 'avg_launch_ms': t_qdq * 1e3 / n_launch,
                     'bytes_per_launch': elems * BYTES_QDQ / n_launch},
        'roofline_stats': {'bound': 'hbm', 'kernel': 'k_minmax (per-channel exact min/max, 4 B/elem)',
                           'achieved': stats_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': stats_gbs / HBM_PEAK_GBS, 'launches_per_step': n_launch,
                           'avg_launch_ms': t_stats * 1e3 / n_launch},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
