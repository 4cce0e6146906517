#!/usr/bin/env python3
"""Harness counterpart of the reference's driver inference/inference_sim.py.

Same quantization flags, same flow (inference_sim.py:375-390 and InferenceModel.__init__ :131-229):

    qparams = get_params()                      # CLI -> {'int': {...18 keys...}, 'qmanager': {...}}
    with QM(args, qparams):                     # patches torch.nn layer classes
        model = build(arch)                     # in-repo ResNet-50 / VGG-16 (random weights)
        set_node_names; mark_before_relu; absorb_bn -> QM().bn_folding = True
        QM().quantize_model(model)              # weights
        for batch in synthetic batches: model(batch)   # every layer output is quantized on the way

What is different, because ImageNet, pretrained weights and torchvision are not available here:
inputs are synthetic, weights are randomly initialised (seed 12345, the reference's seed), so no
accuracy is reported - the harness reports the per-layer time and throughput of the
quantization path instead, and serves as the end-to-end drop-in check of the manager plumbing.

    python -m cnn_quantization_amd.harness.inference_sim -a resnet50 -b 512 -pcq_w -pcq_a --qtype int4 -qw int4
    python -m cnn_quantization_amd.harness.inference_sim -a resnet50 -b 512 -pcq_w -pcq_a --qtype int4 -qw int4 \\
           -c laplace -baa -baw -bcw
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from cnn_quantization_amd.harness import models  # noqa: E402
from cnn_quantization_amd.inference.inference_quantization_manager import QuantizationManagerInference as QM  # noqa: E402
from cnn_quantization_amd.inference.inference_quantization_manager import reset_layer_counters  # noqa: E402
from cnn_quantization_amd.utils import model_prep  # noqa: E402
from cnn_quantization_amd.utils.misc import Singleton  # noqa: E402


def build_parser():
    """The quantization-relevant subset of inference_sim.py:52-112, same names and defaults."""
    p = argparse.ArgumentParser(description='MI355X harness of the post-training quantization simulator')
    p.add_argument('--arch', '-a', default='resnet50', choices=sorted(models.MODELS))
    p.add_argument('-b', '--batch-size', default=256, type=int)
    p.add_argument('--batches', default=1, type=int, help='synthetic batches to run')
    p.add_argument('--image-size', default=224, type=int)
    p.add_argument('--seed', default=12345, type=int)
    p.add_argument('--qtype', default=None)
    p.add_argument('--qweight', '-qw', default='int8')
    p.add_argument('--q_off', action='store_true')
    p.add_argument('--clipping', '-c', default='no')
    p.add_argument('--stats_mode', '-sm', default='no')
    p.add_argument('--stats_kind', '-sk', default='mean')
    p.add_argument('--stats_folder', '-sf', default=None)
    p.add_argument('--stats_batch_avg', '-sba', action='store_true')
    p.add_argument('--kld_threshold', '-kld', action='store_true')
    p.add_argument('--measure_stats', '-ms', action='store_true')
    p.add_argument('--per_channel_quant_weights', '-pcq_w', action='store_true')
    p.add_argument('--per_channel_quant_act', '-pcq_a', action='store_true')
    p.add_argument('--bit_alloc_act', '-baa', action='store_true')
    p.add_argument('--bit_alloc_weight', '-baw', action='store_true')
    p.add_argument('--bit_alloc_rmode', '-bam', default='round')
    p.add_argument('--bit_alloc_prior', '-bap', default='gaus')
    p.add_argument('--bit_alloc_target_act', '-bata', type=float, default=None)
    p.add_argument('--bit_alloc_target_weight', '-batw', type=float, default=None)
    p.add_argument('--bias_corr_act', '-bca', action='store_true')
    p.add_argument('--bias_corr_weight', '-bcw', action='store_true')
    p.add_argument('--var_corr_weight', '-vcw', action='store_true')
    p.add_argument('--measure_entropy', '-me', action='store_true')
    p.add_argument('--mid_thread_quant', '-mtq', action='store_true')
    p.add_argument('--preserve_zero', '-pz', action='store_true')
    p.add_argument('--rho_act', '-ra', default=None, type=float)
    p.add_argument('--rho_weight', '-rw', default=None, type=float)
    p.add_argument('--no-bn-folding', action='store_true')
    p.add_argument('--verbose', action='store_true')
    p.add_argument('--sharded', action='store_true',
                   help='one process per GPU (start under torchrun / torch.distributed.run): every rank takes its shard of each '
                        'batch (inference_sim.py:196-200 splits the batch over DataParallel replicas; here the statistics are '
                        'those of the GLOBAL batch, DESIGN.md section 6) and the ranks check the in-launch exchange together '
                        'after every forward (distributed.xrank_checkpoint), redoing the batch through the collective if a '
                        'wait for a peer expired')
    p.add_argument('--graph', action='store_true',
                   help='also capture the quantized forward into a HIP graph and time its replay (small batches: '
                        'removes the per-launch host overhead; the library only enqueues kernels, so it is capturable)')
    return p


class MeterLogger:
    """The slice of utils/mllog.py the quantizers use: weighted running averages keyed by meterId
    (`avg.entropy.act`, `avg.entropy.weight`; int_quantizer.py:153,179,445,472)."""

    def __init__(self):
        self.metters = {}

    def log_metric(self, key, value, step=None, meterId=None, weight=1.):
        if meterId is None:
            return
        s, w = self.metters.get(meterId, (0., 0.))
        self.metters[meterId] = (s + float(value) * weight, w + weight)

    def averages(self):
        return {k: s / w for k, (s, w) in self.metters.items() if w}


def get_params(args, logger=None):
    """inference_sim.py:345-372."""
    return {
        'int': {
            'clipping': args.clipping, 'stats_kind': args.stats_kind, 'true_zero': args.preserve_zero,
            'kld': args.kld_threshold, 'pcq_weights': args.per_channel_quant_weights,
            'pcq_act': args.per_channel_quant_act, 'bit_alloc_act': args.bit_alloc_act,
            'bit_alloc_weight': args.bit_alloc_weight, 'bit_alloc_rmode': args.bit_alloc_rmode,
            'bit_alloc_prior': args.bit_alloc_prior, 'bit_alloc_target_act': args.bit_alloc_target_act,
            'bit_alloc_target_weight': args.bit_alloc_target_weight, 'bcorr_act': args.bias_corr_act,
            'bcorr_weight': args.bias_corr_weight, 'vcorr_weight': args.var_corr_weight, 'logger': logger,
            'measure_entropy': args.measure_entropy, 'mtd_quant': args.mid_thread_quant,
        },
        'qmanager': {'rho_act': args.rho_act, 'rho_weight': args.rho_weight},
    }


class QuantTimer:
    """Wraps the manager's quantize_instant with HIP events: per-call device time of the
    quantization path, keyed by (id, tag)."""

    def __init__(self, qm):
        self.qm, self.rows, self._orig = qm, [], qm.op_manager.quantize_instant
        qm.op_manager.quantize_instant = self._timed

    def _timed(self, tensor, id, tag="", *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = self._orig(tensor, id, tag, *a, **k)
        e1.record()
        half = a[1] if len(a) > 1 else k.get('half_range', False)
        self.rows.append((str(id), tag, tuple(tensor.shape), tensor.numel(), e0, e1, bool(half)))
        return out

    def summary(self):
        torch.cuda.synchronize()
        return [(i, t, s, n, e0.elapsed_time(e1) * 1e-3, h) for (i, t, s, n, e0, e1, h) in self.rows]


def _init_ranks():
    """The default process group of a --sharded run, from torchrun's environment: RCCL (backend nccl) with one GPU per rank,
    gloo when the ranks share a GPU (test rigs)."""
    import torch.distributed as dist
    if dist.is_initialized():
        return
    rank, local = int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    one_each = torch.cuda.device_count() >= int(os.environ.get('LOCAL_WORLD_SIZE', world))
    torch.cuda.set_device(local if one_each else 0)
    dist.init_process_group('nccl' if one_each else 'gloo', rank=rank, world_size=world)


def run(args, quiet=False):
    """Returns a dict with the per-call timing rows and totals."""
    from cnn_quantization_amd import distributed as D
    from cnn_quantization_amd import ops as ops_mod
    sharded = bool(getattr(args, 'sharded', False))
    redone = 0
    if sharded:
        _init_ranks()
        D.set_xrank_recovery('checkpoint')           # nothing raises between checkpoints: the ranks stay in lock-step
    torch.manual_seed(args.seed)
    Singleton.reset()
    reset_layer_counters()
    logger = MeterLogger()
    dev = torch.device('cuda')
    with QM(args, get_params(args, logger)) as qm:
        model = models.MODELS[args.arch]()
        model_prep.set_node_names(model)
        if args.arch.startswith('resnet'):
            models.mark_before_relu(model)
        model = model.to(dev).eval()
        if not args.no_bn_folding and args.qtype is not None:   # folded BN layers are skipped by the patched class only
            model_prep.absorb_bn(model)
            qm.bn_folding = True
        qm.quantize_model(model)                     # weights (verbose=True inside, like the reference)
        timer = QuantTimer(qm)
        qm.verbose = args.verbose
        g = torch.Generator(device=dev).manual_seed(args.seed)
        t_fwd = []
        with torch.no_grad():
            for _ in range(args.batches + 1):        # first batch is warm-up (MIOpen find, allocator)
                x = torch.randn(args.batch_size, 3, args.image_size, args.image_size, generator=g, device=dev)
                if sharded:                          # the same batch on every rank (same seed); this rank's samples of it
                    n0, n1 = D.shard_batch(args.batch_size, D.rank(qm.group), D.world_size(qm.group))
                    x = x[n0:n1].contiguous()
                for attempt in range(2):
                    timer.rows.clear()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    if args.measure_entropy:
                        # -me: the entropies of all layers' codes in ONE launch at the end of the forward (ops.entropy_batch)
                        with ops_mod.entropy_batch():
                            out = model(x)
                    else:
                        out = model(x)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    # the synchronisation point of the forward: did a wait of the in-launch exchange expire on ANY rank?  Then
                    # the group is on the collective from here on and this batch is run again (its outputs were NaN somewhere)
                    if not sharded or D.xrank_checkpoint(qm.group):
                        break
                    redone += 1
                t_fwd.append(dt)
        rows = timer.summary()
        t_graph = None
        if args.graph and qm.stats_mode.name != 'collect_stats' and not args.measure_entropy:
            qm.op_manager.quantize_instant = timer._orig           # no event pairs inside the capture
            static_x = x.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                model(static_x)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(graph):
                static_out = model(static_x)
            graph.replay()
            torch.cuda.synchronize()
            # not asserted equal: MIOpen's convolutions are not run-to-run reproducible on this stack (an fp32
            # ResNet-50 forward differs from itself by ~1e-8), and quantization amplifies a flipped code
            graph_diff = float((static_out - out).abs().max())
            t_graph = 1e9
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                graph.replay()
                torch.cuda.synchronize()
                t_graph = min(t_graph, time.perf_counter() - t0)
    act = [r for r in rows if 'conv' in r[0]]
    tot_t = sum(r[4] for r in rows)
    res = dict(rows=rows, quant_seconds=tot_t, forward_seconds=t_fwd[-1], conv_elements=sum(r[3] for r in act),
               conv_quant_seconds=sum(r[4] for r in act), entropy=logger.averages(), graph_seconds=t_graph, graph_max_abs_diff=graph_diff if t_graph is not None else None,
               output_finite=bool(torch.isfinite(out).all()), batches_redone=redone, logits=out)
    if not quiet:
        print('%-22s %-22s %-22s %10s %10s %9s' % ('id', 'tag', 'shape', 'Melem', 'us', 'Gelem/s'))
        for (i, t, s, n, dt, _h) in rows:
            print('%-22s %-22s %-22s %10.2f %10.1f %9.1f' % (i, t, 'x'.join(map(str, s)), n / 1e6, dt * 1e6, n / dt / 1e9))
        if res['conv_quant_seconds'] > 0:
            print('forward %.2f ms of which quantization path %.2f ms; conv activations: %.3f G elements in %.2f ms = '
                  '%.1f G elem/s' % (res['forward_seconds'] * 1e3, tot_t * 1e3, res['conv_elements'] / 1e9,
                                     res['conv_quant_seconds'] * 1e3,
                                     res['conv_elements'] / res['conv_quant_seconds'] / 1e9))
        else:   # -sm collect: statistics are gathered, nothing is quantized
            print('forward %.2f ms (statistics collection, no quantization)' % (res['forward_seconds'] * 1e3))
        if t_graph is not None:
            print('HIP-graph replay of the same forward: %.2f ms (eager %.2f ms); max |logit difference| to the eager '
                  'run %.3g (MIOpen convolutions are not run-to-run reproducible)' % (
                      t_graph * 1e3, res['forward_seconds'] * 1e3, res['graph_max_abs_diff']))
        for k, v in res['entropy'].items():
            print('Average bit rate: {} - {}'.format(k, v))
    return res


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit('the harness needs an MI355X (there is no CPU path)')
    return run(args)


if __name__ == '__main__':
    main()
