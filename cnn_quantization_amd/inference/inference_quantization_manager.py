"""Layer-substitution plumbing: the counterpart of the reference's
pytorch_quantizer/quantization/inference/inference_quantization_manager.py ("iqm.py") and of the
base class in pytorch_quantizer/quantization/quantization_manager.py:10-58.

What the reference's driver (inference/inference_sim.py) needs and gets here, unchanged:

    with QM(args, qparams):                 # process-wide singleton used as a context manager
        model = build_model()               # torch.nn.Conv2d/Linear/... now resolve to *WithId
        QM().quantize_model(model)          # weights: per-channel Q/DQ + bias/variance correction
        out = model(x)                      # every layer output goes through quantize_instant

plus `QM().bn_folding`, `.verbose`, `.reset_counters()`, `.disable()`, `.set_8bit_list()`,
`.reload()`.  Layer code reaches the manager as `QMI()` with no arguments.

Behaviour that is part of the contract and reproduced on purpose:
  * ids are handed out at layer CONSTRUCTION by per-class counters (`conv%d_activation`, ...);
  * `before_relu` on a layer selects half-range quantization; an output with 1000 channels is the
    classifier;
  * AvgPool and BatchNorm call quantize_instant with (tensor, tag) only, so their tag lands in
    the `id` slot, the tag is empty and the DEFAULT quantizer (int8 + CLI flags) is used
    (iqm.py:96-99, 275-278);
  * the int4 "keep conv0 at 8 bit" list only applies when a stat_id is given (iqm.py:551-555).

All arithmetic is in libcnnq_hip.so through cnn_quantization_amd.ops / IntQuantizer."""
import os
from enum import Enum
from itertools import count

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..qtypes import DummyQuantizer, int_quantizer
from ..utils.misc import Singleton
from .statistic_manager import StatisticManager
from .statistic_manager_perchannel import StatisticManagerPerChannel

FUSED_RELU_ARCHS = ('alexnet', 'vgg16', 'vgg16_bn', 'inception_v3')


class StatsMode(Enum):
    no_stats = 1
    collect_stats = 2
    use_stats = 3


class MeasureStatistics:
    """Runtime distance logger (`-ms`), mirror of inference/distance_stats.py:17-59: per layer id the
    per-sample sum of squares of every batch, written as ~/mxt-sim/distance/<folder>/distance.csv
    (one column per layer).  The reduction is the device moments kernel (ops.row_sumsq)."""

    def __init__(self, folder):
        from pathlib import Path
        self.enabled = False
        self.folder = os.path.join(str(Path.home()), 'mxt-sim', 'distance', folder)
        self.stats = {}
        self.stats_names = ['dist']

    def save_measure(self, tensor, id):
        d = ops.row_sumsq(tensor.detach().contiguous(), tensor.shape[0]).cpu().numpy().astype(np.float64)
        self.stats[id] = np.concatenate([self.stats[id], d]) if id in self.stats else d

    def __enter__(self):
        self.enabled = True
        self.stats.clear()
        return self

    def __exit__(self, *args):
        if self.enabled and len(self.stats) > 0:
            self.enabled = False
            import shutil
            import pandas as pd
            if os.path.exists(self.folder):
                shutil.rmtree(self.folder)
            os.makedirs(self.folder)
            cols = list(self.stats.keys())
            data = np.array([self.stats[c] for c in cols]).transpose()
            pd.DataFrame(data=data, columns=cols).to_csv(os.path.join(self.folder, 'distance.csv'), index=False)


# --------------------------------------------------------------------------------- patched layers
def _route(layer, out, out_id, tag, *, shifted=False, half_range=False, collect_tag=None, force_global=False,
           bcorr=None):
    """Common tail of every patched layer's forward: collect statistics, or quantize the output
    with (use) / without (no) calibration statistics.  `shifted` reproduces the reference's calls
    that pass (tensor, tag) positionally, leaving the tag in the id slot and the tag empty."""
    qm = QMI()
    if not qm.enabled:
        return out
    if qm.stats_mode is StatsMode.collect_stats:
        kw = dict(force_global_min_max=True) if force_global else {}
        qm.stats_manager.save_tensor_stats(out, tag if collect_tag is None else collect_tag, out_id, **kw)
        return out
    stat_id = out_id if qm.stats_mode is StatsMode.use_stats else None
    if shifted:
        return qm.quantize_instant(out, tag, stat_id=stat_id, half_range=half_range, verbose=qm.verbose)
    return qm.quantize_instant(out, out_id, tag, stat_id=stat_id, half_range=half_range, verbose=qm.verbose,
                               bcorr=bcorr)


class ReLUWithId(nn.ReLU):
    _id = count(0)


class MaxPool2dWithId(nn.MaxPool2d):
    _id = count(0)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.id = next(MaxPool2dWithId._id)

    def forward(self, input):
        out = super().forward(input)
        return _route(self, out, 'maxpool%d_out' % self.id, 'activation_pooling')


class AvgPool2dWithId(nn.AvgPool2d):
    _id = count(0)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.id = next(AvgPool2dWithId._id)

    def forward(self, input):
        out = super().forward(input)
        tag = 'activation_classifier' if out.shape[1] == 1000 else 'activation_pooling'
        return _route(self, out, 'avgpool%d_out' % self.id, tag, shifted=True)


class BatchNorm2dWithId(nn.BatchNorm2d):
    _id = count(0)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.id = next(BatchNorm2dWithId._id)

    def forward(self, input):
        qm = QMI()
        if qm.bn_folding and hasattr(self, 'absorbed'):
            return input
        out = super().forward(input)
        out = _route(self, out, 'bn%d_activation' % self.id, 'activation', shifted=True,
                     half_range=hasattr(self, 'before_relu'))
        if qm.measure_stats.enabled:
            qm.measure_stats.save_measure(out, 'bn%d_activation' % self.id)
        return out


class Conv2dWithId(nn.Conv2d):
    _id = count(0)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.id = next(Conv2dWithId._id)

    def forward(self, input):
        out = super().forward(input)
        qm = QMI()
        act_id = 'conv%d_activation' % self.id
        if qm.enabled:
            tag = 'activation_classifier' if out.shape[1] == 1000 else 'activation'
            half = hasattr(self, 'before_relu')
            raw = out
            correct = qm.stats_mode is StatsMode.use_stats and qm.bcorr_act
            relu_first = half or qm.op_manager.fused_relu
            out = _route(self, raw, act_id, tag, half_range=half,
                         collect_tag=getattr(self, 'internal_name', act_id), bcorr=relu_first if correct else None)
            if correct and not qm.op_manager.last_bcorr_fused:
                # iqm.py:180-196: shift the positive outputs so the channel sums match the fp32 ones (the
                # per-channel quantizers fold this into their own passes, see IntQuantizer.fuse_bcorr)
                out = ops.act_bias_correction_(raw, out.contiguous(), relu_first, group=qm.group)
        if qm.measure_stats.enabled:
            qm.measure_stats.save_measure(out, act_id)
        return out


class LinearWithId(nn.Linear):
    _id = count(0)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.id = next(LinearWithId._id)

    def forward(self, input):
        out = super().forward(input)
        qm = QMI()
        act_id = 'linear%d_activation' % self.id
        if qm.enabled:
            classifier = self.weight.shape[0] == 1000
            tag = 'activation_classifier' if classifier else 'activation_linear'
            half = hasattr(self, 'before_relu') and not classifier
            out = _route(self, out, act_id, tag, half_range=half, force_global=classifier)
        if qm.measure_stats.enabled:
            qm.measure_stats.save_measure(out, act_id)
        return out


def reset_layer_counters():
    """Restart the per-class id counters (a new experiment in the same process; the reference runs
    one model per process and never needs this)."""
    for cls in (ReLUWithId, MaxPool2dWithId, AvgPool2dWithId, BatchNorm2dWithId, Conv2dWithId, LinearWithId):
        cls._id = count(0)


_PATCHED = {'Linear': LinearWithId, 'Conv2d': Conv2dWithId, 'BatchNorm2d': BatchNorm2dWithId,
            'MaxPool2d': MaxPool2dWithId, 'AvgPool2d': AvgPool2dWithId, 'ReLU': ReLUWithId}


# --------------------------------------------------------------------------------- quantizer table
# tag -> (bit-width source, attribute overrides applied after construction); iqm.py:407-476.
# 'act' = args.qtype, 'weight' = args.qweight, otherwise a literal qtype.  Setting `bit_alloc` on a
# quantizer (as the reference does) touches no attribute the quantizer reads; it is kept so the
# objects look the same from outside.
_PER_TENSOR = dict(pcq_w=False, pcq_a=False, sm=StatisticManager)
_TABLE = (
    ('activation_classifier', 'int8', dict(clipping='no', kld=False, stats_kind='max', measure_entropy=False,
                                           **_PER_TENSOR)),
    ('weight', 'weight', dict(pcq_a=False, clipping='no', kld=False, bit_alloc=False, stats_kind='max')),
    ('weight_classifier', 'int8', dict(pcq_a=False, clipping='no', kld=False, bit_alloc=False, stats_kind='max',
                                       measure_entropy=False)),
    ('ignored', 'int8', dict(clipping='no', kld=False, **_PER_TENSOR)),
    ('activation', 'act', dict(force_positive='fused_relu', pcq_w=False)),
    ('activation_linear', 'act', dict(force_positive='fused_relu', **_PER_TENSOR)),
    ('activation_pooling', 'int8', dict(clipping='no', kld=False, bit_alloc=False, measure_entropy=False,
                                        **_PER_TENSOR)),
)


class TruncationOpManagerInference:
    def __init__(self, args, qparams, group=None):
        self.verbose = False
        self.ignore_ids = []
        self.last_bcorr_fused = False
        self._orig = {name: getattr(nn, name) for name in _PATCHED}
        qm = qparams.get('qmanager', {}) if isinstance(qparams, dict) else {}
        self.rho_act, self.rho_weight = qm.get('rho_act'), qm.get('rho_weight')
        self.fp32_clip = self.rho_act is not None or self.rho_weight is not None
        arch = args.arch
        self.fused_relu = arch is not None and (arch in FUSED_RELU_ARCHS or 'squeezenet' in arch)
        self.group = group
        if args.qtype is not None:
            self.quantize = True
            self.quantizers = {}
            if 'bfloat' in args.qtype:
                raise NotImplementedError('bfloat quantizers are not part of the int hot path')
            self._fill(args.qtype, qparams, args.qweight)
            self.quantizer_default = self._load('int8', qparams)

    def _load(self, qtype, qparams):
        family = qtype.rstrip('1234567890')
        if family != 'int':
            raise NotImplementedError('quantizer family %r' % family)
        q = int_quantizer(qtype, qparams[family] if family in qparams else {})
        q.group = self.group
        return q

    def _fill(self, qtype, qparams, qweight):
        for tag, source, overrides in _TABLE:
            name = {'act': qtype, 'weight': qweight}.get(source, source)
            if tag == 'weight' and qweight == 'f32':
                self.quantizers[tag] = DummyQuantizer()
                continue
            q = self._load(name, qparams)
            for key, val in overrides.items():
                setattr(q, key, self.fused_relu if val == 'fused_relu' else val)
            self.quantizers[tag] = q
        self.quantizers['bias'] = DummyQuantizer()

    def __exit__(self, *args):
        pass

    def get_quantizer(self, tag, tensor=None):
        return self.quantizers.get(tag, self.quantizer_default)

    def set_8bit_list(self, ignore_list):
        self.ignore_ids = ignore_list

    def enable(self):
        for name, cls in _PATCHED.items():
            setattr(nn, name, cls)

    def disable(self):
        for name, cls in self._orig.items():
            setattr(nn, name, cls)

    def quantize_instant(self, tensor, id, tag="", stat_id=None, half_range=False, override_att=None,
                         verbose=False, bcorr=None):
        ignored = stat_id is not None and any(l == stat_id for l in self.ignore_ids)
        q = self.get_quantizer('ignored' if ignored else tag)
        q.half_range = half_range
        if verbose:
            print("Quantize {0:21} | Id - {1:18} | {2:} | {3:}".format(tag, str(stat_id), str(q), str(tensor.device)))
        # bcorr (not in the reference's signature): the calling layer wants iqm.py:180-196 applied to the
        # result; quantizers that can fold it into their passes do and report so
        self.last_bcorr_fused = False
        if bcorr is None or not hasattr(q, 'fuse_bcorr'):
            return q(tensor, id, tag, stat_id, override_att)
        q.fuse_bcorr, q.bcorr_fused = bool(bcorr), False
        try:
            res = q(tensor, id, tag, stat_id, override_att)
            self.last_bcorr_fused = q.bcorr_fused
        finally:
            q.fuse_bcorr, q.bcorr_fused = None, False
        return res


# --------------------------------------------------------------------------------- the manager
class QuantizationManagerInference(metaclass=Singleton):
    def __init__(self, args, qparams, group=None):
        self.args = args
        self.verbose = False
        self.group = group
        self.quantize = args.qtype is not None
        self.disable_quantization = args.q_off
        self.op_manager = self.createTruncationManager(args, qparams)
        self.enabled = False
        self.bn_folding = False
        self.bcorr_act = args.bias_corr_act
        self.bcorr_weight = args.bias_corr_weight
        self.vcorr_weight = args.var_corr_weight
        sf = args.stats_folder if args.stats_folder is not None else args.arch
        if args.kld_threshold:
            sf += '_kld_' + args.qtype
        self.stats_manager = None
        if args.stats_mode == 'collect':
            print("Collecting statistics...")
            self.stats_mode = StatsMode.collect_stats
            if args.per_channel_quant_act:
                self.stats_manager = StatisticManagerPerChannel(sf, load_stats=False, batch_avg=args.stats_batch_avg,
                                                                group=group)
            else:
                self.stats_manager = StatisticManager(sf, load_stats=False, batch_avg=args.stats_batch_avg,
                                                      kld_threshold=args.kld_threshold)
        elif args.stats_mode == 'use':
            self.stats_mode = StatsMode.use_stats
            if args.per_channel_quant_act:
                StatisticManagerPerChannel(sf, load_stats=True)
            StatisticManager(sf, load_stats=True)
        else:
            self.stats_mode = StatsMode.no_stats
        self.measure_stats = MeasureStatistics(args.arch)
        if getattr(args, 'measure_stats', False):
            self.measure_stats.__enter__()

    # context manager / switches (quantization_manager.py:14-37)
    def __enter__(self):
        self.enable()
        return self

    def __exit__(self, *args):
        self.op_manager.__exit__(args)
        if self.stats_manager is not None:
            self.stats_manager.__exit__()
        if self.measure_stats is not None:
            self.measure_stats.__exit__()
        self.disable()

    def enable(self):
        if self.quantize:
            self.enabled = not self.disable_quantization
            self.op_manager.enable()

    def disable(self):
        self.enabled = False
        self.op_manager.disable()

    def reload(self, args, qparams={}):
        self.disable()
        self.op_manager = self.createTruncationManager(args, qparams)
        self.enable()

    def reduce_logging_verbosity(self):
        self.op_manager.verbose = False

    def createTruncationManager(self, args, qparams):
        op_manager = TruncationOpManagerInference(args, qparams, group=self.group)
        if args.qtype == 'int4':
            op_manager.set_8bit_list(['conv%d_activation' % i for i in (0,)])
        return op_manager

    def quantize_instant(self, tensor, id, tag="", stat_id=None, half_range=False, override_att=None,
                         verbose=False, bcorr=None):
        return self.op_manager.quantize_instant(tensor, id, tag, stat_id, half_range, override_att, verbose, bcorr)

    def set_8bit_list(self, ignore_ids):
        self.op_manager.set_8bit_list(ignore_ids)

    def reset_counters(self):
        ReLUWithId._id = count(0)

    def quantize_model(self, model):
        """Quantize every Conv2d / Linear weight in place (iqm.py:352-393): per-channel Q/DQ (first
        layer, the one with 3 input channels, at 8 bit; Inception's first two convs likewise), then the
        optional variance / bias correction per output channel."""
        if self.args.stats_mode == 'collect' or not self.quantize:
            return
        inception = type(model).__name__ == 'Inception3'
        for n, m in model.named_modules():
            if isinstance(m, torch.nn.Conv2d):
                eight = m.weight.shape[1] == 3 or (inception and n in ('Conv2d_1a_3x3.conv', 'Conv2d_2a_3x3.conv'))
                weight_q = QMI().quantize_instant(m.weight, n + '.weight', "weight",
                                                  override_att=('num_bits', 8) if eight else None, verbose=True)
            elif isinstance(m, torch.nn.Linear):
                tag = 'weight_classifier' if m.weight.shape[0] == 1000 else 'weight'
                weight_q = QMI().quantize_instant(m.weight, n + '.weight', tag, verbose=True)
            else:
                continue
            if self.vcorr_weight or self.bcorr_weight:
                weight_q = ops.weight_correction(m.weight.data, weight_q, vcorr=self.vcorr_weight,
                                                 bcorr=self.bcorr_weight)
            m.weight.data = weight_q


QMI = QuantizationManagerInference
