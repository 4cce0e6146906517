"""`IntQuantizer` - the op surface of the reference's quantizer
(pytorch_quantizer/quantization/qtypes/int_quantizer.py:56-632, "iq.py" below) re-hosted on the
MI355X kernels: same constructor keys, same `__call__(tensor, id, tag, stat_id, override_att)`,
same externally mutated attributes and the same dispatch order, so the reference's manager and
`inference_sim.py` drive it unchanged.  What differs is below the surface: every branch is a
handful of launches into libcnnq_hip.so (statistics -> parameters -> fused Q/DQ on native NCHW)
with no transposed copies, no elementwise temporaries and no host synchronisation.

Only the arithmetic lives on the device; there is no CPU path (CPU tensors raise)."""
import math

import numpy as np
import torch

from .. import _lib as L
from .. import ops


def _is_pc_act(t):
    """iq.py:110,160,333: 4-D with a spatial extent."""
    return len(t.shape) > 3 and (t.shape[2] > 1 or t.shape[3] > 1)


def _to_f32_vec(v, C):
    a = np.asarray(v, dtype=np.float32).reshape(-1)
    if a.size == 1 and C > 1:
        a = np.repeat(a, C)
    return torch.from_numpy(np.ascontiguousarray(a))


class IntQuantizer:
    def __init__(self, size, params):
        # iq.py:57-90
        self.num_bits = size
        self.stochastic = False
        self.int_exp = False
        self.enforce_true_zero = True
        self.clipping = params['clipping'] if 'clipping' in params else 'no'
        self.stats_kind = params['stats_kind'] if 'stats_kind' in params else 'mean'
        self.kld = params['kld'] if 'kld' in params else False
        self.pcq_w = params['pcq_weights']
        self.pcq_a = params['pcq_act']
        self.bit_alloc_act = params['bit_alloc_act']
        self.bit_alloc_weight = params['bit_alloc_weight']
        self.bcorr_act = params['bcorr_act']
        self.bcorr_weight = params['bcorr_weight']
        self.vcorr_weight = params['vcorr_weight']
        self.bit_alloc_round = params['bit_alloc_rmode'] == 'round'
        self.bit_alloc_prior = params['bit_alloc_prior']
        self.bit_alloc_target_act = params['bit_alloc_target_act'] if params['bit_alloc_target_act'] is not None \
            else self.num_bits
        self.bit_alloc_target_weight = params['bit_alloc_target_weight'] \
            if params['bit_alloc_target_weight'] is not None else self.num_bits
        self.measure_entropy = params['measure_entropy']
        self.logger = params['logger']
        self.mtd_quant = params['mtd_quant']
        # the statistics managers are looked up lazily: `self.sm` is a class the manager may replace
        from ..inference.statistic_manager import StatisticManager
        from ..inference.statistic_manager_perchannel import StatisticManagerPerChannel
        self.sm = StatisticManagerPerChannel if params['pcq_act'] else StatisticManager
        self.force_positive = False
        self.half_range = False
        # process group over which dynamic statistics are made global (None = default group if
        # torch.distributed is initialised, single process otherwise); see distributed.py
        self.group = None
        self._table_cache = {}
        # activation bias correction hand-off (iqm.py:180-196): a layer that wants its output corrected sets
        # fuse_bcorr to the relu-first flag before the call; a per-channel path that folds the correction
        # into its own passes resets it to None and raises bcorr_fused, otherwise the layer corrects after
        self.fuse_bcorr = None
        self.bcorr_fused = False

    # ------------------------------------------------------------------ dispatch, iq.py:92-122
    def __call__(self, tensor, id, tag="", stat_id=None, override_att=None):
        if override_att is not None:
            orig_att = getattr(self, override_att[0])
            setattr(self, override_att[0], override_att[1])
        try:
            if self.kld:
                res = self.gemmlowpKldQuantize(tensor, tag, stat_id=stat_id)
            elif self.clipping != 'no':
                if self.mtd_quant:
                    res = self.mid_tread_quantize_activation(tensor, id)
                else:
                    res = self.gemmlowpClippingQuantize(tensor, id, tag, stat_id=stat_id, clip_type=self.clipping)
            elif self.pcq_w:
                if self.mtd_quant:
                    res = self.mid_tread_quantize_weights_per_channel(tensor, id)
                else:
                    res = self.gemmlowpQuantizeWeightsPerChannel(tensor, id)
            elif self.pcq_a and _is_pc_act(tensor):
                if self.mtd_quant:
                    res = self.mid_tread_quantize_activation_per_channel(tensor, id)
                else:
                    res = self.gemmlowpQuantizeActivationPerChannel(tensor, id, tag, stat_id=stat_id)
            else:
                res = self.gemmlowpMinMaxQuantize(tensor, tag, stat_id=stat_id)
        finally:
            if override_att is not None:
                setattr(self, override_att[0], orig_att)
        return res

    def __repr__(self):
        # iq.py:124-126, printed by the manager in verbose mode
        return 'IntQuantizer - [bits: {}, clipping: {}, bit_alloc_act: {}, bit_alloc_weight: {}, bit_alloc_round: {}, pcq_w: {}, pcq_a: {}, bcorr_act: {}, bcorr_weight: {}, vcorr_weight: {}, kind: {}]'\
            .format(self.num_bits, self.clipping, self.bit_alloc_act, self.bit_alloc_weight, self.bit_alloc_round,
                    self.pcq_w, self.pcq_a, self.bcorr_act, self.bcorr_weight, self.vcorr_weight, self.stats_kind)

    # ------------------------------------------------------------------ helpers
    @property
    def _positive(self):
        return bool(self.force_positive or self.half_range)

    def _stats_table(self, stat_id, C, device, rows):
        """Device table [NSTAT, C] filled from the calibration file for `stat_id`
        (`rows`: {STAT row: (stat name, kind)}); uploaded once per (stat_id, rows, device)."""
        key = (stat_id, tuple(sorted(rows.items())), str(device), id(self.sm))
        tab = self._table_cache.get(key)
        if tab is None:
            host = torch.zeros((L.NSTAT, C), dtype=torch.float32)
            for row, (stat, kind) in rows.items():
                host[row] = _to_f32_vec(self.sm().get_tensor_stat(stat_id, stat, kind=kind), C)
            tab = host.to(device)
            self._table_cache[key] = tab
        return tab

    def _take_bcorr(self):
        """The pending bias-correction request, if this call can fold it in (no entropy measurement)."""
        if self.fuse_bcorr is None or self.measure_entropy:
            return None
        flag, self.fuse_bcorr, self.bcorr_fused = self.fuse_bcorr, None, True
        return flag

    def _log_entropy(self, id, entropy, meter, weight):
        if entropy is not None and self.logger is not None:
            def log(e=entropy):
                self.logger.log_metric(id + '.entropy', float(e), step='auto', meterId=meter, weight=weight)
            eb = getattr(ops, '_ENT_BATCH', None)
            if eb is not None:
                # inside an ops.entropy_batch block (the harness wraps a forward in one): the value exists when the block's one
                # entropy launch has run - the logger is called then, in the order of the layers, as int_quantizer.py:153,179,445
                eb.after(log)
            else:
                log()

    # ------------------------------------------------------------------ per-channel activations
    def gemmlowpQuantizeActivationPerChannel(self, tensor, id, tag="", stat_id=None, min_=None, max_=None):
        """iq.py:409-451.  Dynamic statistics: one fused pipeline on the device.  With `stat_id`
        the statistics come from the calibration file (iq.py:414,421,433)."""
        if min_ is not None or max_ is not None:
            raise NotImplementedError('explicit min_/max_ are an internal hand-off of the reference '
                                      '(iq.py:352); use gemmlowpClippingQuantize')
        prior_b = self.bit_alloc_prior != 'gaus'
        use_ba = bool(self.bit_alloc_act) and self.num_bits <= 4
        table = None
        if stat_id is not None:
            C = tensor.shape[1]
            rows = {L.STAT_MAX: ('max', self.stats_kind)}
            if not self._positive:
                rows[L.STAT_MIN] = ('min', self.stats_kind)
            if use_ba:
                rows[L.STAT_B if prior_b else L.STAT_STD] = ('b' if prior_b else 'std', 'mean')
            table = self._stats_table(stat_id, C, tensor.device, rows)
        out = ops.act_qdq_per_channel(tensor, self.num_bits, positive=self._positive, clip='no',
                                      bit_alloc=self.bit_alloc_act, prior_is_b=prior_b,
                                      target=self.bit_alloc_target_act, round_mode=self.bit_alloc_round,
                                      group=self.group, stats=table, want_entropy=self.measure_entropy,
                                      bcorr=self._take_bcorr())
        if self.measure_entropy:
            out, entropy = out
            self._log_entropy(id, entropy, 'avg.entropy.act', out.numel())
        return out.view(tensor.shape)

    def gemmlowpClippingQuantize(self, tensor, id, tag="", stat_id=None, clip_type='laplace'):
        """iq.py:327-359: ACIQ clipping (laplace / gaus / <p>std), per channel when -pcq_a applies,
        otherwise per tensor with scalar statistics."""
        prior_b = self.bit_alloc_prior != 'gaus'
        if clip_type == 'mix':
            return self._clipping_mix(tensor, id, stat_id, prior_b)
        if self.pcq_a and _is_pc_act(tensor):
            table = None
            if stat_id is not None:
                C = tensor.shape[1]
                rows = {L.STAT_MIN: ('min', 'mean'), L.STAT_MAX: ('max', 'mean'), L.STAT_MEAN: ('mean', 'mean'),
                        L.STAT_B: ('b', 'mean'), L.STAT_STD: ('std', 'mean')}
                table = self._stats_table(stat_id, C, tensor.device, rows)
            out = ops.act_qdq_per_channel(tensor, self.num_bits, positive=self._positive, clip=clip_type,
                                          bit_alloc=self.bit_alloc_act, prior_is_b=prior_b,
                                          target=self.bit_alloc_target_act, round_mode=self.bit_alloc_round,
                                          group=self.group, stats=table, want_entropy=self.measure_entropy,
                                          bcorr=self._take_bcorr())
            if self.measure_entropy:
                out, entropy = out
                self._log_entropy(id, entropy, 'avg.entropy.act', out.numel())
            return out.view(tensor.shape)
        # per-tensor branch (iq.py:353-357): the whole tensor is ONE channel; bit allocation does not
        # apply (iq.py:236 requires per_channel) and delta is the range itself
        table = None
        if stat_id is not None:
            rows = {L.STAT_MIN: ('min', 'mean'), L.STAT_MAX: ('max', 'mean'), L.STAT_MEAN: ('mean', 'mean'),
                    L.STAT_B: ('b', 'mean'), L.STAT_STD: ('std', 'mean')}
            table = self._stats_table(stat_id, 1, tensor.device, rows)
        out = ops.act_qdq_per_channel(tensor, self.num_bits, positive=self._positive, clip=clip_type,
                                      bit_alloc=False, group=self.group, stats=table, whole_tensor=True)
        return out.view(tensor.shape)

    def _clipping_mix(self, tensor, id, stat_id, prior_b):
        """iq.py:310-323: `-c mix` picks the clipping value per channel from the mse_laplace / mse_gaus / mse_lowp columns of
        the statistics file (`-sm use` only: the reference looks them up by stat_id unconditionally).  A column the file
        does not have counts as NaN - every comparison False, i.e. Laplace clipping, which is also what the files the
        reference itself writes (NaN error columns) amount to."""
        if stat_id is None:
            raise ValueError("clipping 'mix' needs a statistics file (-sm use): int_quantizer.py:311-313 reads mse_* by stat_id")
        pc = self.pcq_a and _is_pc_act(tensor)
        C = tensor.shape[1] if pc else 1
        rows = {L.STAT_MIN: ('min', 'mean'), L.STAT_MAX: ('max', 'mean'), L.STAT_MEAN: ('mean', 'mean'),
                L.STAT_B: ('b', 'mean'), L.STAT_STD: ('std', 'mean')}
        table = self._stats_table(stat_id, C, tensor.device, rows)
        mse = torch.full((3, C), float('nan'), dtype=torch.float32)
        for r, name in enumerate(('mse_laplace', 'mse_gaus', 'mse_lowp')):
            try:
                v = self.sm().get_tensor_stat(stat_id, name, 'mean')
            except KeyError:
                v = None
            if v is not None:
                mse[r] = _to_f32_vec(v, C)
        out = ops.act_qdq_mix(tensor, self.num_bits, table, mse, positive=self._positive,
                              bit_alloc=self.bit_alloc_act and pc, prior_is_b=prior_b, target=self.bit_alloc_target_act,
                              round_mode=self.bit_alloc_round, whole_tensor=not pc, want_entropy=self.measure_entropy and pc)
        if self.measure_entropy and pc:
            out, entropy = out
            self._log_entropy(id, entropy, 'avg.entropy.act', out.numel())
        return out.view(tensor.shape)

    # ------------------------------------------------------------------ weights
    def gemmlowpQuantizeWeightsPerChannel(self, tensor, id, min_=None, max_=None):
        """iq.py:453-476: rows of [OFM, IFM*K*K]; weights are replicated, never sharded."""
        out = ops.act_qdq_per_channel(tensor, self.num_bits, positive=False, clip='no',
                                      bit_alloc=self.bit_alloc_weight, prior_is_b=False,
                                      target=self.bit_alloc_target_weight, round_mode=self.bit_alloc_round,
                                      per_channel_dim=0, group=False, want_entropy=self.measure_entropy)
        if self.measure_entropy:
            out, entropy = out
            self._log_entropy(id, entropy, 'avg.entropy.weight', out.numel())
        return out.view(tensor.shape)

    # ------------------------------------------------------------------ per-tensor paths
    def gemmlowpMinMaxQuantize(self, tensor, tag="", stat_id=None):
        """iq.py:361-379."""
        if stat_id is not None:
            kmin, kmax = ('mean', 'mean') if self.stats_kind == 'mean' else ('min', 'max')
            min_ = self.sm().get_tensor_stat(stat_id, 'min', kmin)
            max_ = self.sm().get_tensor_stat(stat_id, 'max', kmax)
            if self._positive:
                min_ = 0
            return self.__gemmlowpQuantize__(tensor, max_ - min_, min_)
        avg = ('activation' in tag and 'classifier' not in tag)
        return ops.minmax_qdq_per_tensor(tensor, self.num_bits, avg_over_batch=avg, zero_min=self._positive,
                                         int_exp=self.int_exp, enforce_true_zero=self.enforce_true_zero,
                                         group=self.group).view(tensor.shape)

    def gemmlowpKldQuantize(self, tensor, tag="", stat_id=None):
        """iq.py:478-486: KLD threshold from the (per-tensor) calibration file as the clipping value."""
        min_ = self.sm().get_tensor_stat(stat_id, 'min', 'mean')
        max_ = self.sm().get_tensor_stat(stat_id, 'max', 'mean')
        kld_th = self.sm().get_tensor_stat(stat_id, 'kld_th', 'mean')
        mean = self.sm().get_tensor_stat(stat_id, 'mean', 'mean')
        range_, offset = self.alpha2DeltaOffset(kld_th, max_, min_, mean)
        return self.__gemmlowpQuantize__(tensor, range_, offset)

    def alpha2DeltaOffset(self, alpha, max_value, min_value, mean, clip2max=False):
        """iq.py:284-300 for host scalars / numpy vectors (stats-file driven callers); the
        per-channel device version lives in cnnq_pc_params."""
        alpha, max_value, min_value, mean = (np.asarray(v.cpu() if isinstance(v, torch.Tensor) else v)
                                             for v in (alpha, max_value, min_value, mean))
        if self._positive:
            delta = np.maximum(mean, 0) + alpha
            if clip2max:
                delta = np.minimum(delta, max_value)
            offset = 0
        else:
            delta = 2 * alpha
            if clip2max:
                delta = np.minimum(delta, max_value - min_value)
            offset = np.maximum(min_value, mean - alpha)
        return delta, offset

    def __gemmlowpQuantize__(self, tensor, delta, offset):
        """iq.py:605-614: scalar range/offset -> the per-tensor GEMMLOWP kernel."""
        from .. import int_quantization
        delta, offset = float(delta), float(offset)
        preserve_zero = self.enforce_true_zero and (offset + delta) > 0 and offset < 0
        return int_quantization.float2gemmlowp(tensor.contiguous(), delta, offset, self.num_bits, self.int_exp,
                                               preserve_zero, None)

    # ------------------------------------------------------------------ mid-tread (config 5)
    def mid_tread_quantize_weights_per_channel(self, tensor, id):
        """iq.py:147-156."""
        out, entropy = ops.mid_tread_qdq(tensor, self.bit_alloc_target_weight, clip=False, sym=True,
                                         per_channel_dim=0, group=False, want_entropy=self.measure_entropy)
        self._log_entropy(id, entropy, 'avg.entropy.weight', out.numel())
        return out.view(tensor.shape)

    def mid_tread_quantize_activation(self, tensor, id):
        """iq.py:158-168."""
        if self.pcq_a and _is_pc_act(tensor):
            return self.mid_tread_quantize_activation_per_channel(tensor, id)
        out, _ = ops.mid_tread_qdq(tensor, self.bit_alloc_target_act, clip=True, sym=not self._positive,
                                   whole_tensor=True, group=self.group, want_entropy=self.measure_entropy)
        return out.view(tensor.shape)

    def mid_tread_quantize_activation_per_channel(self, tensor, id):
        """iq.py:170-183."""
        out, entropy = ops.mid_tread_qdq(tensor, self.bit_alloc_target_act, clip=True, sym=not self._positive,
                                         group=self.group, want_entropy=self.measure_entropy)
        self._log_entropy(id, entropy, 'avg.entropy.act', tensor.numel())
        return out.view(tensor.shape)


def int_quantizer(qtype, quant_params):
    """iq.py:626-632: 'int4' -> IntQuantizer(4, params); bare 'int' -> 32 bits."""
    if len(qtype) > len('int'):
        size = int(qtype[len('int'):])
    else:
        size = 32
    return IntQuantizer(size, quant_params)


__all__ = ['IntQuantizer', 'int_quantizer', 'math']
