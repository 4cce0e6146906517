"""Per-channel calibration statistics: `-sm collect` computes them, `-sm use` serves them.
Mirror of pytorch_quantizer/quantization/inference/statistic_manager_perchannel.py ("smpc.py"):
same constructor, `save_tensor_stats`, `get_tensor_stat`, `__exit__` and the SAME on-disk format
(`~/mxt-sim/statistics/per_channel/<name>/<name>_statistics_perchannel_summary.pkl`: a pickled
dict layer-id -> DataFrame with float32 columns {min,mean,max}_<stat>, one row per channel), so
files written by either implementation are interchangeable.

The seven statistics of one batch come from two coalesced passes over the activation on the
device (cnnq_pc_moments + cnnq_pc_absdev) and ONE device->host copy of a [7, C] table, instead
of a transposed copy, nine full-tensor reductions and seven synchronising copies
(smpc.py:51-79,112)."""
import os
import pickle
import shutil
from pathlib import Path

import numpy as np
import pandas as pd
import torch

from .. import _lib as L
from .. import distributed as D
from .. import ops
from ..utils.misc import Singleton

SAVE_FULL_STATS = False
_ROW = {'max': L.STAT_MAX, 'min': L.STAT_MIN, 'std': L.STAT_STD, 'mean': L.STAT_MEAN,
        'kurtosis': L.STAT_KURT, 'b': L.STAT_B, 'std_pos': L.STAT_STD_POS}


def base_dir():
    return os.path.join(str(Path.home()), 'mxt-sim')


class StatisticManagerPerChannel(metaclass=Singleton):
    def __init__(self, folder, load_stats, stats=('max', 'min', 'std', 'mean', 'kurtosis', 'b', 'std_pos'),
                 batch_avg=False, collect_err=False, group=None):
        self.name = folder
        self.folder = os.path.join(base_dir(), 'statistics/per_channel', folder)
        self.stats_names = list(stats)
        if collect_err:
            raise NotImplementedError('collect_err (mse/cos columns) is a diagnostic outside the hot path')
        self.collect_err = collect_err
        self.batch_avg = batch_avg
        self.group = group
        self.save_stats = not load_stats
        if load_stats:
            stats_file = os.path.join(self.folder, '%s_statistics_perchannel_summary.pkl' % self.name)
            assert os.path.exists(stats_file), stats_file
            with open(stats_file, 'rb') as f:
                self.stats = pickle.load(f)
        else:
            self.stats = {}

    def save_tensor_stats(self, tensor, tag, id, tensors_q={}, force_global_min_max=False):
        # FC and 1x1-spatial outputs are not per-channel quantized (smpc.py:47-48)
        if len(tensor.shape) < 3 or (tensor.shape[2] == 1 and tensor.shape[3] == 1):
            return
        N, C = tensor.shape[0], tensor.shape[1]
        HW = tensor.numel() // (N * C)
        x = tensor.detach().contiguous()
        table, _ = ops.pc_stats(x, N, C, HW, need_b='b' in self.stats_names,
                                need_kurt='kurtosis' in self.stats_names,
                                need_relu='std_pos' in self.stats_names, group=self.group)
        if D.world_size(self.group) > 1 and not D.xrank_checkpoint(self.group):
            # sharded: a wait of the in-launch exchange expired on some rank (the table is NaN there); the group is on the
            # collective now - this layer's table again, before anything of it is recorded
            table, _ = ops.pc_stats(x, N, C, HW, need_b='b' in self.stats_names, need_kurt='kurtosis' in self.stats_names,
                                    need_relu='std_pos' in self.stats_names, group=self.group)
        if self.batch_avg and not force_global_min_max:
            # mean over the batch of the per-sample extrema (smpc.py:72,78): rows = (n, c) pairs; with several ranks
            # the sums and the sample counts travel, so every rank holds the mean over the GLOBAL batch
            rows, _ = ops.pc_stats(x, 1, N * C, HW, local_only=True)
            table = table.clone()
            smax = rows[L.STAT_MAX].view(N, C).double().sum(dim=0)
            smin = rows[L.STAT_MIN].view(N, C).double().sum(dim=0)
            cnt = torch.full_like(smax, float(N))
            if D.world_size(self.group) > 1:
                rec = D.all_gather_records(torch.stack([smax, smin, cnt]), self.group).sum(dim=0)
                smax, smin, cnt = rec[0], rec[1], rec[2]
            table[L.STAT_MAX] = (smax / cnt).float()
            table[L.STAT_MIN] = (smin / cnt).float()
        host = table.cpu().numpy()          # the only synchronisation of this call
        layer = self.stats.setdefault(id, {})
        for sn in self.stats_names:
            st = host[_ROW[sn]].copy()
            layer[sn] = st if sn not in layer else np.vstack([layer[sn], st])

    def get_tensor_stat(self, id, stat, kind='mean'):
        if self.stats is not None:
            return self.stats[id]['%s_%s' % (kind, stat)]
        return None

    def __exit__(self, *args):
        if not self.save_stats:
            return
        if D.world_size(self.group) > 1:
            # every rank holds the same (global) statistics: rank 0 writes, the others wait for the files
            if D.rank(self.group) != 0:
                torch.distributed.barrier(group=self.group)
                return
            self._write()
            torch.distributed.barrier(group=self.group)
            return
        self._write()

    def _write(self):
        if os.path.exists(self.folder):
            shutil.rmtree(self.folder)
        os.makedirs(self.folder)
        if SAVE_FULL_STATS:
            with open(os.path.join(self.folder, 'statistics_perchannel.pkl'), 'wb') as f:
                pickle.dump(self.stats, f)
        self._save_summary()

    def _save_summary(self):
        """min / mean / max over the collected batches per channel (smpc.py:152-174)."""
        columns = []
        for c in self.stats_names:
            columns += ['min_%s' % c, 'mean_%s' % c, 'max_%s' % c]
        summary = {}
        for layer in self.stats:
            df = pd.DataFrame(columns=columns)
            for s in self.stats_names:
                if s in self.stats[layer]:
                    t = self.stats[layer][s]
                    many = len(t.shape) > 1
                    df['min_%s' % s] = t.min(axis=0) if many else [t.min(axis=0)]
                    df['mean_%s' % s] = t.mean(axis=0) if many else [t.mean(axis=0)]
                    df['max_%s' % s] = t.max(axis=0) if many else [t.max(axis=0)]
            summary[layer] = df
        with open(os.path.join(self.folder, '%s_statistics_perchannel_summary.pkl' % self.name), 'wb') as f:
            pickle.dump(summary, f)
