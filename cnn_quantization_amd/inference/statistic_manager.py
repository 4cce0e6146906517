"""Per-tensor calibration statistics (CSV), mirror of
pytorch_quantizer/quantization/inference/statistic_manager.py: same API and file layout
(`~/mxt-sim/statistics/<name>/<name>_summary.csv`, columns {min,mean,max}_<stat> indexed by layer
id).  Serves the per-tensor quantizers (pooling, classifier, linear) in `-sm use` mode.

The scalar statistics of a batch are one pass of the device kernels over the tensor viewed as a
single channel.  `kld_threshold=True` adds the `kld_th` column (statistic_manager.py:80-82: the
maximum over the batch's samples of the KLD-optimal clipping threshold) from the device
histogram + search kernels (ops.kld_thresholds).  The error columns (mse_*/cos_*,
statistic_manager.py:22-30) exist in the reference's files but no caller ever passes the quantized
tensors they need (`tensors_q`), so they always hold NaN there; `collect_err=True` reproduces
those columns, a non-empty `tensors_q` is rejected."""
import os
import shutil
from pathlib import Path

import numpy as np
import pandas as pd
import torch

from .. import _lib as L
from .. import distributed as D
from .. import ops
from ..utils.misc import Singleton, sorted_nicely


def base_dir():
    return os.path.join(str(Path.home()), 'mxt-sim')


class StatisticManager(metaclass=Singleton):
    def __init__(self, folder, load_stats, stats=('max', 'min', 'std', 'mean', 'kurtosis', 'mean_abs', 'b', 'dim'),
                 batch_avg=False, kld_threshold=False, collect_err=False, group=None):
        self.name = folder
        self.group = group          # process group of a batch-sharded run (None: the default group, if any)
        self.folder = os.path.join(base_dir(), 'statistics', folder)
        self.stats_names = list(stats)
        self.collect_err = collect_err
        if collect_err:
            self.stats_names += ['mse_lowp', 'mse_gaus', 'mse_laplace', 'cos_lowp', 'cos_gaus', 'cos_laplace']
        self.kld_threshold = kld_threshold
        if kld_threshold:
            self.stats_names.append('kld_th')
        self.batch_avg = batch_avg
        self.stats = {}
        self.metadata = {}
        self.save_stats = not load_stats
        if load_stats:
            stats_file = os.path.join(self.folder, '%s_summary.csv' % self.name)
            assert os.path.exists(stats_file), stats_file
            self.stats_df = pd.read_csv(stats_file, index_col=0)
        else:
            self.stats_df = None

    def save_tensor_stats(self, tensor, tag, id, tensors_q={}, force_global_min_max=False):
        if len(tensors_q) > 0:
            raise NotImplementedError('error columns from quantized tensors: no caller in the reference')
        x = tensor.detach().contiguous()
        n = x.numel()
        # with several ranks x is this rank's batch shard: the moment records travel (ops.pc_stats), every rank
        # holds the statistics of the GLOBAL batch
        world = D.world_size(self.group)
        table, mom = ops.pc_stats(x, 1, 1, n, need_b=True, need_kurt=True, need_relu=True, group=self.group)
        if world > 1 and not D.xrank_checkpoint(self.group):
            # a wait of the in-launch exchange expired on some rank: the group is on the collective now - the table again
            table, mom = ops.pc_stats(x, 1, 1, n, need_b=True, need_kurt=True, need_relu=True, group=self.group)
        host = table.cpu().numpy()[:, 0]
        m = mom.cpu().numpy()[:, 0]
        total = m[L.MOM_COUNT]      # elements of the global batch (== n on one rank)
        vals = {'max': host[L.STAT_MAX], 'min': host[L.STAT_MIN], 'std': host[L.STAT_STD],
                'mean': host[L.STAT_MEAN], 'kurtosis': host[L.STAT_KURT], 'b': host[L.STAT_B],
                'mean_abs': np.float32((2. * m[L.MOM_SUM_RELU] - m[L.MOM_SUM]) / total), 'dim': int(total)}
        if self.batch_avg and not force_global_min_max and x.dim() > 1:
            rows, _ = ops.pc_stats(x, 1, x.shape[0], n // x.shape[0], local_only=True)
            rec = torch.stack([rows[L.STAT_MAX].double().sum(), rows[L.STAT_MIN].double().sum(),
                               torch.tensor(float(x.shape[0]), dtype=torch.float64, device=x.device)]).view(3, 1)
            if world > 1:           # sums and sample counts travel: the mean over the global batch
                rec = D.all_gather_records(rec, self.group).sum(dim=0)
            r = rec.cpu().numpy()[:, 0]
            vals['max'] = np.float32(r[0] / r[2])
            vals['min'] = np.float32(r[1] / r[2])
        for s in self.stats_names:
            if s.startswith('mse_') or s.startswith('cos_'):
                vals[s] = np.nan
        if self.kld_threshold:
            th = ops.kld_thresholds(x, x.shape[0] if x.dim() > 1 else 1)[:, 0].max().view(1, 1)
            if world > 1:           # the maximum over the samples of every rank
                th = D.all_gather_records(th, self.group).max().view(1, 1)
            vals['kld_th'] = float(th.item())
        row = np.array([[vals[s] for s in self.stats_names]], dtype=np.float64)
        if id in self.stats:
            self.stats[id] = np.concatenate([self.stats[id], row])
        else:
            self.stats[id] = row
            self.metadata[id] = tag

    def get_tensor_stat(self, id, stat, kind='mean'):
        if self.stats_df is not None:
            return self.stats_df.loc[id, '%s_%s' % (kind, stat)]
        return None

    def get_tensor_stats(self, id, kind=None):
        kind = kind or {}
        if self.stats_df is None:
            return (None,) * 6
        return tuple(self.stats_df.loc[id, '%s_%s' % (kind.get(s, 'mean'), s)]
                     for s in ('min', 'max', 'mean', 'std', 'mean_abs', 'b'))

    def __exit__(self, *args):
        if not self.save_stats:
            return
        if D.world_size(self.group) > 1:
            # every rank holds the same (global) statistics: rank 0 writes, the others wait for the files
            if D.rank(self.group) == 0:
                self._write()
            torch.distributed.barrier(group=self.group)
            return
        self._write()

    def _write(self):
        if os.path.exists(self.folder):
            shutil.rmtree(self.folder)
        os.makedirs(self.folder)
        frames = {}
        for s_id in self.stats:
            df = pd.DataFrame(columns=self.stats_names, data=self.stats[s_id])
            df.to_csv(os.path.join(self.folder, '%s.csv' % s_id), index=False)
            frames[s_id] = df
        columns = []
        for c in self.stats_names:
            columns += ['min_%s' % c, 'mean_%s' % c, 'max_%s' % c]
        summary = pd.DataFrame(columns=['internal_name'] + columns)
        for s_id in sorted_nicely(frames.keys()):
            summary.loc[s_id, 'internal_name'] = self.metadata[s_id]
            for c in self.stats_names:
                summary.loc[s_id, 'min_%s' % c] = frames[s_id][c].min()
                summary.loc[s_id, 'mean_%s' % c] = frames[s_id][c].mean()
                summary.loc[s_id, 'max_%s' % c] = frames[s_id][c].max()
            summary.loc[s_id, 'dim'] = frames[s_id]['dim'][0]
        summary.to_csv(os.path.join(self.folder, '%s_summary.csv' % self.name), index=True)
