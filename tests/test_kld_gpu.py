"""KLD calibration on the device (cnnq_kld_hist / cnnq_kld_search through ops.kld_thresholds) against
the oracle and the reference-recorded vectors.

Tiers: the 2001-bin histogram is integer work -> bit-exact.  The 994 divergences are float32 sums in
the reference (scipy entropy on float32 inputs) and fp64 sums here -> compared at rtol 2e-5 / atol 2e-7;
the chosen threshold must be the reference's, except that when the reference's own curve holds another
candidate within that tolerance of its minimum either may win (not observed on the fixtures)."""
import os
import shutil

import numpy as np
import pytest
import torch

from cnn_quantization_amd import ops
from oracle import kld_oracle as K

pytestmark = pytest.mark.gpu
CASES = ['laplace', 'relu', 'gauss_outlier', 'band', 'zeros', 'tiny', 'grid', 'negskew', 'huge_zero']
RTOL, ATOL = 2e-5, 2e-7


def check_row(x_np, out, hist, div):
    mn, mx, dmin, th, ohist, othr, odiv = K.kld_threshold(x_np, full=True)
    assert np.array_equal(hist.astype(np.int64), ohist), 'histogram differs'
    assert np.array_equal(np.isnan(div), np.isnan(odiv))
    ok = ~np.isnan(odiv)
    np.testing.assert_allclose(div[ok], odiv[ok], rtol=RTOL, atol=ATOL)
    k = int(out[2])
    assert out[0] == othr[k]                        # the edge belonging to the chosen candidate, float64-exact
    if out[0] != th:                                 # only legitimate for a near tie on the reference's curve
        assert abs(odiv[k] - dmin) <= ATOL + RTOL * abs(dmin), (out, th, odiv[k], dmin)
    return th


@pytest.mark.parametrize('name', CASES)
def test_single_row_cases(golden, name):
    g = golden('kld')
    x_np = g.np('in_' + name)
    x = torch.from_numpy(x_np).cuda()
    out, hist, div = ops.kld_thresholds(x, 1, want_parts=True)
    th = check_row(x_np, out[0].cpu().numpy(), hist[0].cpu().numpy(), div[0].cpu().numpy())
    assert th == float(g.np('th_' + name))
    assert out[0, 0].item() == float(g.np('th_' + name))


def test_batch_rows_and_unaligned():
    rng = np.random.default_rng(5)
    for shape in ((6, 3, 37, 41), (4, 8, 56, 56), (3, 70001)):
        x_np = rng.laplace(0.02, 0.9, shape).astype(np.float32)
        if len(shape) == 4:
            x_np[1] = np.maximum(x_np[1], 0)
        x = torch.from_numpy(x_np).cuda()
        out, hist, div = ops.kld_thresholds(x, want_parts=True)
        out, hist, div = out.cpu().numpy(), hist.cpu().numpy(), div.cpu().numpy()
        rows = x_np.reshape(shape[0], -1)
        for r in range(shape[0]):
            assert hist[r].sum() == rows[r].size
            check_row(rows[r], out[r], hist[r], div[r])


def test_statistic_manager_kld_column(golden, tmp_path, monkeypatch):
    from cnn_quantization_amd.inference.statistic_manager import StatisticManager
    from cnn_quantization_amd.utils.misc import Singleton
    monkeypatch.setenv('HOME', str(tmp_path))
    Singleton._instances.pop(StatisticManager, None)
    g = golden('kld')
    names = [str(s) for s in g.np('sm_names')]
    ref = dict(zip(names, g.np('sm_row')[0]))
    m = StatisticManager('kld_test', load_stats=False, kld_threshold=True, collect_err=True)
    assert m.stats_names == names
    m.save_tensor_stats(g.t('sm_in').cuda(), 'activation_linear', 'linear0_activation')
    got = dict(zip(names, m.stats['linear0_activation'][0]))
    assert got['kld_th'] == ref['kld_th']
    for c in names:
        if c.startswith('mse_') or c.startswith('cos_'):
            assert np.isnan(got[c])
        elif c != 'kld_th':
            np.testing.assert_allclose(got[c], ref[c], rtol=2e-5, atol=2e-6)
    m.__exit__()
    import pandas as pd
    df = pd.read_csv(os.path.join(str(tmp_path), 'mxt-sim', 'statistics', 'kld_test', 'kld_test_summary.csv'), index_col=0)
    assert df.loc['linear0_activation', 'mean_kld_th'] == ref['kld_th']
    Singleton._instances.pop(StatisticManager, None)


def test_distance_logger(golden, tmp_path, monkeypatch):
    from cnn_quantization_amd.inference.inference_quantization_manager import MeasureStatistics
    monkeypatch.setenv('HOME', str(tmp_path))
    g = golden('kld')
    t = g.t('sm_in').cuda()
    ms = MeasureStatistics('toy')
    with ms:
        ms.save_measure(t, 'conv0_activation')
        ms.save_measure(t * 2, 'conv0_activation')
        np.testing.assert_allclose(ms.stats['conv0_activation'], g.np('dist'), rtol=2e-6)
    import pandas as pd
    df = pd.read_csv(os.path.join(str(tmp_path), 'mxt-sim', 'distance', 'toy', 'distance.csv'))
    np.testing.assert_allclose(df['conv0_activation'].values, g.np('dist'), rtol=2e-6)
