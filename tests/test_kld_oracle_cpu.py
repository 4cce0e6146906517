"""The KLD oracle (oracle/kld_oracle.py) against vectors recorded from the REFERENCE's
inference/kld_threshold.py (tests/golden/make_golden_kld.py): min, max, the minimal divergence and the
optimal threshold must be bit-identical - the restatement replaces the reference's Python loops with
prefix sums but performs the same float32 smoothing and the same scipy entropy call."""
import numpy as np
import pytest

from oracle import kld_oracle as K

CASES = ['laplace', 'relu', 'gauss_outlier', 'band', 'zeros', 'tiny', 'grid', 'negskew', 'huge_zero']


def same(a, b):
    return (a == b) or (np.isnan(a) and np.isnan(b))


@pytest.mark.parametrize('name', CASES)
def test_threshold_matches_reference(golden, name):
    g = golden('kld')
    mn, mx, d, th = K.kld_threshold(g.np('in_' + name))
    assert mn == g.np('min_' + name) and mx == g.np('max_' + name)
    assert same(d, float(g.np('div_' + name))), (d, g.np('div_' + name))
    assert th == float(g.np('th_' + name))


def test_batch_statistic_and_distance(golden):
    g = golden('kld')
    names = [str(s) for s in g.np('sm_names')]
    row = g.np('sm_row')[0]
    assert K.kld_threshold_batch(g.np('sm_in')) == row[names.index('kld_th')]
    # the error columns exist and hold NaN (no caller passes the quantized tensors)
    for c in ('mse_lowp', 'mse_gaus', 'mse_laplace', 'cos_lowp', 'cos_gaus', 'cos_laplace'):
        assert np.isnan(row[names.index(c)])
    x = g.np('sm_in')
    d = np.concatenate([K.row_sumsq(x), K.row_sumsq(2 * x)])
    np.testing.assert_allclose(d, g.np('dist'), rtol=2e-6)


def test_fixture_records_edge_semantics(golden):
    assert str(golden('kld').np('edge_dtype')) == 'float64'
