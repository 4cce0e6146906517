#!/usr/bin/env python3
"""Golden vectors for the KLD calibration row, produced by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference).  The reference's
pytorch_quantizer/quantization/inference/kld_threshold.py is loaded unmodified by file path and
called on seeded arrays; only inputs and outputs are stored (tests/golden/kld.npz).  The per-tensor
StatisticManager's `kld_th` column (max over the samples of a batch, statistic_manager.py:80-82)
and the distance logger (distance_stats.py:22-33) are captured the same way.

numpy version note (this container: numpy 2.2.6, the reference pins none): with float32 input,
numpy >= 2 builds the 2002 histogram edges in float32 and the reference's own symmetry assertion
(kld_threshold.py:26) fails for ordinary data, i.e. the reference cannot run its float32 call path
here.  numpy 1.x - contemporary with the reference - evaluated the edges in float64.  The fixtures
therefore hand the reference the SAME float32 values widened to float64 (exact), for which the
installed numpy runs it unmodified with float64 edges; that is the semantics the oracle and the
device kernels restate (recorded in the fixture as `edge_dtype`).

    python tests/golden/make_golden_kld.py
"""
import importlib.util
import os
import sys
import tempfile
import warnings

import numpy as np

REF = os.environ.get('CNNQ_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))
os.environ['HOME'] = tempfile.mkdtemp(prefix='cnnq_golden_home_')
sys.path.insert(0, REF)


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


kld = load('ref_kld_threshold', 'pytorch_quantizer/quantization/inference/kld_threshold.py')


def laplace(rng, n, loc=0., scale=1.):
    return rng.laplace(loc, scale, n).astype(np.float32)


def cases():
    rng = np.random.default_rng(20190704)
    out = {}
    out['laplace'] = laplace(rng, 20000, 0.1, 0.7)
    out['relu'] = np.maximum(laplace(rng, 20000, -0.2, 1.0), 0).astype(np.float32)
    g = rng.normal(0.3, 0.5, 30000).astype(np.float32)
    g[123] = 40.0
    out['gauss_outlier'] = g
    out['band'] = rng.uniform(5.0, 6.0, 5000).astype(np.float32)          # nothing near zero: NaN divergences
    out['zeros'] = np.zeros(1000, np.float32)                             # degenerate range
    out['tiny'] = laplace(rng, 10)
    out['grid'] = (rng.integers(-1024, 1025, 8000) / 1024.).astype(np.float32)   # values on a lattice
    out['negskew'] = (-np.maximum(laplace(rng, 12000, 0.0, 2.0), 0) + 0.01).astype(np.float32)
    big = np.zeros(17_000_000 + 100_000, np.float32)                     # a bin count above 2^24
    big[::170] = laplace(rng, big[::170].size, 0.0, 1.5)
    out['huge_zero'] = big
    return out


def main():
    rec = {}
    for name, arr in cases().items():
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            mn, mx, div, th = kld._get_optimal_threshold(arr.astype(np.float64), num_bins=2001, num_quantized_bins=15)
            th15 = kld.get_kld_threshold_15bins(arr.astype(np.float64))
        assert th15 == th
        rec['in_' + name] = arr
        rec['min_' + name] = np.float64(mn)
        rec['max_' + name] = np.float64(mx)
        rec['div_' + name] = np.float64(div)
        rec['th_' + name] = np.float64(th)
        print('%-14s n=%9d  min %.5f max %.5f  min_div %.6g  th %.7g' % (name, arr.size, mn, mx, div, th))

    # per-tensor statistics manager with the kld_th column, and the distance logger
    import torch
    from pytorch_quantizer.quantization.inference import statistic_manager as rsm
    from pytorch_quantizer.quantization.inference import distance_stats as rds
    rng = np.random.default_rng(77)
    t = torch.from_numpy(rng.laplace(0.05, 0.8, (4, 3, 24, 24)).astype(np.float32))
    m = rsm.StatisticManager('kld_golden', load_stats=False, kld_threshold=True, collect_err=True,
                             stats=['max', 'min', 'std', 'mean', 'kurtosis', 'mean_abs', 'b', 'dim'])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m.save_tensor_stats(t.double(), 'activation_linear', 'linear0_activation')   # see the numpy note above
    rec['sm_in'] = t.numpy()
    rec['sm_names'] = np.array(m.stats_names)
    rec['sm_row'] = np.asarray(m.stats['linear0_activation'], dtype=np.float64)
    ms = rds.MeasureStatistics('kld_golden')
    ms.__enter__()
    ms.save_measure(t, 'conv0_activation')
    ms.save_measure(t * 2, 'conv0_activation')
    rec['dist'] = np.asarray(ms.stats['conv0_activation'], dtype=np.float64)
    print('sm row', dict(zip(m.stats_names, rec['sm_row'][0])))
    print('dist', rec['dist'])
    rec['edge_dtype'] = np.array('float64')
    np.savez_compressed(os.path.join(OUT, 'kld.npz'), **rec)
    print('wrote', os.path.join(OUT, 'kld.npz'), os.path.getsize(os.path.join(OUT, 'kld.npz')), 'bytes')


if __name__ == '__main__':
    main()
