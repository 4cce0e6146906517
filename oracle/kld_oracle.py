"""CPU oracle for the KLD calibration row.  TEST INFRASTRUCTURE ONLY: imported by
tests/, never by the product path.

Restates pytorch_quantizer/quantization/inference/kld_threshold.py:6-84 (the TensorRT-style
"minimise KL(P || Q_15bins) over 994 symmetric clipping thresholds of a 2001-bin histogram"
search the reference borrowed from MXNet's contrib/quantization.py) with the inner 15-bin loops
replaced by prefix sums.  Pinned: tests/test_kld_oracle_cpu.py checks it against
tests/golden/kld.npz, which was produced by running the reference file itself
(tests/golden/make_golden_kld.py; see the numpy version note there: histogram edges are float64).

Third-party arithmetic the reference calls and this restatement calls the same way:
  numpy.histogram (uniform bins: index estimate + one-step correction against the edges, right edge
  inclusive; numpy 2.2.6 here, unpinned in the reference) and scipy.stats.entropy(p, q) =
  sum(pk * log(pk / qk)) with pk, qk normalised, evaluated in float32 because both inputs are float32
  (scipy 1.15.3 here).
"""
import warnings

import numpy as np
from scipy.stats import entropy

NUM_BINS = 2001
NUM_QUANTIZED_BINS = 15


def smooth(p, eps=0.0001):
    """kld_threshold.py:86-103: zeros get eps, the non-zeros pay for it; float32 result."""
    zeros = (p == 0).astype(np.float32)
    nonzeros = (p != 0).astype(np.float32)
    n_zeros = zeros.sum()
    n_nonzeros = p.size - n_zeros
    if not n_nonzeros:
        raise ValueError('all entries are 0')
    eps1 = eps * float(n_zeros) / float(n_nonzeros)
    out = p.astype(np.float32)
    out += eps * zeros + (-eps1) * nonzeros
    return out


def histogram(arr, num_bins=NUM_BINS):
    """kld_threshold.py:19-23: symmetric range around zero from the data's extreme magnitude."""
    a = np.asarray(arr).astype(np.float64).ravel()
    mn, mx = np.min(a), np.max(a)
    th = max(abs(mn), abs(mx))
    hist, edges = np.histogram(a, bins=num_bins, range=(-th, th))
    return mn, mx, hist, edges


def divergences(hist, edges, num_quantized_bins=NUM_QUANTIZED_BINS):
    """kld_threshold.py:29-79: for i = nq//2 .. nbins//2 keep bins [zero-i, zero+i], fold the outliers
    into the end bins (P), merge the kept bins into nq equal groups and spread each group's mass
    evenly over its non-empty bins (Q; the last group's expansion stops one bin short of the end, so
    Q's last bin is always empty - the reference's `stop = -1`), smooth both, KL(P || Q)."""
    num_bins = hist.size
    zero = num_bins // 2
    half_q = num_quantized_bins // 2
    n_cand = num_bins // 2 + 1 - half_q
    thresholds = np.zeros(n_cand)
    div = np.zeros(n_cand)
    csum = np.concatenate([[0], np.cumsum(hist)])
    for i in range(half_q, num_bins // 2 + 1):
        start, stop = zero - i, zero + i + 1
        thresholds[i - half_q] = edges[stop]
        sl = hist[start:stop]
        m = sl.size
        p = sl.copy()
        p[0] += csum[start]
        p[-1] += csum[num_bins] - csum[stop]
        w = m // num_quantized_bins
        first = np.arange(num_quantized_bins) * w
        seg = np.minimum(np.arange(m) // w, num_quantized_bins - 1)
        mass = np.add.reduceat(sl, first).astype(np.int32)            # the reference accumulates in int32
        live = sl != 0
        live[-1] = False                                                # never written by the expansion
        norm = np.add.reduceat(live.astype(np.int64), first)
        with np.errstate(divide='ignore', invalid='ignore'):
            level = (mass.astype(np.float64) / norm.astype(np.float64)).astype(np.float32)
        q = np.where(live, level[seg], np.float32(0)).astype(np.float32)
        ps = smooth(p)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            try:
                qs = smooth(q)
            except ValueError:
                qs = q                                                  # entropy() then yields nan (0/0)
            div[i - half_q] = entropy(ps, qs)
    return thresholds, div


def kld_threshold(arr, num_bins=NUM_BINS, num_quantized_bins=NUM_QUANTIZED_BINS, full=False):
    """-> (min, max, min_divergence, opt_threshold); full=True adds (hist, thresholds, divergences)."""
    mn, mx, hist, edges = histogram(arr, num_bins)
    thresholds, div = divergences(hist, edges, num_quantized_bins)
    k = int(np.argmin(div))                                             # first NaN wins, else first minimum
    if full:
        return mn, mx, div[k], thresholds[k], hist, thresholds, div
    return mn, mx, div[k], thresholds[k]


def kld_threshold_batch(t):
    """statistic_manager.py:80-82: the `kld_th` statistic = max over the samples of a batch."""
    t = np.asarray(t)
    return max(kld_threshold(t[i])[3] for i in range(t.shape[0]))


def row_sumsq(t):
    """distance_stats.py:22-33: per-sample sum of squares (float32, as torch.sum(t**2, -1))."""
    t = np.asarray(t, dtype=np.float32)
    t = t.reshape(t.shape[0], -1)
    return (t.astype(np.float64) ** 2).sum(-1)
