"""CPU oracle for the per-channel quantize / clip / dequantize hot path.

TEST INFRASTRUCTURE ONLY.  This file restates, op for op on CPU torch/numpy, the arithmetic
of the reference (submission2019/cnn-quantization) for the rows of SURVEY.md section 8(a).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it; the product package ``cnn_quantization_amd`` never does (its compute runs in
the HIP library and fails loudly when that library is missing).

Parity is PINNED: ``tests/golden/make_golden.py`` imports the reference itself in the build
container and stores input/output vectors under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function below against them bit for bit.

Every function cites the reference lines it follows (paths relative to the reference root,
``pytorch_quantizer/quantization/qtypes/int_quantizer.py`` abbreviated ``iq.py``,
``pytorch_quantizer/quantization/inference/inference_quantization_manager.py`` ``iqm.py``,
``pytorch_quantizer/quantization/inference/statistic_manager_perchannel.py`` ``smpc.py``).
All tensors are fp32, activations are NCHW.
"""
import math

import numpy as np
import torch

F32_MAX = float(np.finfo(np.float32).max)

# ACIQ multipliers, keyed by bit width (iq.py:81-85).
ALPHA_GAUS = {1: 1.24, 2: 1.71, 3: 2.15, 4: 2.55, 5: 2.93, 6: 3.28, 7: 3.61, 8: 3.92}
ALPHA_GAUS_POS = {1: 1.71, 2: 2.15, 3: 2.55, 4: 2.93, 5: 3.28, 6: 3.61, 7: 3.92, 8: 4.2}
ALPHA_LAPLACE = {0: 1.05, 1: 1.86, 2: 2.83, 3: 3.89, 4: 5.03, 5: 6.2, 6: 7.41, 7: 8.64, 8: 9.89}
ALPHA_LAPLACE_POS = {0: 1.86, 1: 2.83, 2: 3.89, 3: 5.02, 4: 6.2, 5: 7.41, 6: 8.64, 7: 9.89, 8: 11.16}


def _is_pc_act(x):
    """4-D activation with a spatial extent: the per-channel predicate of iq.py:110,160,333."""
    return x.dim() > 3 and (x.shape[2] > 1 or x.shape[3] > 1)


def _as_f32(v):
    """iq.py:21-25 ``to_cuda``: tensors pass through, everything else becomes an fp32 tensor."""
    if isinstance(v, torch.Tensor):
        return v
    return torch.tensor(np.asarray(v), dtype=torch.float32)


def _channel_rows(x):
    """[N,C,H,W] -> [C, N*H*W] (the transpose+copy of iq.py:427-428,534-535; smpc.py:51-52)."""
    return x.detach().transpose(0, 1).contiguous().view(x.shape[1], -1)


# ----------------------------------------------------------------------------- stats (a2, a3)
def _stats_over_last_dim(t, names, avg_over_batch):
    out = {}
    for s in names:
        if s == 'max':
            v = t.max(dim=-1)[0]
        elif s == 'min':
            v = t.min(dim=-1)[0]
        elif s == 'mean':
            v = t.mean(dim=-1)
        elif s == 'b':
            v = torch.mean(torch.abs(t - t.mean(dim=-1).unsqueeze(-1)), dim=-1)
        elif s == 'std':
            v = torch.std(t, dim=-1, unbiased=True)
        else:
            raise KeyError(s)
        out[s] = torch.mean(v, dim=0) if avg_over_batch else v
    return out


def act_stats_perchannel(x, names, avg_over_batch=False):
    """Row a2, iq.py:530-555.  avg_over_batch=False: over [C, N*H*W]; True: over H*W per
    (n, c) and then the mean over n."""
    t = x.view(x.shape[0], x.shape[1], -1) if avg_over_batch else _channel_rows(x)
    return _stats_over_last_dim(t, names, avg_over_batch)


def act_stats(x, names, avg_over_batch=False):
    """Row a3, iq.py:507-528.  Per tensor, or per sample followed by the mean over samples."""
    if avg_over_batch:
        return _stats_over_last_dim(x.view(x.shape[0], -1), names, True)
    t = x.reshape(-1)
    out = {}
    for s in names:
        if s == 'max':
            out[s] = t.max()
        elif s == 'min':
            out[s] = t.min()
        elif s == 'mean':
            out[s] = t.mean()
        elif s == 'b':
            out[s] = torch.mean(torch.abs(t - t.mean()))
        elif s == 'std':
            out[s] = t.std(unbiased=True)
    return out


# ----------------------------------------------------------------------------- core Q/DQ (a4)
def qdq_core(t, delta, offset, num_bits=None, bit_alloc=None, return_parts=False):
    """Row a4, iq.py:557-603 (enforce_true_zero is hard-wired True, iq.py:62).

    ``t`` is [C, M] with per-row ``delta``/``offset`` ([C] or 0-dim), or any shape with
    0-dim ``delta``/``offset`` (the per-tensor use at iq.py:357).  Returns the dequantized
    tensor; with ``return_parts`` also (codes fp32, scale, zero_point, qmax)."""
    delta = _as_f32(delta)
    offset = _as_f32(offset)
    if bit_alloc is None:
        qmax = 2. ** num_bits - 1.
        scale = delta / (qmax - 0.)
    else:
        qmax = 2. ** bit_alloc - 1.
        scale = torch.where(qmax > 0, delta / (qmax - 0.), torch.tensor([0.]))
    scale = torch.max(scale, torch.tensor([1e-8]))
    zero_point = torch.round(0. - offset / scale)
    q = torch.div(t.detach(), scale.unsqueeze(-1))
    q = torch.add(q, zero_point.unsqueeze(-1))
    if bit_alloc is None:
        q = q.clamp(0., qmax).round()
    else:
        qm = qmax.view(qmax.numel(), 1)
        q = torch.where(q.gt(qm), qm, q)
        q = q.clamp(min=0.).round()
    codes = q.clone()
    y = torch.mul(torch.add(q, -zero_point.unsqueeze(-1)), scale.unsqueeze(-1)).view(t.shape)
    if return_parts:
        return y, codes, scale, zero_point, qmax
    return y


# ----------------------------------------------------------------------------- bit allocation (a8)
def bits_alloc(alpha, num_bits, round_mode=False):
    """iq.py:381-391 (paper eq. 11).  ``num_bits`` may be a Python float."""
    B = len(alpha) * 2 ** num_bits
    p = alpha ** (2. / 3)
    bins = (B * p) / p.sum()
    bits = torch.round(torch.log2(bins)) if round_mode else torch.ceil(torch.log2(bins))
    bits[bits < 0] = 0
    bits[bits > 8] = 8
    return bits


def bits_alloc_fixed_target(alpha, num_bits, round_mode=False):
    """iq.py:393-407: at most 10 corrections of the target until mean(bits) is within 0.01."""
    goal = num_bits
    target = goal
    delta = 1.
    it = 0
    bits = None
    while abs(2 * delta) > 0.01 and it < 10:
        it += 1
        bits = bits_alloc(alpha, target, round_mode)
        delta = (goal - bits.mean()) / 2
        target += delta.item()
    return bits


# ----------------------------------------------------------------------------- ACIQ (a7, a9)
def aciq_factor(bits, clip_type, positive):
    """Table lookups of iq.py:248,251 (laplace) and iq.py:264 (gaus) for one bit width."""
    if clip_type == 'laplace':
        return (ALPHA_LAPLACE_POS if positive else ALPHA_LAPLACE)[bits]
    if clip_type == 'gaus':
        return (ALPHA_GAUS_POS if positive else ALPHA_GAUS)[bits]
    raise KeyError(clip_type)


def alpha_laplace(b, num_bits, positive, bit_alloc=None):
    """iq.py:227-253 given the statistic ``b`` (and the per-channel bit allocation when
    bit allocation is active: iq.py:247-249)."""
    if bit_alloc is not None:
        fac = np.array([aciq_factor(int(nb.item()), 'laplace', positive) for nb in bit_alloc])
        fac = torch.tensor(fac, dtype=torch.float32)
    else:
        fac = aciq_factor(num_bits, 'laplace', positive)
    return _as_f32(b) * fac


def alpha_to_delta_offset(alpha, max_value, min_value, mean, positive):
    """Row a9, iq.py:284-300 (numpy arithmetic; ``clip2max`` is never set on this path)."""
    alpha = alpha.numpy() if isinstance(alpha, torch.Tensor) else alpha
    max_value = max_value.numpy() if isinstance(max_value, torch.Tensor) else max_value
    min_value = min_value.numpy() if isinstance(min_value, torch.Tensor) else min_value
    mean = mean.numpy() if isinstance(mean, torch.Tensor) else mean
    if positive:
        return np.maximum(np.array(mean), 0) + alpha, 0
    return 2 * alpha, np.maximum(min_value, mean - alpha)


# ----------------------------------------------------------------------------- activations (a5, a6)
def act_per_channel_qdq(x, num_bits, half_range=False, force_positive=False, bit_alloc_act=False,
                        bit_alloc_prior='gaus', bit_alloc_target=None, bit_alloc_round=True,
                        min_=None, max_=None, prior_stat=None, return_parts=False):
    """Row a5, iq.py:409-451 with stat_id=None unless min_/max_/prior_stat are supplied (those
    stand for the stats-file lookups of iq.py:414,421,433)."""
    if min_ is None:
        if force_positive or half_range:
            min_ = 0
        else:
            min_ = act_stats_perchannel(x, ['min'])['min']
    min_ = _as_f32(min_)
    if max_ is None:
        max_ = act_stats_perchannel(x, ['max'])['max']
    max_ = _as_f32(max_)
    N, C, H, W = x.shape
    t = _channel_rows(x)
    bits = None
    if bit_alloc_act and num_bits <= 4:
        if prior_stat is None:
            prior = 'std' if bit_alloc_prior == 'gaus' else 'b'
            prior_stat = act_stats_perchannel(x, [prior])[prior]
        target = bit_alloc_target if bit_alloc_target is not None else num_bits
        bits = bits_alloc_fixed_target(_as_f32(prior_stat), target, bit_alloc_round)
    y, codes, scale, zp, qmax = qdq_core(t, max_ - min_, min_, num_bits=num_bits, bit_alloc=bits,
                                         return_parts=True)
    y = y.view(C, N, H, W).transpose(0, 1).contiguous()
    if return_parts:
        codes = codes.view(C, N, H, W).transpose(0, 1).contiguous()
        return y, dict(codes=codes, scale=scale, zero_point=zp, qmax=qmax, bit_alloc=bits,
                       min=min_, max=max_)
    return y


def act_clipping_qdq(x, num_bits, clip_type='laplace', half_range=False, force_positive=False,
                     pcq_a=True, bit_alloc_act=False, bit_alloc_prior='gaus', bit_alloc_target=None,
                     bit_alloc_round=True, return_parts=False):
    """Row a6, iq.py:327-359 with stat_id=None (dynamic statistics)."""
    positive = force_positive or half_range
    if pcq_a and _is_pc_act(x):
        st = act_stats_perchannel(x, ['min', 'max'])
        mean = act_stats_perchannel(x, ['mean'], avg_over_batch=True)['mean']
        min_v, max_v = st['min'], st['max']
        # get_alpha (iq.py:302-309) per channel
        bits = None
        if clip_type == 'laplace':
            b = act_stats_perchannel(x, ['b'])['b']
            if bit_alloc_act and num_bits <= 4:
                prior = 'std' if bit_alloc_prior == 'gaus' else 'b'
                pr = act_stats_perchannel(x, [prior])[prior]
                target = bit_alloc_target if bit_alloc_target is not None else num_bits
                bits = bits_alloc_fixed_target(pr, target, bit_alloc_round)
            alpha = alpha_laplace(b, num_bits, positive, bits)
        elif clip_type == 'gaus':
            std = act_stats_perchannel(x, ['std'])['std']
            alpha = std * aciq_factor(num_bits, 'gaus', positive)
        elif 'std' in clip_type:
            std = act_stats_perchannel(x, ['std'])['std']
            alpha = float(clip_type.replace('std', '')) * std
        else:
            raise KeyError(clip_type)
        rng, off = alpha_to_delta_offset(alpha, max_v, min_v, mean, positive)
        off = _as_f32(off)
        rng = _as_f32(rng)
        out = act_per_channel_qdq(x.contiguous(), num_bits, half_range, force_positive, bit_alloc_act,
                                  bit_alloc_prior, bit_alloc_target, bit_alloc_round,
                                  min_=off, max_=off + rng, return_parts=return_parts)
        if return_parts:
            out[1].update(alpha=alpha, range=rng, offset=off, mean=mean, stat_min=min_v, stat_max=max_v)
        return out
    # per-tensor branch, iq.py:353-357
    st = act_stats(x, ['min', 'max', 'mean'])
    if clip_type == 'laplace':
        alpha = act_stats(x, ['b'])['b'] * aciq_factor(num_bits, 'laplace', positive)
    elif clip_type == 'gaus':
        alpha = act_stats(x, ['std'])['std'] * aciq_factor(num_bits, 'gaus', positive)
    else:
        alpha = float(clip_type.replace('std', '')) * act_stats(x, ['std'])['std']
    rng, off = alpha_to_delta_offset(float(alpha), float(st['max']), float(st['min']), float(st['mean']),
                                     positive)
    y = qdq_core(x.contiguous(), _as_f32(rng), _as_f32(off), num_bits=num_bits)
    if return_parts:
        return y, dict(range=rng, offset=off)
    return y


# ----------------------------------------------------------------------------- weights (a10, a11)
def act_clipping_mix_qdq(x, num_bits, stats, mse, half_range=False, force_positive=False, bit_alloc_act=False,
                         bit_alloc_prior='gaus', bit_alloc_target=None, bit_alloc_round=True, return_parts=False):
    """clip_type == 'mix' of iq.py:310-323 on the per-channel `-sm use` route (iq.py:327-352): `stats` holds the file's
    per-channel `mean_*` columns {min, max, mean, b, std}, `mse` the columns {laplace, gaus, lowp}.  Per channel the
    clipping value is the Gaussian one where mse_gaus < mse_laplace, else the Laplace one, and the min/max half range
    where mse_lowp < mse_gaus (comparisons with NaN are False: a file whose error columns are NaN - all the reference's
    own collection ever writes - gives plain Laplace clipping)."""
    positive = force_positive or half_range
    f32 = lambda v: np.asarray(v, dtype=np.float32)
    mn, mx, mean, b, std = (f32(stats[k]) for k in ('min', 'max', 'mean', 'b', 'std'))
    bits = None
    if bit_alloc_act and num_bits <= 4:
        prior = std if bit_alloc_prior == 'gaus' else b
        target = bit_alloc_target if bit_alloc_target is not None else num_bits
        bits = bits_alloc_fixed_target(torch.from_numpy(prior), target, bit_alloc_round)
    a_lap = alpha_laplace(torch.from_numpy(b), num_bits, positive, bits)
    a_lap = a_lap.numpy() if isinstance(a_lap, torch.Tensor) else f32(a_lap)
    a_gaus = std * np.float32(aciq_factor(num_bits, 'gaus', positive))
    a_lowp = (mx - mn) / 2
    with np.errstate(invalid='ignore'):
        alpha = np.where(f32(mse['gaus']) < f32(mse['laplace']), a_gaus, a_lap)
        alpha = np.where(f32(mse['lowp']) < f32(mse['gaus']), a_lowp, alpha)
    rng, off = alpha_to_delta_offset(alpha, mx, mn, mean, positive)
    off = _as_f32(off)
    rng = _as_f32(rng)
    out = act_per_channel_qdq(x.contiguous(), num_bits, half_range, force_positive, bit_alloc_act, bit_alloc_prior,
                              bit_alloc_target, bit_alloc_round, min_=off, max_=off + rng,
                              prior_stat=torch.from_numpy(std if bit_alloc_prior == 'gaus' else b), return_parts=return_parts)
    if return_parts:
        out[1].update(alpha=alpha, range=rng, offset=off)
    return out


def weights_per_channel_qdq(w, num_bits, bit_alloc_weight=False, bit_alloc_target=None,
                            bit_alloc_round=True, return_parts=False):
    """Row a10, iq.py:453-476: rows are the output channels of [OFM, IFM*K*K]."""
    t = w.view(w.shape[0], -1)
    min_ = t.min(-1)[0]
    max_ = t.max(-1)[0]
    bits = None
    if bit_alloc_weight and num_bits <= 4:
        target = bit_alloc_target if bit_alloc_target is not None else num_bits
        bits = bits_alloc_fixed_target(t.std(-1), target, bit_alloc_round)
    y, codes, scale, zp, qmax = qdq_core(t, max_ - min_, min_, num_bits=num_bits, bit_alloc=bits,
                                         return_parts=True)
    y = y.view(w.shape)
    if return_parts:
        return y, dict(codes=codes.view(w.shape), scale=scale, zero_point=zp, qmax=qmax, bit_alloc=bits)
    return y


def weight_correction(w, w_q, vcorr=False, bcorr=False):
    """Row a11, iqm.py:374-391: per-output-channel variance then mean correction."""
    shape1 = (-1, 1, 1, 1) if w_q.dim() == 4 else (-1, 1)
    if vcorr or bcorr:
        bias_q = w_q.view(w_q.shape[0], -1).mean(-1).view(shape1)
        bias_orig = w.view(w.shape[0], -1).mean(-1).view(shape1)
    if vcorr:
        eps = torch.tensor([1e-8])
        var_corr = w.view(w.shape[0], -1).std(dim=-1) / (w_q.view(w_q.shape[0], -1).std(dim=-1) + eps)
        w_q = (w_q - bias_q) * var_corr.view(shape1) + bias_q
    if bcorr:
        w_q = w_q - bias_q + bias_orig
    return w_q


def act_bias_correction(out, out_q, relu_first):
    """Row a12, iqm.py:188-196 (only with -sm use and -bca)."""
    if relu_first:
        out = torch.nn.functional.relu(out)
    temp = _channel_rows(out)
    q_bias = temp.sum(-1) - _channel_rows(out_q).sum(-1)
    count = (temp > 0).sum(-1).type(q_bias.dtype)
    q_bias = q_bias / (count + torch.tensor([1e-8]))
    return out_q + (out_q > 0).type(out_q.dtype) * q_bias.view(1, q_bias.numel(), 1, 1)


# ----------------------------------------------------------------------------- per-tensor kernel (a13)
def float2gemmlowp(x, rng, offset, num_bits, int_exp, enforce_true_zero, noise=None):
    """Row a13: numpy restatement of kernels/gemmlowp.cu:8-25 (device loop) and :30-45 (host
    wrapper).  fp32 arithmetic with one rounding per operation (the a*b-c of gemmlowp.cu:23 is
    evaluated without FMA contraction), roundf = round half away from zero,
    fminf/fmaxf drop NaNs.  ``range <= 0`` returns the input itself (gemmlowp.cu:31-32)."""
    rng = np.float32(rng)
    offset = np.float32(offset)
    if rng <= 0:
        return x
    xin = x.detach().contiguous().numpy().astype(np.float32, copy=False)
    qmax = np.float32((1 << num_bits) - 1)
    scale = np.float32(rng / qmax)
    if int_exp:
        scale = np.float32(2.0 ** int(math.ceil(np.log2(scale))))
    with np.errstate(all='ignore'):
        zp = np.float32(-offset / scale)
        zp = np.float32(np.copysign(np.floor(np.abs(zp) + np.float32(0.5)), zp))  # roundf
        shift = zp if enforce_true_zero else np.float32(-offset)
        if enforce_true_zero:
            t = (xin / scale).astype(np.float32) + shift
        else:
            t = ((xin + shift).astype(np.float32) / scale).astype(np.float32)
        if noise is not None:
            t = t + noise.detach().numpy().astype(np.float32, copy=False)
        t = np.fmin(t, qmax)
        t = np.fmax(t, np.float32(0.))
        t = roundf_np(t)
        if enforce_true_zero:
            y = ((t - shift).astype(np.float32) * scale).astype(np.float32)
        else:
            y = ((t * scale).astype(np.float32) - shift).astype(np.float32)
    return torch.from_numpy(y).view(x.shape)


def roundf_np(t):
    """C roundf on fp32 arrays: half away from zero, exact (no double rounding)."""
    t = t.astype(np.float32, copy=False)
    a = np.abs(t)
    f = np.floor(a)
    r = np.where(a - f >= np.float32(0.5), f + np.float32(1.), f).astype(np.float32)
    r = np.where(np.isfinite(t), np.copysign(r, t), t)
    return r.astype(np.float32)


def gemmlowp_minmax_qdq(x, num_bits, tag='', half_range=False, force_positive=False,
                        enforce_true_zero=True, int_exp=False, min_=None, max_=None):
    """iq.py:361-379 + iq.py:605-614: dynamic min/max (per sample then batch mean for
    'activation' tags that are not classifier tags), then the per-tensor kernel."""
    if min_ is None or max_ is None:
        st = act_stats(x, ['min', 'max'], avg_over_batch=('activation' in tag and 'classifier' not in tag))
        min_, max_ = st['min'], st['max']
    if force_positive or half_range:
        min_ = 0
    delta, offset = max_ - min_, min_
    preserve_zero = bool(enforce_true_zero and (offset + delta) > 0 and offset < 0)
    return float2gemmlowp(x, float(delta), float(offset), num_bits, int_exp, preserve_zero)


# ----------------------------------------------------------------------------- mid-tread (a14, a15)
def shannon_entropy(t):
    """Row a15, utils/entropy.py:6-17: entropy in bits of the value histogram of the whole tensor."""
    pk = torch.unique(t.flatten(), return_counts=True)[1]
    probs = pk.float() / pk.sum()
    probs[probs == 0] = 1
    return (-probs * torch.log2(probs)).sum()


def omega_alloc(sigma, target_bins):
    """iq.py:128-135 (paper eq. 10): real-valued bin allocation."""
    B = len(sigma) * target_bins
    p = sigma ** (2. / 3)
    return (B * p) / p.sum()


def alpha_mult_from_tables(omega, sym, omega_table, alpha_table):
    """iq.py:137-145 with GPU semantics: ``omega`` itself is NOT modified."""
    om = omega.detach().numpy().copy()
    if not sym:
        om *= 2
    i = omega_table.searchsorted(om)
    inc = (alpha_table[i] - alpha_table[i - 1]) / (omega_table[i] - omega_table[i - 1])
    return alpha_table[i] - inc * (omega_table[i] - om)


def mid_tread_core(t, target, clip, sym, omega_table=None, alpha_table=None, want_entropy=False,
                   return_parts=False, stats=None):
    """Row a14, iq.py:185-225 on a [R, M] matrix.  stats (tests only): dict of per-row `std`, `mean`, `b` to use INSTEAD of
    the row statistics of t - the arithmetic downstream of the statistics on a table somebody else reduced (the device's)."""
    std = t.std(-1) if stats is None else stats['std']
    omega = omega_alloc(std, target_bins=(2 ** target)).round()
    mu = None
    if clip:
        am = t.new_tensor(alpha_mult_from_tables(omega, sym, omega_table, alpha_table))
        mu = t.mean(dim=-1) if stats is None else stats['mean']
        b = torch.mean(torch.abs(t - mu.unsqueeze(-1)), dim=-1) if stats is None else stats['b']
        rng = (2 * am * b) if sym else (torch.max(mu, mu.new_tensor([0.])) + am * b)
    else:
        rng = (t.max(-1)[0] - t.min(-1)[0]) if sym else t.max(-1)[0]
    Delta = torch.where(omega > 0, rng / omega, t.new_tensor([F32_MAX]))
    q = (t / Delta.unsqueeze(-1)).round()
    c_min = c_max = None
    if clip:
        mu_q = mu / Delta if sym else torch.max(mu, mu.new_tensor([0.])) / Delta
        c_max = mu_q + (omega / 2 if sym else omega)
        c_min = (mu_q - omega / 2) if sym else t.new_tensor([0])
        q = torch.min(q, c_max.unsqueeze(-1))
        q = torch.max(q, c_min.unsqueeze(-1))
    entropy = shannon_entropy(q) if want_entropy else None
    codes = q.clone()
    y = q * Delta.unsqueeze(-1)
    if return_parts:
        return y, entropy, dict(codes=codes, omega=omega, Delta=Delta, c_min=c_min, c_max=c_max, std=std,
                                alpha_mult=am if clip else None)
    return y, entropy


def mid_tread_act_per_channel(x, target, half_range=False, force_positive=False, omega_table=None,
                              alpha_table=None, want_entropy=False):
    """iq.py:170-183."""
    N, C, H, W = x.shape
    sym = not (force_positive or half_range)
    y, ent = mid_tread_core(_channel_rows(x), target, True, sym, omega_table, alpha_table, want_entropy)
    return y.view(C, N, H, W).transpose(0, 1).contiguous(), ent


def mid_tread_weights_per_channel(w, target, want_entropy=False):
    """iq.py:147-156."""
    y, ent = mid_tread_core(w.view(w.shape[0], -1), target, False, True, want_entropy=want_entropy)
    return y.view(w.shape), ent


# ----------------------------------------------------------------------------- stats collection (a16)
COLLECT_STATS = ['max', 'min', 'std', 'mean', 'kurtosis', 'b', 'std_pos']


def collect_stats_perchannel(x, batch_avg=False, force_global_min_max=False):
    """Row a16, smpc.py:45-79: the seven per-channel statistics of one batch ([C] each), or
    None for tensors the reference skips (smpc.py:47-48)."""
    if x.dim() < 3 or (x.shape[2] == 1 and x.shape[3] == 1):
        return None
    t = _channel_rows(x)
    per_sample = x.view(x.shape[0], x.shape[1], -1)
    mean_ = t.mean(-1)
    std_ = torch.std(t, dim=-1, unbiased=True)
    out = {}
    out['kurtosis'] = torch.mean(((t - mean_.unsqueeze(-1)) / std_.unsqueeze(-1)) ** 4, dim=-1) - 3
    out['b'] = torch.mean(torch.abs(t - mean_.unsqueeze(-1)), dim=-1)
    out['std'] = std_
    out['std_pos'] = torch.std(torch.nn.functional.relu(t), dim=-1, unbiased=True)
    out['mean'] = mean_
    if force_global_min_max:
        out['max'] = t.max(-1)[0]
        out['min'] = t.min(-1)[0]
    else:
        out['max'] = torch.mean(per_sample.max(dim=-1)[0], dim=0) if batch_avg else t.max(-1)[0]
        out['min'] = torch.mean(per_sample.min(dim=-1)[0], dim=0) if batch_avg else \
            torch.min(per_sample.min(dim=-1)[0], dim=0)[0]
    return out
