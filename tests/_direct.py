"""The DIRECT form of the single-launch parity checks: the oracle's own functions (oracle/quant_oracle.py, pinned to the reference's
vectors by tests/test_oracle_golden.py) applied to the statistics table the DEVICE reduced, and what must then come out of the
launch bit for bit - parameters, codes, floats.  Shared by the one-GPU tests (test_aciq_single_gpu.py,
test_midtread_single_gpu.py) and the sharded ones (test_sharded_single_gpu.py), where the table is the global batch's and x the
whole batch."""
import numpy as np
import torch

from oracle import quant_oracle as O


def aciq_on_table(x_cpu, st, bits_dev, num_bits, half, ba):
    """Config 3: alpha, delta / offset, parameters and the element-wise result the oracle derives from the table `st`
    (iq.py:227-253, 284-300, 393-407 -> 557-603)."""
    from cnn_quantization_amd import _lib as L
    C = x_cpu.shape[1]
    st = st.cpu()
    bits = O.bits_alloc_fixed_target(st[L.STAT_STD], num_bits, True) if ba else None
    if ba:
        assert torch.equal(bits_dev.cpu(), bits)
    alpha = O.alpha_laplace(st[L.STAT_B], num_bits, half, bits)
    delta, offset = O.alpha_to_delta_offset(alpha, st[L.STAT_MAX], st[L.STAT_MIN], st[L.STAT_MEAN], half)
    delta, offset = torch.as_tensor(delta, dtype=torch.float32), torch.as_tensor(offset, dtype=torch.float32) * torch.ones(C)
    max_ = offset + delta                                    # iq.py:351 then :443 (two fp32 roundings)
    t = O._channel_rows(x_cpu)
    y, codes, scale, zp, qmax = O.qdq_core(t, max_ - offset, offset, num_bits=num_bits, bit_alloc=bits, return_parts=True)
    N, _, H, W = x_cpu.shape
    y = y.view(C, N, H, W).transpose(0, 1).contiguous()
    codes = codes.view(C, N, H, W).transpose(0, 1).contiguous()
    return dict(alpha=torch.as_tensor(alpha, dtype=torch.float32), delta=max_ - offset, offset=offset, scale=scale, zp=zp,
                qmax=qmax * torch.ones(C), y=y, codes=codes)


def midtread_tables():
    """The (omega, alpha) interpolation tables as the oracle takes them (numpy fp64) - the product's copy of iq.py:41-51, which
    tests/test_oracle_golden.py::test_tables compares with the reference's."""
    from cnn_quantization_amd.qtypes._midtread_tables import ALPHA_TABLE, OMEGA_TABLE
    return np.asarray(OMEGA_TABLE, dtype=np.float64), np.asarray(ALPHA_TABLE, dtype=np.float64)


def midtread_on_table(x_cpu, st, target, sym, want_entropy=True):
    """Config 5 with clipping (iq.py:185-225): omega, the clipping multiplier, Delta, the clamp bounds, the codes, the floats
    and the entropy of the codes from the oracle's mid_tread_core on the rows of x with std / mean / b taken from `st`."""
    from cnn_quantization_amd import _lib as L
    st = st.cpu()
    N, C, H, W = x_cpu.shape
    ot, at = midtread_tables()
    t = O._channel_rows(x_cpu)
    y, ent, p = O.mid_tread_core(t, target, True, sym, ot, at, want_entropy=want_entropy, return_parts=True,
                                 stats=dict(std=st[L.STAT_STD], mean=st[L.STAT_MEAN], b=st[L.STAT_B]))
    y = y.view(C, N, H, W).transpose(0, 1).contiguous()
    codes = p['codes'].view(C, N, H, W).transpose(0, 1).contiguous()
    c_min = p['c_min'] * torch.ones(C)
    return dict(y=y, codes=codes, entropy=None if ent is None else float(ent), omega=p['omega'], alpha_mult=p['alpha_mult'],
                delta=p['Delta'], c_min=c_min, c_max=p['c_max'])
