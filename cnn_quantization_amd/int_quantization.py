"""Drop-in for the reference's native extension module `int_quantization`
(kernels/int_quantization.cpp:10-12, built from kernels/gemmlowp.cu): same entry point, same
seven arguments, same early return - backed by cnnq_pt_setup + cnnq_pt_qdq (HIP, gfx950).

`import int_quantization` resolves to this module once it is registered under that name:
`sys.modules['int_quantization'] = cnn_quantization_amd.int_quantization` (INTEGRATION.md section 2,
exercised by tests/test_reference_shim_cpu.py)."""
import torch

from . import ops


def float2gemmlowp(input, range, offset, num_bits, int_exp, enforce_true_zero, noise=None):
    """float2gemmlowp(Tensor in, float range, float offset, int num_bits, bool int_exp,
    bool enforce_true_zero, Tensor noise) -> Tensor   (kernels/gemmlowp.cu:30-45).

    `range` / `offset` may be Python numbers or 0-dim tensors (pybind converted those through
    __float__, an implicit device sync - the same happens here).  Returns a NEW tensor, or the
    input tensor itself when range <= 0 (gemmlowp.cu:31-32)."""
    rng = float(range)
    off = float(offset)
    if rng <= 0:
        return input
    if not isinstance(input, torch.Tensor) or not input.is_cuda:
        raise RuntimeError('int_quantization.float2gemmlowp needs a CUDA/HIP float32 tensor')
    x = input.contiguous()
    ptp = ops.pt_setup(x.device, int(num_bits), range_offset=(rng, off), int_exp=bool(int_exp),
                       enforce_true_zero=bool(enforce_true_zero))
    return ops.pt_qdq(x, ptp, noise=noise)
