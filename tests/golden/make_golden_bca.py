#!/usr/bin/env python3
"""Golden vectors for the activation bias correction, produced by RUNNING THE
REFERENCE's Conv2dWithId.forward (inference_quantization_manager.py:160-200) in `-sm use` + `-bca`
state.  The correction does not depend on how the quantized tensor was produced, so the driver swaps
the manager's quantize_instant for a fixed 0.25-grid rounding and records, per layer call,
(conv output, quantized output before the correction, corrected output, relu-first flag).
Runs only in the build container; reuses the reference setup of make_golden_manager.py."""
import contextlib
import io
import os

import numpy as np
import torch

import make_golden_manager as M

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.set_num_threads(1)
    args = M.make_args(bias_corr_act=True)
    M.reset_reference_state()
    rec = {}
    with contextlib.redirect_stdout(io.StringIO()):
        with M.QM(args, M.make_qparams(args)):
            qm = M.QM()
            qm.stats_mode = M.iqm.StatsMode.use_stats
            qm.bcorr_act = True
            seen = {}

            def fake_quantize(tensor, id, tag="", stat_id=None, half_range=False, override_att=None, verbose=False):
                q = torch.round(tensor * 4) / 4
                if half_range:
                    q = torch.clamp(q, min=0)
                seen['out'], seen['out_q'] = tensor.clone(), q.clone()
                return q
            qm.quantize_instant = fake_quantize
            g = torch.Generator().manual_seed(4242)
            cases = [('relu_first', (4, 8, 7, 7), True), ('full_range', (3, 20, 5, 9), False),
                     ('one_channel_dead', (2, 6, 4, 4), True)]
            for name, shape, before_relu in cases:
                conv = torch.nn.Conv2d(shape[1], shape[1], 1, bias=False)      # the patched Conv2dWithId
                with torch.no_grad():
                    conv.weight.copy_(torch.eye(shape[1]).view(shape[1], shape[1], 1, 1))
                if before_relu:
                    conv.before_relu = True
                x = torch.randn(shape, generator=g) * 1.3 + 0.2
                if name == 'one_channel_dead':
                    x[:, 2] = -x[:, 2].abs()                                   # no positive element: count = 0
                with torch.no_grad():
                    y = conv(x)
                rec[name + '/out'] = seen['out'].numpy()
                rec[name + '/out_q'] = seen['out_q'].numpy()
                rec[name + '/corrected'] = y.numpy()
                rec[name + '/relu_first'] = np.array(before_relu)
    path = os.path.join(OUT, 'bca.npz')
    np.savez_compressed(path, **rec)
    print('wrote', path, os.path.getsize(path), 'bytes', sorted(rec))


if __name__ == '__main__':
    main()
