#!/usr/bin/env python3
"""Manager-level golden vectors: drive the REFERENCE's QuantizationManagerInference + *WithId layers
(inference_quantization_manager.py) on a small seeded network and record, per flag set,
  * the ordered verbose trace (`Quantize <tag> | Id - <stat_id> | <repr(quantizer)>`),
  * the quantized (and bias/variance corrected) weights after quantize_model,
  * every patched layer's output and the network output,
  * for -sm collect/use: the statistics files the reference wrote (copied under stats_home/).

Runs only in the build container.  Two things cannot run here and are substituted IN THIS DRIVER:
  * the CUDA extension `int_quantization.float2gemmlowp` -> the oracle's restatement of
    kernels/gemmlowp.cu (oracle/quant_oracle.py); fixtures of per-tensor layers (maxpool, avgpool,
    classifier, conv0 in -sm use) therefore pin the plumbing, not that kernel (it has its own pin);
  * `torch.cuda.FloatTensor` (the zero noise tensor of int_quantizer.py:610) -> torch.FloatTensor.
A stub `torchvision` (models.Inception3) satisfies the import at inference_quantization_manager.py:3.
"""
import argparse
import contextlib
import io
import os
import shutil
import sys
import tempfile
import types

import numpy as np

REF = os.environ.get('CNNQ_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
HOME = tempfile.mkdtemp(prefix='cnnq_golden_home_')
os.environ['HOME'] = HOME
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from oracle import quant_oracle as O  # noqa: E402

sys.path.insert(0, REF)
_iq = types.ModuleType('int_quantization')
_iq.float2gemmlowp = lambda t, rng, off, bits, int_exp, etz, noise: O.float2gemmlowp(
    t, float(rng), float(off), bits, bool(int_exp), bool(etz))
sys.modules['int_quantization'] = _iq
_tv = types.ModuleType('torchvision')
_tv.models = types.ModuleType('torchvision.models')
_tv.models.Inception3 = type('Inception3', (nn.Module,), {})
sys.modules['torchvision'] = _tv
sys.modules['torchvision.models'] = _tv.models
torch.cuda.FloatTensor = torch.FloatTensor

from pytorch_quantizer.quantization.inference import inference_quantization_manager as iqm  # noqa: E402
from utils.misc import Singleton  # noqa: E402

QM = iqm.QuantizationManagerInference

# GPU semantics for get_alpha_mult: on CPU `.cpu().numpy()` aliases and the
# reference doubles the caller's omega in place; on its real device it does not.  Hand it a clone.
_iqmod = sys.modules['pytorch_quantizer.quantization.qtypes.int_quantizer']
_orig_alpha_mult = _iqmod.IntQuantizer.get_alpha_mult
_iqmod.IntQuantizer.get_alpha_mult = staticmethod(lambda omega, sym=True: _orig_alpha_mult(omega.clone(), sym=sym))


def make_args(**kw):
    a = dict(arch='toynet', qtype='int4', qweight='int4', q_off=False, stats_mode='no', stats_folder=None,
             kld_threshold=False, per_channel_quant_act=True, stats_batch_avg=False, bias_corr_act=False,
             bias_corr_weight=False, var_corr_weight=False, measure_stats=False)
    a.update(kw)
    return argparse.Namespace(**a)


def make_qparams(args, **kw):
    p = dict(clipping='no', stats_kind='mean', true_zero=False, kld=False, pcq_weights=True,
             pcq_act=args.per_channel_quant_act, bit_alloc_act=False, bit_alloc_weight=False, bit_alloc_rmode='round',
             bit_alloc_prior='gaus', bit_alloc_target_act=None, bit_alloc_target_weight=None,
             bcorr_act=args.bias_corr_act, bcorr_weight=args.bias_corr_weight, vcorr_weight=args.var_corr_weight,
             logger=None, measure_entropy=False, mtd_quant=False)
    p.update(kw)
    return {'int': p, 'qmanager': {'rho_act': None, 'rho_weight': None}}


def build_toynet():
    """Built while the manager is enabled, so nn.Conv2d etc. are the patched classes."""
    class ToyNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 8, 3, padding=1, bias=False)       # conv0: 3 input channels -> 8-bit weights
            self.relu = nn.ReLU()
            self.maxpool = nn.MaxPool2d(2)
            self.conv2 = nn.Conv2d(8, 16, 1, bias=False)                  # before_relu
            self.bn2 = nn.BatchNorm2d(16)                                 # not absorbed: goes to the default quantizer
            self.conv3 = nn.Conv2d(16, 16, 3, padding=1, bias=True)       # before_relu
            self.conv4 = nn.Conv2d(16, 8, 1, bias=False)                  # full range
            self.avgpool = nn.AvgPool2d(4)
            self.fc = nn.Linear(8 * 2 * 2, 1000)

        def forward(self, x):
            x = self.maxpool(self.relu(self.conv1(x)))
            x = self.relu(self.bn2(self.conv2(x)))
            x = self.relu(self.conv3(x))
            x = self.conv4(x)
            x = self.avgpool(x)
            return self.fc(x.view(x.size(0), -1))

    torch.manual_seed(777)
    m = ToyNet()
    m.conv1.before_relu = True
    m.conv2.before_relu = True
    m.bn2.before_relu = True
    m.conv3.before_relu = True
    for n, mod in m.named_modules():
        mod.internal_name = 'ToyNet/' + n
    with torch.no_grad():
        m.bn2.running_mean.normal_(0, 0.1)
        m.bn2.running_var.uniform_(0.5, 1.5)
    return m.eval()


def reset_reference_state():
    for cls in list(Singleton._instances):
        Singleton._instances.pop(cls)
    for c in (iqm.Conv2dWithId, iqm.LinearWithId, iqm.MaxPool2dWithId, iqm.AvgPool2dWithId, iqm.BatchNorm2dWithId,
              iqm.ReLUWithId):
        from itertools import count
        c._id = count(0)


def run(args, qparams, x, record):
    """One experiment exactly in the order of inference_sim.py:375-390 / InferenceModel.__init__."""
    reset_reference_state()
    outs = {}
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        with QM(args, qparams):
            model = build_toynet()
            hooks = []
            for n, mod in model.named_modules():
                if isinstance(mod, (nn.Conv2d, nn.Linear, nn.MaxPool2d, nn.AvgPool2d, nn.BatchNorm2d)):
                    hooks.append(mod.register_forward_hook(
                        lambda m, i, o, n=n: outs.__setitem__(n, o.detach().clone())))
            w_before = {n: p.detach().clone() for n, p in model.named_parameters() if n.endswith('weight')}
            QM().quantize_model(model)
            QM().verbose = True
            with torch.no_grad():
                y = model(x)
    trace = [ln for ln in buf.getvalue().splitlines() if ln.startswith('Quantize ')]
    if record is not None:
        record['trace'] = np.array(trace)
        record['y'] = y.detach().numpy()
        for n, p in model.named_parameters():
            if n.endswith('weight') and p.dim() > 1:
                record['w_before_sum/' + n] = np.float64(w_before[n].double().sum().item())
                record['w_after/' + n] = p.detach().numpy()
        for n, o in outs.items():
            record['out/' + n] = o.numpy()
    return y


CONFIGS = [
    ('cfg2', dict(), dict()),
    ('cfg2_bcw_vcw', dict(bias_corr_weight=True, var_corr_weight=True), dict()),
    ('cfg3', dict(bias_corr_weight=True), dict(clipping='laplace', bit_alloc_act=True, bit_alloc_weight=True)),
    ('cfg5_vgg', dict(arch='vgg16'), dict(clipping='laplace', mtd_quant=True, measure_entropy=False,
                                          bit_alloc_target_act=4, bit_alloc_target_weight=4)),
    ('int8_per_tensor', dict(qtype='int8', qweight='int8', per_channel_quant_act=False), dict(pcq_weights=False)),
]


def main():
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(31337)
    x = torch.randn(4, 3, 16, 16, generator=g)
    d = {'x': x.numpy(), 'configs': np.array([c[0] for c in CONFIGS] + ['use_cfg3_bca'])}
    for name, akw, qkw in CONFIGS:
        args = make_args(**akw)
        rec = {}
        run(args, make_qparams(args, **qkw), x, rec)
        for k, v in rec.items():
            d[name + '/' + k] = v
        print(name, len(rec['trace']), 'quantize calls')
    # -sm collect (per-channel and per-tensor files), then -sm use with ACIQ + bit allocation + -bca
    xs = [torch.randn(4, 3, 16, 16, generator=g) for _ in range(3)]
    for pc in (True, False):
        args = make_args(stats_mode='collect', per_channel_quant_act=pc, stats_folder='toynet_stats')
        reset_reference_state()
        with contextlib.redirect_stdout(io.StringIO()):
            with QM(args, make_qparams(args)):
                model = build_toynet()
                QM().quantize_model(model)
                with torch.no_grad():
                    for xb in xs:
                        model(xb)
    args = make_args(stats_mode='use', stats_folder='toynet_stats', bias_corr_act=True, bias_corr_weight=True)
    rec = {}
    run(args, make_qparams(args, clipping='laplace', bit_alloc_act=True, bit_alloc_weight=True), x, rec)
    for k, v in rec.items():
        d['use_cfg3_bca/' + k] = v
    print('use_cfg3_bca', len(rec['trace']), 'quantize calls')
    for i, xb in enumerate(xs):
        d['collect_x%d' % i] = xb.numpy()
    dst = os.path.join(OUT, 'stats_home')
    if os.path.exists(dst):
        shutil.rmtree(dst)
    shutil.copytree(os.path.join(HOME, 'mxt-sim'), os.path.join(dst, 'mxt-sim'))
    path = os.path.join(OUT, 'manager.npz')
    np.savez_compressed(path, **d)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024.))
    for r, _, fs in os.walk(dst):
        for f in fs:
            print('  stats file', os.path.relpath(os.path.join(r, f), OUT))


if __name__ == '__main__':
    main()
