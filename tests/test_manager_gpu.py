"""The layer-substitution plumbing (cnn_quantization_amd.inference.inference_quantization_manager)
against traces and tensors recorded from the REFERENCE's manager on the same seeded toy network
(tests/golden/make_golden_manager.py).

What must be identical: the ordered trace of quantize calls - tag, stat_id and repr(quantizer) (i.e.
which quantizer with which flags handles which layer, incl. the reference's positional quirks) - and
the quantized weights (same inputs, bit-exact Q/DQ).  Layer OUTPUTS are compared within a few
quantization steps: the convolutions themselves run in MIOpen here and in CPU aten in the fixture, so
the quantizers do not see bit-identical inputs."""
import argparse
import contextlib
import io
import os
import pickle
import shutil
from itertools import count

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


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
    """Same topology, marks and seed as the fixture generator; built under the patched classes."""
    class ToyNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 8, 3, padding=1, bias=False)
            self.relu = nn.ReLU()
            self.maxpool = nn.MaxPool2d(2)
            self.conv2 = nn.Conv2d(8, 16, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(16)
            self.conv3 = nn.Conv2d(16, 16, 3, padding=1, bias=True)
            self.conv4 = nn.Conv2d(16, 8, 1, bias=False)
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
    for mod in (m.conv1, m.conv2, m.bn2, m.conv3):
        mod.before_relu = True
    for n, mod in m.named_modules():
        mod.internal_name = 'ToyNet/' + n
    with torch.no_grad():
        m.bn2.running_mean.normal_(0, 0.1)
        m.bn2.running_var.uniform_(0.5, 1.5)
    return m.eval()


def fresh_manager_state():
    from cnn_quantization_amd.inference import inference_quantization_manager as M
    from cnn_quantization_amd.utils.misc import Singleton
    Singleton.reset()
    for c in (M.Conv2dWithId, M.LinearWithId, M.MaxPool2dWithId, M.AvgPool2dWithId, M.BatchNorm2dWithId, M.ReLUWithId):
        c._id = count(0)
    return M


def run(args, qparams, xs):
    M = fresh_manager_state()
    outs, buf = {}, io.StringIO()
    with contextlib.redirect_stdout(buf):
        with M.QuantizationManagerInference(args, qparams):
            assert nn.Conv2d is M.Conv2dWithId                       # classes are swapped while enabled
            model = build_toynet()
            wsum = {n: float(p.detach().double().sum()) for n, p in model.named_parameters() if p.dim() > 1}
            model = model.cuda()
            for n, mod in model.named_modules():
                if isinstance(mod, (nn.Conv2d, nn.Linear, nn.MaxPool2d, nn.AvgPool2d, nn.BatchNorm2d)):
                    mod.register_forward_hook(lambda m, i, o, n=n: outs.__setitem__(n, o.detach().cpu().clone()))
            M.QMI().quantize_model(model)
            M.QMI().verbose = True
            with torch.no_grad():
                for x in xs:
                    y = model(x.cuda())
    assert nn.Conv2d is not M.Conv2dWithId                           # and restored afterwards
    torch.cuda.synchronize()
    trace = [ln.rsplit(' | ', 1)[0] for ln in buf.getvalue().splitlines() if ln.startswith('Quantize ')]
    weights = {n: p.detach().cpu() for n, p in model.named_parameters() if p.dim() > 1}
    return trace, weights, outs, y.cpu(), wsum


CONFIGS = {
    'cfg2': (dict(), dict()),
    'cfg2_bcw_vcw': (dict(bias_corr_weight=True, var_corr_weight=True), dict()),
    'cfg3': (dict(bias_corr_weight=True), dict(clipping='laplace', bit_alloc_act=True, bit_alloc_weight=True)),
    'cfg5_vgg': (dict(arch='vgg16'), dict(clipping='laplace', mtd_quant=True, measure_entropy=False,
                                          bit_alloc_target_act=4, bit_alloc_target_weight=4)),
    'int8_per_tensor': (dict(qtype='int8', qweight='int8', per_channel_quant_act=False), dict(pcq_weights=False)),
}


def close_enough(a, b, step_frac=0.15):
    """Same tensor up to quantization-boundary flips: most elements agree to 1e-4 of the range, none
    is further than `step_frac` of the range away."""
    a, b = a.float().numpy(), np.asarray(b, dtype=np.float32)
    rng = float(b.max() - b.min()) + 1e-12
    d = np.abs(a - b)
    return d.max() <= step_frac * rng and (d > 1e-4 * rng).mean() < 0.08


@pytest.mark.parametrize('name', list(CONFIGS))
def test_trace_weights_outputs(golden, name):
    g = golden('manager')
    akw, qkw = CONFIGS[name]
    args = make_args(**akw)
    trace, weights, outs, y, wsum = run(args, make_qparams(args, **qkw), [g.t('x')])
    ref_trace = [str(t).rsplit(' | ', 1)[0] for t in g.np(name + '/trace')]
    assert trace == ref_trace
    for n, w in weights.items():
        assert abs(wsum[n] - float(g.np('%s/w_before_sum/%s' % (name, n)))) < 1e-9      # same initial weights
        ref = g.np('%s/w_after/%s' % (name, n))
        if name in ('cfg2', 'int8_per_tensor'):
            assert np.array_equal(w.numpy().view(np.uint32), ref.view(np.uint32)), n      # bit-exact
        else:
            np.testing.assert_allclose(w.numpy(), ref, rtol=2e-5, atol=float(np.abs(ref).max()) * 0.13, err_msg=n)
            assert (np.abs(w.numpy() - ref) > 1e-6).mean() < 0.02, n
    for n, o in outs.items():
        assert close_enough(o, g.np('%s/out/%s' % (name, n))), (name, n)
    assert close_enough(y, g.np(name + '/y')), name


def test_use_mode_with_reference_written_stats(golden, tmp_path, monkeypatch):
    """-sm use -c laplace -baa -baw -bcw -bca driven by the statistics files the REFERENCE wrote
    (pickle + CSV): proves the file formats are interchangeable and exercises the conv0 -> 'ignored'
    8-bit rule and the activation bias correction."""
    g = golden('manager')
    shutil.copytree(os.path.join(GOLDEN, 'stats_home', 'mxt-sim'), str(tmp_path / 'mxt-sim'))
    monkeypatch.setenv('HOME', str(tmp_path))
    args = make_args(stats_mode='use', stats_folder='toynet_stats', bias_corr_act=True, bias_corr_weight=True)
    qp = make_qparams(args, clipping='laplace', bit_alloc_act=True, bit_alloc_weight=True)
    trace, weights, outs, y, _ = run(args, qp, [g.t('x')])
    ref_trace = [str(t).rsplit(' | ', 1)[0] for t in g.np('use_cfg3_bca/trace')]
    assert trace == ref_trace
    assert any('ignored' not in t and 'conv0_activation' in t and 'bits: 8' in t for t in trace)   # conv0 kept at 8 bit
    for n, o in outs.items():
        assert close_enough(o, g.np('use_cfg3_bca/out/%s' % n)), n
    assert close_enough(y, g.np('use_cfg3_bca/y'))


def test_collect_mode_writes_reference_schema(golden, tmp_path, monkeypatch):
    """-sm collect over the same three batches: same layer ids, same DataFrame columns / dtypes as the
    reference's pickle, values equal up to the conv implementation."""
    g = golden('manager')
    monkeypatch.setenv('HOME', str(tmp_path))
    args = make_args(stats_mode='collect', stats_folder='mine')
    run(args, make_qparams(args), [g.t('collect_x%d' % i) for i in range(3)])
    mine = pickle.load(open(str(tmp_path / 'mxt-sim/statistics/per_channel/mine/mine_statistics_perchannel_summary.pkl'), 'rb'))
    ref = pickle.load(open(os.path.join(GOLDEN, 'stats_home/mxt-sim/statistics/per_channel/toynet_stats/'
                                        'toynet_stats_statistics_perchannel_summary.pkl'), 'rb'))
    assert sorted(mine) == sorted(ref)
    for layer in ref:
        assert list(mine[layer].columns) == list(ref[layer].columns)
        assert list(mine[layer].dtypes) == list(ref[layer].dtypes)
        assert mine[layer].shape == ref[layer].shape
        a, b = mine[layer].values.astype(np.float64), ref[layer].values.astype(np.float64)
        kurt = np.array(['kurtosis' in c for c in ref[layer].columns])
        np.testing.assert_allclose(a[:, ~kurt], b[:, ~kurt], rtol=2e-4, atol=2e-5, err_msg=layer)
        np.testing.assert_allclose(a[:, kurt], b[:, kurt], rtol=5e-3, atol=5e-3, err_msg=layer)
