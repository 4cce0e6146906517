"""INTEGRATION.md section 2, executed: the REFERENCE's own manager module, imported unmodified with
the sys.modules shims that section shows, builds its tag -> quantizer table out of THIS package's
IntQuantizer.  No tensor is quantized (that needs the GPU); what is proven is that the attribute
surface the reference pokes at after construction exists and behaves (inference_quantization_manager.py
:407-476, 549-562).  Runs only where /root/reference exists (the build container)."""
import argparse
import os
import subprocess
import sys

import pytest

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, types, argparse, os, tempfile
os.environ['HOME'] = tempfile.mkdtemp()
sys.path.insert(0, %(root)r)
import torch.nn as nn
import cnn_quantization_amd
from cnn_quantization_amd import int_quantization
import cnn_quantization_amd.qtypes.int_quantizer, cnn_quantization_amd.qtypes.dummy_quantizer
import cnn_quantization_amd.inference.statistic_manager_perchannel, cnn_quantization_amd.inference.statistic_manager
# ---- the shims of INTEGRATION.md section 2
sys.modules['int_quantization'] = int_quantization
sys.modules['pytorch_quantizer.quantization.qtypes.int_quantizer'] = sys.modules['cnn_quantization_amd.qtypes.int_quantizer']
sys.modules['pytorch_quantizer.quantization.qtypes.dummy_quantizer'] = sys.modules['cnn_quantization_amd.qtypes.dummy_quantizer']
sys.modules['pytorch_quantizer.quantization.inference.statistic_manager_perchannel'] = \
    sys.modules['cnn_quantization_amd.inference.statistic_manager_perchannel']
sys.modules['pytorch_quantizer.quantization.inference.statistic_manager'] = \
    sys.modules['cnn_quantization_amd.inference.statistic_manager']
tv = types.ModuleType('torchvision'); tv.models = types.ModuleType('torchvision.models')
tv.models.Inception3 = type('Inception3', (nn.Module,), {})
sys.modules['torchvision'] = tv; sys.modules['torchvision.models'] = tv.models
sys.path.insert(0, %(ref)r)
from pytorch_quantizer.quantization.inference import inference_quantization_manager as iqm   # the REFERENCE's manager
from cnn_quantization_amd.qtypes.int_quantizer import IntQuantizer
args = argparse.Namespace(arch='resnet50', qtype='int4', qweight='int4', q_off=False, stats_mode='no', stats_folder=None,
                          kld_threshold=False, per_channel_quant_act=True, stats_batch_avg=False, bias_corr_act=False,
                          bias_corr_weight=True, var_corr_weight=False, measure_stats=False)
qp = {'int': dict(clipping='laplace', stats_kind='mean', true_zero=False, kld=False, pcq_weights=True, pcq_act=True,
                  bit_alloc_act=True, bit_alloc_weight=True, bit_alloc_rmode='round', bit_alloc_prior='gaus',
                  bit_alloc_target_act=None, bit_alloc_target_weight=None, bcorr_act=False, bcorr_weight=True,
                  vcorr_weight=False, logger=None, measure_entropy=False, mtd_quant=False),
      'qmanager': {'rho_act': None, 'rho_weight': None}}
with iqm.QuantizationManagerInference(args, qp) as qm:
    assert nn.Conv2d is iqm.Conv2dWithId
    table = qm.op_manager.quantizers
    for tag in ('activation', 'activation_linear', 'activation_pooling', 'activation_classifier', 'weight',
                'weight_classifier', 'ignored'):
        assert type(table[tag]) is IntQuantizer, (tag, type(table[tag]))
        print(tag, '|', repr(table[tag]))
    assert type(qm.op_manager.quantizer_default) is IntQuantizer
    assert table['activation'].clipping == 'laplace' and table['activation'].pcq_a and not table['activation'].pcq_w
    assert table['activation_pooling'].num_bits == 8 and table['activation_pooling'].clipping == 'no'
    assert qm.op_manager.ignore_ids == ['conv0_activation']
print('SHIM-OK')
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason='the reference tree is only present in the build container')
def test_reference_manager_builds_its_table_from_our_quantizer(golden):
    out = subprocess.run([sys.executable, '-c', SCRIPT % dict(root=ROOT, ref=REF)], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and 'SHIM-OK' in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    # the reprs equal those the reference's own quantizers print for the same flags (golden trace, cfg3)
    g = golden('manager')
    ref_reprs = {}
    for line in g.np('cfg3/trace'):
        tag, _, rep = [t.strip() for t in str(line).split('|')[:3]]
        ref_reprs.setdefault(tag.replace('Quantize', '').strip(), rep)
    mine = dict(l.split(' | ', 1) for l in out.stdout.splitlines() if ' | IntQuantizer' in l)
    for tag in ('activation', 'activation_pooling', 'activation_classifier', 'weight', 'weight_classifier'):
        assert mine[tag] == ref_reprs[tag], (tag, mine[tag], ref_reprs[tag])
