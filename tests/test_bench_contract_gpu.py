"""The driver's bench.py contract, checked on a small batch: exactly one JSON line on stdout with the agreed keys,
the roofline and cpu_baseline objects, and internally consistent numbers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
                        '--batch', '8'] + list(extra), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_json_line_contract():
    d = run_bench()
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['unit'] == 'elements/s' and d['dtype'] == 'f32' and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config']
    elems = 53 * 0 + sum(c * hw * hw * n for (c, hw, n) in ((64, 112, 1), (256, 56, 4), (128, 56, 1), (512, 28, 5),
                                                            (64, 56, 6), (256, 28, 1), (1024, 14, 7), (128, 28, 7),
                                                            (512, 14, 1), (2048, 7, 4), (256, 14, 11), (512, 7, 5))) * 8
    assert abs(d['value'] * d['ms_per_step'] * 1e-3 - elems) / elems < 1e-6       # value = elements / step time
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0 < r['frac'] < 1
    assert abs(r['achieved'] - r['bytes_per_launch'] / (r['avg_launch_ms'] * 1e-3) / 1e9) / r['achieved'] < 1e-6
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['unit'] == 'elements/s' and c['cores'] >= 1 and c['value'] > 0 and c['sample']
