"""The C ABI from a plain C client on the GPU: tests/c/cabi_demo.c (gcc -std=c99, links libcnnq_hip.so and the
HIP runtime only) quantizes a tensor with cnnq_pc_minmax_qdq and checks it bit for bit against scalar C, then runs the
one-call single-launch route (with a group workspace) and the two halves of the multi-GPU form and compares the three."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_client_bit_exact(tmp_path):
    gcc = shutil.which('gcc')
    if gcc is None or not os.path.isdir('/opt/rocm/include/hip'):
        pytest.skip('no C toolchain / HIP headers on this box')
    from cnn_quantization_amd import _build
    lib_dir = os.path.dirname(_build.build())
    exe = str(tmp_path / 'cabi_demo')
    subprocess.run([gcc, '-std=c99', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I', os.path.join(ROOT, 'include'),
                    os.path.join(ROOT, 'tests', 'c', 'cabi_demo.c'), '-L', lib_dir, '-lcnnq_hip', '-L/opt/rocm/lib',
                    '-lamdhip64', '-lm', '-o', exe], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=lib_dir + ':/opt/rocm/lib:' + os.environ.get('LD_LIBRARY_PATH', ''))
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'bit-exact' in r.stdout and 'agree bit for bit' in r.stdout, r.stdout
