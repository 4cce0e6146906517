"""Build libcnnq_hip.so (the C-ABI library of include/cnnq_hip.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so then travels
to the GPU box with the repository snapshot."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = [os.path.join(PKG, 'csrc', 'cnnq_kernels.hip')]
HDR = [os.path.join(ROOT, 'include', 'cnnq_hip.h')] + sorted(
    os.path.join(PKG, 'csrc', f) for f in os.listdir(os.path.join(PKG, 'csrc')) if f.endswith('.hip.h'))
LIB = os.path.join(PKG, 'libcnnq_hip.so')
# --offload-compress (round 6): the gfx950 code object inside the library compressed (zstd; the HIP runtime of ROCm 7 unpacks it
# at load): 11 MB -> 2 MB of snapshot on every push to a GPU box, the same code
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '--offload-compress']


def hipcc_path():
    for c in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found: cannot build %s' % LIB)


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SRC + HDR)


def build(force=False, verbose=False):
    """Compile the library if it is missing or older than its sources; return its path."""
    if not force and not stale():
        return LIB
    cmd = [hipcc_path()] + FLAGS + ['-I', os.path.join(ROOT, 'include')] + SRC + ['-o', LIB + '.tmp']
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
