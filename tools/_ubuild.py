"""Build-on-demand for the microbenchmark libraries next to this file: so('ubench_rw') compiles tools/ubench_rw.hip into
tools/ubench_rw.so with hipcc for gfx950 when the .so is missing or older than its source, and returns the path.  The
.so files are not tracked and do not travel to the GPU box (.gpurunignore): a tool builds what it needs where it runs
(hipcc is in the image; ~10 s per file)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def so(name, extra=()):
    src, lib = os.path.join(HERE, name + '.hip'), os.path.join(HERE, name + '.so')
    if not os.path.exists(src):
        raise FileNotFoundError(src)
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-I',
               os.path.join(os.path.dirname(HERE), 'include'), '-I', os.path.join(os.path.dirname(HERE), 'cnn_quantization_amd', 'csrc')]
        subprocess.run(cmd + list(extra) + [src, '-o', lib + '.tmp'], check=True)
        os.replace(lib + '.tmp', lib)
    return lib
