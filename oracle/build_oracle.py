"""Compile oracle/qdq_core.c (the plain-C restatement) with gcc.  Test infrastructure only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'qdq_core.c')
LIB = os.path.join(HERE, 'liboracle_qdq.so')


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    subprocess.run(['gcc', '-O2', '-ffp-contract=off', '-fno-fast-math', '-shared', '-fPIC', SRC, '-o', LIB + '.tmp',
                    '-lm'], check=True)
    os.replace(LIB + '.tmp', LIB)
    return LIB


def load():
    import ctypes
    lib = ctypes.CDLL(build())
    P, L, F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float
    lib.oracle_pc_qdq.argtypes = [P, P, P, L, L, L, P, P, P]
    lib.oracle_pc_qdq.restype = None
    lib.oracle_pt_qdq.argtypes = [P, P, L, F, F, F, ctypes.c_int]
    lib.oracle_pt_qdq.restype = None
    return lib


if __name__ == '__main__':
    print(build(force=True))
