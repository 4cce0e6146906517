#!/usr/bin/env python3
"""Registers / scratch / LDS of the library's kernels, from the gfx950 code object inside libcnnq_hip.so (runs on the CPU:
llvm-objcopy of .hip_fatbin + clang-offload-bundler + llvm-readelf).  kregs.py [substring ...]: only kernels whose demangled name contains one of them."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'cnn_quantization_amd', 'libcnnq_hip.so')


def main():
    lib = os.environ.get('CNNQ_HIP_LIB') or LIB
    want = sys.argv[1:]
    with tempfile.TemporaryDirectory() as d:
        co, fat = os.path.join(d, 'dev.co'), os.path.join(d, 'fat.bin')
        subprocess.run([LLVM + '/llvm-objcopy', '-O', 'binary', '--only-section=.hip_fatbin', lib, fat], check=True)
        subprocess.run([LLVM + '/clang-offload-bundler', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        '--input=' + fat, '--output=' + co, '--unbundle'], check=True, capture_output=True)
        notes = subprocess.run([LLVM + '/llvm-readelf', '--notes', co], check=True, capture_output=True, text=True).stdout
    rows = []
    for b in notes.split('  - .agpr_count:')[1:]:
        get = lambda k: re.search(r'\.%s:\s+(\S+)' % k, b).group(1)
        rows.append((get('name'), int(get('vgpr_count')), int(get('sgpr_count')), int(get('private_segment_fixed_size')),
                     int(get('group_segment_fixed_size'))))
    names = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.split('\n')
    for (_, vg, sg, sp, lds), dn in zip(rows, names):
        dn = dn.replace('(anonymous namespace)::', '').split('(')[0]
        if want and not any(w in dn for w in want):
            continue
        waves = 512 // max(vg, 1) if vg else 8
        print('%-64s vgpr %3d (<= %d waves/SIMD)  sgpr %3d  scratch %4d  lds %6d' % (dn[:64], vg, min(waves, 8), sg, sp, lds))


if __name__ == '__main__':
    main()
