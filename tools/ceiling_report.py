#!/usr/bin/env python3
"""Round 4: the structural ceiling of the read-once kernels, measured on ONE box next to the kernels themselves.

  1. streaming rates of the box: read only / write only / copy with 4 KB one-shot tiles in address order;
  2. the same three with the kernels' 128 KB register tiles, in address order and in the sample-strided order;
  3. the bare register-tile copy whose resident set is what a per-channel exchange forces: blocks of 196 KB per sample x
     all 512 samples (tools/ubench_order: 49 column blocks) - the replay of k_mmq_flat's address stream without
     arithmetic and without a meeting - against whole-tensor address order;
  4. k_mmq_flat / k_mmq_group per layer (tools/bench_group.py) with the product library and with the development
     builds that compile out the stores / the meeting / the loads (tools/alt/libcnnq_abl*.so, -DFLAT_ABL=n).

Writes markdown to stdout; `python tools/ceiling_report.py > gpurun_out/.../ceiling.md`."""
import ctypes, os, re, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(here)
sys.path.insert(0, root)
import torch  # noqa: E402
import bench  # noqa: E402


def load(name, fn, argtypes):
    lib = ctypes.CDLL(__import__('_ubuild').so(name[:-3]) if name.endswith('.so') else os.path.join(here, name))
    f = getattr(lib, fn)
    f.restype = ctypes.c_float
    f.argtypes = argtypes
    return f


def main():
    P_, I_ = ctypes.c_void_p, ctypes.c_int
    urw = load('ubench_rw.so', 'urw', [I_] * 5 + [P_] * 3 + [I_] * 3)
    uord = load('ubench_order.so', 'uord', [P_] * 2 + [I_] * 8)
    print('# Structural ceiling of the read-once kernels - box `%s`\n' % bench.box_id())
    out = torch.zeros(16, device='cuda')
    N, C, hw = 512, 256, 56
    P4 = C * hw * hw // 4
    x = torch.randn(N * P4 * 4, device='cuda'); y = torch.empty_like(x)
    nb = x.numel() * 4

    def rw(J, R, order, mode):
        ms = urw(J, R, order, mode, 1, x.data_ptr(), y.data_ptr(), out.data_ptr(), N, P4, 5)
        return nb * (2 if mode == 0 else 1) / ms / 1e6      # GB/s
    print('Tensor [512,256,56,56] (1644 MB), non-temporal accesses, TB/s of the bytes each mode moves.\n')
    print('| tiles | order | read only | write only | copy |\n|---|---|---|---|---|')
    for (J, R, nm) in ((1, 1, '4 KB one-shot'), (1, 8, '32 KB (8 rows x 4 KB)'), (1, 32, '128 KB (32 rows x 4 KB)')):
        for order, onm in ((0, 'address order'), (1, 'sample-strided')):
            print('| %s | %s | %.2f | %.2f | %.2f |' % (nm, onm, rw(J, R, order, 1) / 1e3, rw(J, R, order, 2) / 1e3, rw(J, R, order, 0) / 1e3))
    print('\nBare 128 KB register-tile copy (load all, then store all; no arithmetic, no meeting) by the width of the co-resident block:\n')
    print('| block (column blocks of 4 KB) | bytes per sample | copy, column-fastest | copy, member-fastest | write only, column-fastest |\n|---|---|---|---|---|')
    for G in (16, 49, 98, 196, 784):
        r = []
        for (order, mode, mult) in ((0, 0, 2), (1, 0, 2), (0, 2, 1)):
            ms = uord(x.data_ptr(), y.data_ptr(), N, P4, G, order, 0, mode, 1, 5)
            r.append(nb * mult / ms / 1e9)                    # TB/s
        print('| %d%s | %d KB | %.2f | %.2f | %.2f |' % (G, ' (what the registers of the chip hold: 784 workgroups)' if G == 49 else ' (whole tensor)' if G == 784 else '', G * 4, r[0], r[1], r[2]))
    del x, y
    torch.cuda.empty_cache()
    print('\nThe kernels on the same box (`tools/bench_group.py`, rotating buffers, us per launch and TB/s of the 8 B/elem):\n')
    libs = [('product library', os.path.join(root, 'cnn_quantization_amd', 'libcnnq_hip.so'))]
    for a, nm in ((10, 'meeting and Q/DQ arithmetic compiled out: the bare copy of the kernel\'s own address stream'), (2, 'meeting compiled out'), (1, 'stores of y compiled out'), (3, 'stores and meeting compiled out (read only)'),
                  (4, 'loads compiled out (write only)'), (6, 'loads and meeting compiled out')):
        p = os.path.join(here, 'alt', 'libcnnq_abl%d.so' % a)
        if os.path.exists(p):
            libs.append((nm, p))
    shapes = '64x112,256x56,512x28,1024x14,256x14'
    rows = {}
    for nm, path in libs:
        env = dict(os.environ, CNNQ_HIP_LIB=path)
        r = subprocess.run([sys.executable, os.path.join(here, 'bench_group.py'), '--rounds', '1', '--reps', '8', '--shapes', shapes],
                           capture_output=True, text=True, env=env, timeout=600)
        for line in r.stdout.splitlines():
            m = re.match(r'C=\s*(\d+) HW=\s*(\d+).*group\s+([0-9.]+) us\s+(\d+) GB/s\(8B\)', line)
            if m:
                rows.setdefault((int(m.group(1)), int(m.group(2))), {})[nm] = (float(m.group(3)), int(m.group(4)))
    names = [nm for nm, _ in libs]
    print('| layer (b512) | ' + ' | '.join(names) + ' |\n|---|' + '---|' * len(names))
    for (Cc, HW), d in rows.items():
        print('| C=%d HW=%d | ' % (Cc, HW) + ' | '.join('%.1f us (%.2f)' % (d[n][0], d[n][1] / 1e3) if n in d else '-' for n in names) + ' |')


if __name__ == '__main__':
    main()
