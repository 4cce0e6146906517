"""CPU-only checks of the boundary: the C-ABI library builds (hipcc cross-compiles gfx950 without
a GPU), loads, exports every symbol include/cnnq_hip.h declares, rejects bad arguments before
touching the device, and the product path refuses CPU tensors instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'cnnq_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(cnnq_[a-z0-9_]+)\s*\(', text)))


def test_library_builds_and_exports_every_declared_symbol():
    from cnn_quantization_amd import _build, _lib
    path = _build.build()
    assert os.path.exists(path)
    lib = _lib.load()
    declared = header_functions()
    assert declared, 'no functions parsed from the header'
    for name in declared:
        assert hasattr(lib, name), 'libcnnq_hip.so does not export %s' % name
    assert sorted(_lib.SIGNATURES) == declared, 'ctypes signatures out of sync with the header'
    assert b'gfx950' in lib.cnnq_version()


def test_no_kernel_instance_uses_scratch():
    """Round 6: three instances of the shipped library carried 20 bytes of scratch - a spilled tile row at
    the 168-register edge of the K = 32 tiles.  None may: the gfx950 code object inside libcnnq_hip.so is unbundled and every
    kernel's .private_segment_fixed_size read from its notes (llvm tools of the ROCm image; runs on the CPU)."""
    import subprocess
    import tempfile
    from cnn_quantization_amd import _build
    llvm = '/opt/rocm/lib/llvm/bin'
    if not os.path.exists(os.path.join(llvm, 'llvm-readelf')):
        pytest.skip('llvm tools not found')
    lib = _build.build()
    with tempfile.TemporaryDirectory() as d:
        co, fat = os.path.join(d, 'dev.co'), os.path.join(d, 'fat.bin')
        subprocess.run([llvm + '/llvm-objcopy', '-O', 'binary', '--only-section=.hip_fatbin', lib, fat], check=True)
        subprocess.run([llvm + '/clang-offload-bundler', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--input=' + fat,
                        '--output=' + co, '--unbundle'], check=True, capture_output=True)
        notes = subprocess.run([llvm + '/llvm-readelf', '--notes', co], check=True, capture_output=True, text=True).stdout
    kernels = notes.split('  - .agpr_count:')[1:]
    assert len(kernels) > 300, len(kernels)
    spilling = []
    for b in kernels:
        name = re.search(r'\.name:\s+(\S+)', b).group(1)
        if int(re.search(r'\.private_segment_fixed_size:\s+(\d+)', b).group(1)) > 0:
            spilling.append(name)
    assert not spilling, spilling


def test_enums_match_header():
    from cnn_quantization_amd import _lib as L
    text = open(os.path.join(ROOT, 'include', 'cnnq_hip.h')).read()
    for name, val in re.findall(r'(CNNQ_(?:STAT|MOM|DEV|QP|DIAG)_[A-Z_0-9]+)\s*=\s*(\d+)', text):
        assert getattr(L, name[len('CNNQ_'):]) == int(val), name
    for name, val in re.findall(r'(CNNQ_N[A-Z]+)\s*=\s*(\d+)', text):
        assert getattr(L, name[len('CNNQ_'):]) == int(val), name
    assert ctypes.sizeof(L.ParamsCfg) == 40


def test_argument_validation_needs_no_device():
    from cnn_quantization_amd import _lib as L
    lib = L.load()
    assert lib.cnnq_pc_groups(0, 4, 4, 1) == -1                      # CNNQ_EINVAL
    assert lib.cnnq_pc_groups(4, 1 << 20, 1 << 12, 1) == -2          # CNNQ_ERANGE: plane >= 2^31
    assert lib.cnnq_pc_groups(512, 64, 112 * 112, 1) > 0
    assert lib.cnnq_pc_moments(None, 1, 1, 1, 0, None, None) == -1
    assert lib.cnnq_pc_qdq(None, None, 1, 1, 1, None, None, None, 0, None) == -1
    assert lib.cnnq_pt_qdq(None, None, 0, None, None, None) == -1
    assert lib.cnnq_kld_hist(None, 1, 1, None, None, None) == -1
    assert lib.cnnq_kld_hist(8, 70000, 16, 8, 8, None) == -2         # rows beyond the grid's y extent
    assert lib.cnnq_kld_hist(8, 4, 1 << 31, 8, 8, None) == -2        # row length >= 2^31
    assert lib.cnnq_kld_search(None, 1, None, None, None, None) == -1
    with pytest.raises(L.CnnqError):
        L.check(-2, 'x')
    # ACIQ factor tables end at 8 bits (iq.py:14-41: the reference's dictionaries raise KeyError): wider codes with
    # laplace / gaus clipping are refused before anything is launched (the dummy pointers are never dereferenced)
    import ctypes
    cfg = L.ParamsCfg(num_bits=16, positive=0, clip=1, pstd=0., bit_alloc=0, prior_is_b=0, target=16., round_mode=1,
                      direct_range=0)
    assert lib.cnnq_pc_params(64, 4, ctypes.byref(cfg), 64, 64, None) == -1
    assert lib.cnnq_pc_aciq_qdq(64, 64, 2, 4, 16, ctypes.byref(cfg), 64, 64, 64, None) == -1
    cfg.clip = 2
    assert lib.cnnq_pc_params(64, 4, ctypes.byref(cfg), 64, 64, None) == -1
    from cnn_quantization_amd import ops
    for clip in ('laplace', 'gaus'):
        with pytest.raises(L.CnnqError):
            ops._params_cfg(16, False, clip, False, False, None, True, False)
    assert ops._params_cfg(16, False, '2std', False, False, None, True, False).clip == 3
    assert ops._params_cfg(16, False, 'no', False, False, None, True, False).clip == 0


def test_geometry_plan_covers_resnet50_shapes():
    """Every ResNet-50 / VGG-16 activation shape of the BASELINE configs gets a valid plan with a
    bounded number of partial groups."""
    from cnn_quantization_amd import _lib as L
    lib = L.load()
    shapes = [(512, 64, 112 * 112), (512, 256, 56 * 56), (512, 512, 28 * 28), (512, 1024, 14 * 14),
              (512, 2048, 7 * 7), (512, 512, 7 * 7), (512, 64, 224 * 224), (512, 512, 14 * 14),
              (1, 512, 512 * 3 * 3), (1, 1000, 2048), (1, 512, 25690112 // 512)]
    for N, C, HW in shapes:
        for al in (0, 1):
            g = lib.cnnq_pc_groups(N, C, HW, al)
            assert 0 < g <= 64 * 8192, (N, C, HW, al, g)


def test_stats_route_names_the_kernel_family_without_a_device():
    """cnnq_pc_stats_route (round 6): which of the three routes the seven statistics take for a geometry - the planner's answer,
    no device needed.  ResNet-50 at batch 512: flat tiles up to 256 members per channel (112x112: 196), row pieces for
    one-channel-per-lane rows (14x14), the chain for the large 7x7 tensors (at the 64-sample shard the small 7x7 one is a row-piece launch too) and for
    shapes without a 16-byte tiling."""
    from cnn_quantization_amd import _lib as L
    lib = L.load()
    ws = 18 << 20
    want = {(512, 64, 112 * 112): 1, (512, 256, 56 * 56): 1, (512, 512, 28 * 28): 1, (512, 1024, 14 * 14): 2, (512, 256, 14 * 14): 2,
            (512, 2048, 7 * 7): 0, (512, 512, 7 * 7): 0, (64, 512, 7 * 7): 2, (64, 2048, 7 * 7): 0, (512, 64, 224 * 224): 0,
            (3, 16, 5 * 9): 2, (3, 5, 7 * 9): 0}
    for (N, C, HW), r in want.items():
        assert lib.cnnq_pc_stats_route(N, C, HW, 1, ws, 0) == r, (N, C, HW)
    assert lib.cnnq_pc_stats_route(512, 512, 49, 1, ws, 8) == 2         # flags bit 3 (tests) lifts the size rule
    assert lib.cnnq_pc_stats_route(512, 256, 56 * 56, 1, 1 << 20, 0) == 0     # a workspace too small for the plan: the chain
    assert lib.cnnq_pc_stats_route(512, 256, 56 * 56, 0, ws, 0) == 0        # no 16-byte alignment, no 16-byte tiling


def test_no_cpu_fallback():
    from cnn_quantization_amd import ops, int_quantization
    from cnn_quantization_amd._lib import CnnqError
    x = torch.randn(2, 4, 3, 3)
    with pytest.raises(CnnqError):
        ops.act_qdq_per_channel(x, 4)
    with pytest.raises(CnnqError):
        ops.pc_stats(x, 2, 4, 9)
    with pytest.raises(RuntimeError):
        int_quantization.float2gemmlowp(x, 1.0, -0.5, 8, False, True, None)
    # range <= 0 returns the input object itself, as kernels/gemmlowp.cu:31-32 does
    assert int_quantization.float2gemmlowp(x, 0.0, 0.0, 8, False, True, None) is x


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from cnn_quantization_amd import _build, _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_build, 'LIB', str(tmp_path / 'libcnnq_hip.so'))
    with pytest.raises(_lib.CnnqError, match='no CPU fallback'):
        _lib.load()


def test_c_restatement_matches_torch_oracle():
    """oracle/qdq_core.c (plain C, IEEE divide / rintf / roundf) against the torch-based oracle:
    an independent check of the arithmetic the HIP kernels implement."""
    import numpy as np
    from oracle import build_oracle, quant_oracle as O
    lib = build_oracle.load()
    gen = torch.Generator().manual_seed(3)
    N, C, H, W = 3, 6, 5, 7
    x = (torch.randn(N, C, H, W, generator=gen) * 2).contiguous()
    bits = torch.tensor([0., 1., 2., 4., 8., 3.])
    t = x.transpose(0, 1).contiguous().view(C, -1)
    mn, mx = t.min(-1)[0], t.max(-1)[0]
    y_ref, codes_ref, scale, zp, qmax = O.qdq_core(t, mx - mn, mn, bit_alloc=bits, return_parts=True)
    y_ref = y_ref.view(C, N, H, W).transpose(0, 1).contiguous()
    y = torch.empty_like(x)
    codes = torch.empty(x.shape, dtype=torch.uint8)
    sc, z, qm = scale.contiguous(), zp.contiguous(), qmax.contiguous()
    lib.oracle_pc_qdq(x.data_ptr(), y.data_ptr(), codes.data_ptr(), N, C, H * W, sc.data_ptr(), z.data_ptr(),
                      qm.data_ptr())
    assert np.array_equal(y.numpy().view(np.uint32), y_ref.numpy().view(np.uint32))
    assert torch.equal(codes.float(), codes_ref.view(C, N, H, W).transpose(0, 1))
    # per-tensor kernel restatement
    v = torch.randn(1001, generator=gen) * 3
    mn, mx = float(v.min()), float(v.max())
    for etz in (True, False):
        ref = O.float2gemmlowp(v, mx - mn, mn, 8, False, etz)
        qmaxf = 255.0
        scale_f = np.float32(np.float32(mx - mn) / np.float32(qmaxf))
        zpf = np.float32(-np.float32(mn) / scale_f)
        zpf = np.float32(np.copysign(np.floor(np.abs(zpf) + np.float32(0.5)), zpf))
        shift = zpf if etz else np.float32(-np.float32(mn))
        out = torch.empty_like(v)
        lib.oracle_pt_qdq(v.data_ptr(), out.data_ptr(), v.numel(), float(scale_f), float(shift), qmaxf, int(etz))
        assert torch.equal(out, ref), etz


def test_header_is_plain_c99_and_a_c_client_links(tmp_path):
    """include/cnnq_hip.h must be consumable by a C compiler (the boundary is a C ABI), and a plain C program
    must link against libcnnq_hip.so without any C++ runtime of its own (tests/c/cabi_demo.c; run on the GPU by
    tests/test_cabi_c_gpu.py)."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    inc = os.path.join(ROOT, 'include')
    src = tmp_path / 't.c'
    src.write_text('#include "cnnq_hip.h"\nint main(void) { return cnnq_pc_groups(1, 1, 1, 1) > 0 ? 0 : 1; }\n')
    subprocess.run([gcc, '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-fsyntax-only', '-I', inc, str(src)],
                   check=True)
    if not os.path.isdir('/opt/rocm/include/hip'):
        pytest.skip('no HIP headers')
    from cnn_quantization_amd import _build
    _build.build()
    exe = tmp_path / 'cabi_demo'
    subprocess.run([gcc, '-std=c99', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I', inc,
                    os.path.join(ROOT, 'tests', 'c', 'cabi_demo.c'), '-L', os.path.dirname(_build.LIB), '-lcnnq_hip',
                    '-L/opt/rocm/lib', '-lamdhip64', '-lm', '-o', str(exe)], check=True)
    assert exe.exists()
