"""ctypes binding of libcnnq_hip.so (include/cnnq_hip.h).  There is NO fallback: if the
library is missing or a call fails the error is raised to the caller."""
import ctypes
import os

from . import _build

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_int64
_F = ctypes.c_float

# rows of the device tables (mirrors the enums of include/cnnq_hip.h)
STAT_MIN, STAT_MAX, STAT_MEAN, STAT_STD, STAT_B, STAT_KURT, STAT_STD_POS, NSTAT = 0, 1, 2, 3, 4, 5, 6, 7
MOM_MIN, MOM_MAX, MOM_SUM, MOM_SUMSQ, MOM_COUNT, MOM_SUM_RELU, MOM_SUMSQ_RELU, NMOM = 0, 1, 2, 3, 4, 5, 6, 7
DEV_ABS, DEV_Z4, NDEV = 0, 1, 2
QP_SCALE, QP_ZP, QP_QMAX, NQP = 0, 1, 2, 3
MT_DELTA, MT_CMIN, MT_CMAX, MT_OMEGA, MT_ALPHA, MT_WSTART, NMT = 0, 1, 2, 3, 4, 5, 6
MT_HIST_BINS = 131072
MT_HIST_WINDOW, MT_HIST_REPLICAS = 128, 256


def mt_hist_words(C):
    """CNNQ_MT_HIST_WORDS(C) of include/cnnq_hip.h"""
    return MT_HIST_BINS + 2 + 2 * C + MT_HIST_REPLICAS * MT_HIST_WINDOW + 1
KLD_BINS, KLD_QBINS, KLD_NCAND = 2001, 15, 994
DIAG_BITS, DIAG_ALPHA, DIAG_DELTA, DIAG_OFFSET, NDIAG = 0, 1, 2, 3, 4


class ParamsCfg(ctypes.Structure):
    _fields_ = [('num_bits', ctypes.c_int32), ('positive', ctypes.c_int32), ('clip', ctypes.c_int32),
                ('pstd', ctypes.c_float), ('bit_alloc', ctypes.c_int32), ('prior_is_b', ctypes.c_int32),
                ('target', ctypes.c_double), ('round_mode', ctypes.c_int32), ('direct_range', ctypes.c_int32)]


class XRankCtx(ctypes.Structure):
    """cnnq_xrank_ctx of include/cnnq_hip.h"""
    _fields_ = [('windows', ctypes.c_void_p), ('rank', ctypes.c_int32), ('world', ctypes.c_int32), ('cmax', ctypes.c_int32),
                ('seq', ctypes.c_uint32), ('seq_dev', ctypes.c_void_p), ('status', ctypes.c_void_p),
                ('timeout_ticks', ctypes.c_int64)]


# every symbol include/cnnq_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    'cnnq_version': (ctypes.c_char_p, []),
    'cnnq_pc_groups': (_I, [_L, _L, _L, _I]),
    'cnnq_plan_describe': (_I, [_L, _L, _L, _I, _I, ctypes.POINTER(ctypes.c_int32)]),
    'cnnq_pc_moments': (_I, [_P, _L, _L, _L, _I, _P, _P]),
    'cnnq_pc_combine': (_I, [_P, _I, _L, _I, _P, _P, _P]),
    'cnnq_pc_absdev': (_I, [_P, _L, _L, _L, _P, _I, _P, _P]),
    'cnnq_pc_combine_dev': (_I, [_P, _I, _L, _P, _I, _P, _P, _P]),
    'cnnq_pc_stats_workspace': (ctypes.c_size_t, [_L, _L, _L, _I]),
    'cnnq_pc_stats': (_I, [_P, _L, _L, _L, _I, _I, _I, _P, _P, _P, _P]),
    'cnnq_pc_stats_single': (_I, [_P, _L, _L, _L, _I, _I, _I, _P, ctypes.c_size_t, _P, _P, ctypes.c_uint32, _P]),
    'cnnq_pc_stats_auto': (_I, [_P, _L, _L, _L, _I, _I, _I, _P, _P, ctypes.c_size_t, _P, _P, _P]),
    'cnnq_pc_stats_route': (_I, [_L, _L, _L, _I, ctypes.c_size_t, ctypes.c_uint32]),
    'cnnq_pc_params': (_I, [_P, _L, ctypes.POINTER(ParamsCfg), _P, _P, _P]),
    'cnnq_pc_qdq': (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _I, _P]),
    'cnnq_pc_quantize_pack4': (_I, [_P, _P, _L, _L, _L, _P, _P]),
    'cnnq_pc_dequantize_pack4': (_I, [_P, _P, _L, _L, _L, _P, _P]),
    'cnnq_pc_quantize_u8': (_I, [_P, _P, _L, _L, _L, _P, _P]),
    'cnnq_pc_dequantize_u8': (_I, [_P, _P, _L, _L, _L, _P, _P]),
    'cnnq_pc_packed_layout': (_I, [_P, _L, _L, _P, _P]),
    'cnnq_pc_quantize_packed': (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _P]),
    'cnnq_pc_quantize_packed_form': (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _I, _P]),
    'cnnq_pc_dequantize_packed': (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _P]),
    'cnnq_pc_dequantize_packed_form': (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _I, _P]),
    'cnnq_xrank_window_bytes': (ctypes.c_size_t, [_I, _I]),
    'cnnq_xrank_alloc': (_I, [_I, _I, _P, _P]),
    'cnnq_pc_minmax_qdq_xrank': (_I, [_P, _P, _L, _L, _L, _I, _I, _P, _P, ctypes.c_size_t, _P, _I, _I, _I, ctypes.c_uint32, _P, _L,
                                      _P]),
    'cnnq_pc_minmax_qdq_xrank_dev': (_I, [_P, _P, _L, _L, _L, _I, _I, _P, _P, ctypes.c_size_t, _P, _I, _I, _I, _P, _P, _L, _P, _P, _P]),
    'cnnq_pc_minmax_qdq_xrank_seq': (_I, [_P, _P, _L, _L, _L, _I, _I, _P, _P, ctypes.c_size_t, _P, _I, _I, _I, ctypes.c_uint32, _P, _I, _P, _L,
                                          _P, _P, _P]),
    'cnnq_hist_replicas_fold': (_I, [_P, _P, _P]),
    'cnnq_pc_minmax': (_I, [_P, _L, _L, _L, _P, _P]),
    'cnnq_pc_minmax_strided': (_I, [_P, _L, _L, _L, _L, _P, _P]),
    'cnnq_pc_qdq_strided': (_I, [_P, _P, _L, _L, _L, _L, _P, _P, _P, _I, _P]),
    'cnnq_pc_minmax_reduce': (_I, [_P, _I, _L, _P, _P]),
    'cnnq_pc_minmax_params': (_I, [_P, _I, _L, _I, _I, _P, _P]),
    'cnnq_pc_minmax_qdq': (_I, [_P, _P, _L, _L, _L, _I, _I, _P, _P, _P, _P, _P]),
    'cnnq_pc_resident_describe': (_I, [_L, _L, _L, ctypes.POINTER(ctypes.c_int32)]),
    'cnnq_pc_minmax_qdq_resident': (_I, [_P, _P, _L, _L, _L, _I, _I, _P, _P, _P]),
    'cnnq_group_ws_alloc': (_I, [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    'cnnq_group_ws_free': (_I, [_P]),
    'cnnq_group_ws_status': (_I, [_P, ctypes.POINTER(ctypes.c_uint32)]),
    'cnnq_group_ws_status_clear': (_I, [_P]),
    'cnnq_pc_group_workspace': (ctypes.c_size_t, [_L, _L, _L]),
    'cnnq_group_ws_at_rest': (_I, [_P, ctypes.POINTER(ctypes.c_uint64)]),
    'cnnq_pc_group_describe': (_I, [_L, _L, _L, ctypes.POINTER(ctypes.c_int32)]),
    'cnnq_pc_minmax_qdq_group': (_I, [_P, _P, _L, _L, _L, _I, _I, _P, _P, _P, ctypes.c_uint32, _P]),
    'cnnq_pc_minmax_local': (_I, [_P, _L, _L, _L, _P, _P, _P]),
    'cnnq_pc_minmax_local_auto': (_I, [_P, _L, _L, _L, _P, _P, ctypes.c_size_t, _P, _P]),
    'cnnq_pc_gathered_qdq': (_I, [_P, _P, _L, _L, _L, _P, _I, _I, _I, _P, _P]),
    'cnnq_pc_minmax_qdq_workspace': (ctypes.c_size_t, [_L, _L, _L]),
    'cnnq_pc_minmax_qdq_auto': (_I, [_P, _P, _L, _L, _L, _I, _I, _P, _P, ctypes.c_size_t, _I, _P]),
    'cnnq_hist_replica_bytes': (ctypes.c_size_t, []),
    'cnnq_pc_minmax_qdq_single': (_I, [_P, _P, _L, _L, _L, _I, _I, _P, ctypes.c_size_t, _P, _P, _P, _P, _P, _P]),
    'cnnq_entropy_replicas': (_I, [_P, _P, _P]),
    'cnnq_entropy_replicas_batch': (_I, [_P, _I, _P, _P]),
    'cnnq_midtread_entropy_batch': (_I, [_I, _P, _P, _P, _P, _P, _P]),
    'cnnq_pc_aciq_workspace': (ctypes.c_size_t, [_L, _L, _L, _I]),
    'cnnq_pc_aciq_qdq': (_I, [_P, _P, _L, _L, _L, ctypes.POINTER(ParamsCfg), _P, _P, _P, _P]),
    'cnnq_pc_aciq_qdq_single': (_I, [_P, _P, _L, _L, _L, ctypes.POINTER(ParamsCfg), _P, _P, ctypes.c_size_t, _P, _P, _P, _P, _P,
                                     ctypes.c_uint32, _P]),
    'cnnq_pc_aciq_qdq_auto': (_I, [_P, _P, _L, _L, _L, ctypes.POINTER(ParamsCfg), _P, _P, ctypes.c_size_t, _P, _P, _P]),
    'cnnq_pc_aciq_fused_xrank': (_I, [_P, _P, _L, _L, _L, ctypes.POINTER(ParamsCfg), _P, _P, ctypes.c_size_t, _P, _P, _P, _P,
                                      ctypes.POINTER(XRankCtx), ctypes.c_uint32, _P]),
    'cnnq_pc_midtread_fused_xrank': (_I, [_P, _P, _L, _L, _L, ctypes.c_double, _I, _P, _I, _P, _P, ctypes.c_size_t, _P, _P, _P, _P,
                                          ctypes.POINTER(XRankCtx), ctypes.c_uint32, _P]),
    'cnnq_pc_stats_xrank': (_I, [_P, _L, _L, _L, _I, _I, _I, _P, _P, ctypes.c_size_t, _P, _P, ctypes.POINTER(XRankCtx), ctypes.c_uint32,
                                 _P]),
    'cnnq_pc_weight_correct': (_I, [_P, _L, _L, _P, _P, _I, _I, _P]),
    'cnnq_pc_bcorr_sums': (_I, [_P, _P, _L, _L, _L, _I, _P, _P]),
    'cnnq_pc_qdq_bcorr_sums': (_I, [_P, _L, _L, _L, _P, _I, _P, _P]),
    'cnnq_pc_qdq_bcorr': (_I, [_P, _P, _L, _L, _L, _P, _P, _I, _P]),
    'cnnq_pc_bcorr_bias': (_I, [_P, _I, _L, _P, _P, _P]),
    'cnnq_pc_bcorr_apply': (_I, [_P, _L, _L, _L, _P, _P]),
    'cnnq_pc_midtread_params': (_I, [_P, _L, ctypes.c_double, _I, _I, _P, _I, _P, _P]),
    'cnnq_pc_midtread_qdq': (_I, [_P, _P, _L, _L, _L, _P, _I, _P, _P, _P]),
    'cnnq_pc_midtread_qdq_single': (_I, [_P, _P, _L, _L, _L, ctypes.c_double, _I, _P, _I, _P, _P, ctypes.c_size_t, _P, _P, _P,
                                         ctypes.c_uint32, _P]),
    'cnnq_midtread_entropy': (_I, [_P, _P, _L, _L, _P, _P]),
    'cnnq_midtread_entropy_count': (_I, [_P, _P, _L, _P, _P, _P]),
    'cnnq_entropy': (_I, [_P, _I, _P, _P]),
    'cnnq_pt_setup': (_I, [ctypes.POINTER(_F), _P, _L, _I, _I, _I, _I, _I, _I, _P, _P]),
    'cnnq_pt_qdq': (_I, [_P, _P, _L, _P, _P, _P]),
    'cnnq_pt_minmax_qdq_fused': (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _I, _P, ctypes.c_size_t, _P, _P]),
    'cnnq_xrank_open': (_I, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]),
    'cnnq_xrank_close': (_I, [_P]),
    'cnnq_xrank_free': (_I, [_P]),
    'cnnq_kld_hist': (_I, [_P, _L, _L, _P, _P, _P]),
    'cnnq_kld_search': (_I, [_P, _L, _P, _P, _P, _P]),
}

_lib = None


ENOTSUP = -3


class CnnqError(RuntimeError):
    pass


def lib_path():
    # CNNQ_HIP_LIB: development aid - load another build of the same library (kernel experiments)
    return os.environ.get('CNNQ_HIP_LIB') or _build.LIB


def load():
    """Load (once) and return the ctypes handle; raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise CnnqError('%s is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                        '(hipcc --offload-arch=gfx950); there is no CPU fallback' % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        kind = {-1: 'CNNQ_EINVAL', -2: 'CNNQ_ERANGE', -3: 'CNNQ_ENOTSUP'}.get(rc, 'hipError %d' % rc)
        raise CnnqError('%s failed: %s' % (what, kind))
