"""Tensor-level wrappers over the C ABI (include/cnnq_hip.h): device pointers, sizes and the
current HIP stream go straight to libcnnq_hip.so.  torch supplies memory and streams only.

No function here synchronises with the host; none has a CPU path - CPU tensors are rejected
and a missing library raises (cnn_quantization_amd._lib.load)."""
import ctypes
import os

import torch

from . import _lib as L
from . import distributed as D


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# the raw handle of the current stream of a device: torch keeps a direct accessor (the one its own compiled code
# uses); torch.cuda.current_stream() builds a Stream object per call, ~1.5 us, three times per hot call before
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None) or (
    lambda index: torch.cuda.current_stream(index).cuda_stream)


def _stream(t):
    return ctypes.c_void_p(_raw_stream(t.device.index))


def _dev_f32(x, what='tensor'):
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise L.CnnqError('%s must be a CUDA/HIP tensor (there is no CPU path)' % what)
    if x.dtype != torch.float32:
        raise L.CnnqError('%s must be float32, got %s' % (what, x.dtype))
    if x.device.index != torch.cuda.current_device():
        # kernels are enqueued in the calling thread's current device context (one process per GPU is the
        # deployment model; in-process multi-GPU callers must enter torch.cuda.device(x.device) first)
        raise L.CnnqError('%s is on %s but the current device is cuda:%d' % (what, x.device, torch.cuda.current_device()))
    if x.requires_grad:
        x = x.detach()
    return x if x.is_contiguous() else x.contiguous()


# switches, read ONCE at import (the hot call used to look three of them up per tensor: weak #10 of the round-2
# review); reload_switches() re-reads them for callers that change the environment afterwards (tests, tools)
def reload_switches():
    global _RESIDENT, _SINGLE_CODES, _DIRECT_RCCL, _PT_FUSED, _XRANK_ON, _ACIQ_SINGLE
    _ACIQ_SINGLE = os.environ.get('CNNQ_ACIQ_SINGLE', '1') != '0'      # 0: the ACIQ path always takes the five-launch chain (A/B)
    _XRANK_ON = D.xrank_mode() != '0'                                  # opt-in (CNNQ_XRANK=1 / auto, D.set_xrank_mode): sharded config 2 exchanges INSIDE the single launch
    _PT_FUSED = os.environ.get('CNNQ_PT_FUSED', '0') == '1'           # 1: config 1 in one launch (slower: see ops.minmax_qdq_per_tensor)
    _RESIDENT = os.environ.get('CNNQ_RESIDENT', '1') != '0'            # 0: never take a single-launch kernel
    _SINGLE_CODES = os.environ.get('CNNQ_SINGLE_CODES', '1') != '0'    # 0: codes / entropy requests take the chain
    _DIRECT_RCCL = os.environ.get('CNNQ_DIRECT_RCCL', '1')
    _XPLAN.clear()


_XPLAN = {}     # per (group, device, stream, geometry): what the multi-GPU hot call needs, looked up once
reload_switches()

_SCRATCH = {}
_WS_BYTES = {}


def release_plans():
    """Drop the cached multi-GPU exchange plans (they hold process groups and direct RCCL communicators): called by
    rccl.close_all(), so that no plan outlives its communicator."""
    _XPLAN.clear()


def release_workspaces():
    """Give back every cached buffer: scratch tensors, the replica histograms and the fine-grained exchange
    workspaces of all streams (cnnq_group_ws_free).  Synchronises the device first; later calls re-allocate."""
    torch.cuda.synchronize()
    release_plans()
    _SCRATCH.clear()
    _HIST_REP.clear()
    _ENT_TABLES.clear()
    lib = L.load()
    for ws in list(_GROUP_WS.values()) + [w for pool in _GROUP_POOL.values() for w in pool]:
        lib.cnnq_group_ws_free(ws)
    _GROUP_WS.clear()
    _GROUP_POOL.clear()


def _scratch(x, tag, nbytes, st=None):
    """A per-(device, stream, tag) scratch buffer (uint8, at least nbytes), grown on demand and never returned
    to callers: launches on one stream are ordered, so consecutive calls may share it.  Saves the 2-4
    torch.empty calls (8-15 us of host time) the small layers would otherwise pay per call."""
    key = (x.device.index, _raw_stream(x.device.index) if st is None else st, tag)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            # inside a stream capture the buffer comes from the graph's private pool: hand it out WITHOUT caching it
            # (a cached one would later serve eager calls from memory that belongs to the graph); the pool keeps the
            # block for the graph's replays and reuses it in capture order
            return torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=x.device)
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=x.device)
        _SCRATCH[key] = buf
    return buf


def _out_like(x, out):
    """The result buffer: a new tensor like x, or the caller's - which must match x exactly."""
    if out is None:
        return torch.empty_like(x)
    if not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
            and out.shape == x.shape and out.device == x.device):
        raise L.CnnqError('out must be a contiguous float32 tensor on %s shaped like the input' % (x.device,))
    nbytes = x.numel() * 4
    if out.data_ptr() < x.data_ptr() + nbytes and x.data_ptr() < out.data_ptr() + nbytes:
        # the kernels declare x and y __restrict__, and a single-launch workgroup whose bounded wait expires recomputes
        # its channels' extrema by re-reading x after other members may have stored y (include/cnnq_hip.h: x != y).
        # Byte ranges, not storages: two views of one arena that do not overlap are fine.
        raise L.CnnqError('out must not overlap the input (in-place quantization is not supported)')
    return out


def geometry(x, per_channel_dim=1):
    """(N, C, HW) of a tensor addressed as x[N][C][HW].  4-D activations: channel dim 1.
    per_channel_dim=0 (weights [OFM, ...], iq.py:455): N = 1, C = OFM."""
    if per_channel_dim == 0:
        return 1, x.shape[0], x.numel() // x.shape[0]
    n, c = x.shape[0], x.shape[1]
    return n, c, x.numel() // (n * c)


# ------------------------------------------------------------------------------------- statistics
def pc_moments(x, N, C, HW, want_relu=False):
    """Pass A partial records [G, NMOM, C] (float64)."""
    lib = L.load()
    G = lib.cnnq_pc_groups(N, C, HW, int(x.data_ptr() % 16 == 0))
    if G <= 0:
        L.check(G, 'cnnq_pc_groups(%d,%d,%d)' % (N, C, HW))
    part = torch.empty((G, L.NMOM, C), dtype=torch.float64, device=x.device)
    L.check(lib.cnnq_pc_moments(_ptr(x), N, C, HW, int(want_relu), _ptr(part), _stream(x)), 'cnnq_pc_moments')
    return part


def pc_combine(part, has_relu=False, stats=None, want_mom=True):
    """Merge [G, NMOM, C] records -> (mom [NMOM, C] f64, stats [NSTAT, C] f32)."""
    lib = L.load()
    G, _, C = part.shape
    mom = torch.empty((L.NMOM, C), dtype=torch.float64, device=part.device) if want_mom else None
    if stats is None:
        stats = torch.zeros((L.NSTAT, C), dtype=torch.float32, device=part.device)
    L.check(lib.cnnq_pc_combine(_ptr(part), G, C, int(has_relu), _ptr(mom), _ptr(stats), _stream(part)),
            'cnnq_pc_combine')
    return mom, stats


def pc_absdev(x, N, C, HW, stats, want_kurt=False):
    lib = L.load()
    G = lib.cnnq_pc_groups(N, C, HW, int(x.data_ptr() % 16 == 0))
    part2 = torch.empty((G, L.NDEV, C), dtype=torch.float64, device=x.device)
    L.check(lib.cnnq_pc_absdev(_ptr(x), N, C, HW, _ptr(stats), int(want_kurt), _ptr(part2), _stream(x)),
            'cnnq_pc_absdev')
    return part2


def pc_combine_dev(part2, mom, stats=None, want_kurt=False, want_sums=False):
    lib = L.load()
    G, _, C = part2.shape
    dev = torch.empty((L.NDEV, C), dtype=torch.float64, device=part2.device) if want_sums else None
    L.check(lib.cnnq_pc_combine_dev(_ptr(part2), G, C, _ptr(mom), int(want_kurt), _ptr(dev), _ptr(stats),
                                    _stream(part2)), 'cnnq_pc_combine_dev')
    return dev


def pc_stats(x, N, C, HW, need_b=False, need_kurt=False, need_relu=False, group=None, local_only=False):
    """Per-channel statistics table [NSTAT, C] of x[N][C][HW] (rows MIN, MAX, MEAN, STD always;
    B / KURT / STD_POS on request) plus the merged moment record [NMOM, C].

    With a process group of world size > 1 the tensor is this rank's batch shard: the fp64
    moment records are all-gathered (<= 7*C*8 bytes per rank, latency-bound on xGMI) and merged
    in rank order on every rank, so all ranks hold the statistics of the GLOBAL batch."""
    x = _dev_f32(x, 'x')
    world = 1 if local_only else D.world_size(group)
    if not local_only and (world > 1 or D.forced_exchange()) and _ACIQ_SINGLE and _RESIDENT:
        # the batch is sharded and the group has a (verified) in-launch exchange: the table of the GLOBAL batch from one read of
        # this rank's shard (cnnq_pc_stats_xrank; round 6) - every rank takes this route or none does
        res = _pc_stats_xrank(x, N, C, HW, need_b, need_kurt, need_relu, group)
        if res is not None:
            return res
    if world == 1 and (local_only or not D.forced_exchange()):
        # one C call, one cached workspace (cnnq_pc_stats_auto)
        lib = L.load()
        nbytes = lib.cnnq_pc_stats_workspace(N, C, HW, int(x.data_ptr() % 16 == 0))
        if nbytes == 0:
            L.check(min(lib.cnnq_pc_groups(N, C, HW, 1), -1), 'cnnq_pc_groups(%d,%d,%d)' % (N, C, HW))
        stats = torch.empty((L.NSTAT, C), dtype=torch.float32, device=x.device)
        mom = torch.empty((L.NMOM, C), dtype=torch.float64, device=x.device)
        st = _raw_stream(x.device.index)
        gws = _group_workspace(x, st) if (_ACIQ_SINGLE and _RESIDENT) else None
        # one launch that reads x once where the shape has a flat plan (cnnq_pc_stats_single), else the three-launch chain
        L.check(lib.cnnq_pc_stats_auto(_ptr(x), N, C, HW, int(bool(need_b)), int(bool(need_kurt)), int(bool(need_relu)),
                                       _ptr(_scratch(x, 'stats', nbytes, st)), gws, GROUP_WS_BYTES if gws is not None else 0,
                                       _ptr(mom), _ptr(stats), st), 'cnnq_pc_stats')
        return stats, mom
    # the collective route (ranks sharing a GPU, CNNQ_XRANK=0, after a recovery; also a 1-rank group under CNNQ_FORCE_EXCHANGE=1,
    # which times the real collectives on one GPU): the chain's passes with an all_gather of the fp64 records behind each
    exchanging = world > 1 or D.forced_exchange()
    part = pc_moments(x, N, C, HW, need_relu)
    if exchanging:
        mom_local, _ = pc_combine(part, need_relu)
        part = D.all_gather_records(mom_local, group)
    mom, stats = pc_combine(part, need_relu)
    if need_b or need_kurt:
        part2 = pc_absdev(x, N, C, HW, stats, need_kurt)
        if exchanging:
            dev_local = pc_combine_dev(part2, mom, None, need_kurt, want_sums=True)
            part2 = D.all_gather_records(dev_local, group)
        pc_combine_dev(part2, mom, stats, need_kurt)
    return stats, mom


def _xrank_for(group, C, words=1):
    """The group's in-launch exchange (D.xrank_exchange: opt-in mode, verified against the collective on every rank) when a launch
    over C channels with `words` slots per channel fits its windows, else None.  Depends on the group and the layer only - never
    on this rank's shard - so all ranks decide alike."""
    if not _XRANK_ON:
        return None
    xr = D.xrank_exchange(group)
    return xr if (xr is not None and xr.fits(C, words)) else None


def _xrank_plan(kind, x, N, C, HW, group, st, ws_bytes):
    """What the sharded single-call routes need per (group, stream, geometry), looked up once: the exchange, the scratch
    workspace and the group workspace.  None: the group has no in-launch exchange.  (Under a stream capture nothing is cached:
    a capture-time scratch buffer belongs to the graph's pool.)"""
    key = (kind, id(group), x.device.index, st, N, C, HW, x.data_ptr() % 16 == 0)
    plan = _XPLAN.get(key)
    if plan is None:
        xr = _xrank_for(group, C, 8)
        if xr is None:
            plan = False
        else:
            plan = (xr, group, _scratch(x, kind, ws_bytes(), st), _group_workspace(x, st))
        if not torch.cuda.is_current_stream_capturing():
            _XPLAN[key] = plan
    return plan or None


def _pc_stats_xrank(x, N, C, HW, need_b, need_kurt, need_relu, group, flags=0):
    """Config 4 of a batch shard in ONE host call and - where the shard has a flat-tile plan - one launch and one read of x
    (cnnq_pc_stats_xrank: k_stats_flat with the cross-rank stage; otherwise the chain's two passes with their records made global
    through the same window slots).  No collective.  Returns (stats, mom) of the GLOBAL batch, or None when the group has no
    in-launch exchange."""
    lib = L.load()
    st = _raw_stream(x.device.index)

    def nbytes():
        n = lib.cnnq_pc_stats_workspace(N, C, HW, int(x.data_ptr() % 16 == 0))
        if n == 0:
            L.check(min(lib.cnnq_pc_groups(N, C, HW, 1), -1), 'cnnq_pc_groups(%d,%d,%d)' % (N, C, HW))
        return n
    plan = _xrank_plan('stats', x, N, C, HW, group, st, nbytes)
    if plan is None:
        return None
    xr, _, ws, gws = plan
    stats = torch.empty((L.NSTAT, C), dtype=torch.float32, device=x.device)
    mom = torch.empty((L.NMOM, C), dtype=torch.float64, device=x.device)
    ctx = xr.ctx(st)
    rc = lib.cnnq_pc_stats_xrank(x.data_ptr(), N, C, HW, 1 if need_b else 0, 1 if need_kurt else 0, 1 if need_relu else 0, ws.data_ptr(),
                                 gws, GROUP_WS_BYTES if gws is not None else 0, mom.data_ptr(), stats.data_ptr(), ctypes.byref(ctx),
                                 int(flags), st)
    if rc:
        L.check(rc, 'cnnq_pc_stats_xrank')
    return stats, mom


def _aciq_ws_bytes(x, N, C, HW):
    lib = L.load()
    al = int(x.data_ptr() % 16 == 0)
    key = ('aciq', N, C, HW, al)
    nbytes = _WS_BYTES.get(key)
    if nbytes is None:
        nbytes = _WS_BYTES[key] = (lib.cnnq_pc_aciq_workspace(N, C, HW, al) + 15) // 16 * 16
    return nbytes


def _aciq_qdq_xrank(x, N, C, HW, cfg, group, want_parts, out, flags=0):
    """Config 3 of a batch shard (Laplace clipping, optional bit allocation on the 'gaus' prior) in ONE host call
    (cnnq_pc_aciq_fused_xrank; round 6): pass A, its record made global through the exchange windows, (bit allocation), ONE launch
    for pass B + the ranks' sums of |x - mean| + parameters + Q/DQ - the four launches of one GPU, 12 bytes per element, no
    collective, where the chain moves 16 around two.  Returns y [, parts], or None when the group has no in-launch exchange (the
    caller takes the chain)."""
    lib = L.load()
    st = _raw_stream(x.device.index)
    plan = _xrank_plan('aciq', x, N, C, HW, group, st,
                       lambda: _aciq_ws_bytes(x, N, C, HW) + (L.NSTAT + L.NQP + L.NDIAG) * C * 4 + L.NMOM * C * 8)
    if plan is None:
        return None
    xr, _, ws, gws = plan
    y = _out_like(x, out)
    if want_parts:
        tabs = torch.empty((L.NSTAT + L.NQP + L.NDIAG, C), dtype=torch.float32, device=x.device)
        mom = torch.empty((L.NMOM, C), dtype=torch.float64, device=x.device)
        tp, mp = tabs.data_ptr(), mom.data_ptr()
    else:                       # nobody outside the call reads the tables: they live behind the workspace
        mp = ws.data_ptr() + _aciq_ws_bytes(x, N, C, HW)
        tp = mp + L.NMOM * C * 8
    sp, qp, dp = tp, tp + L.NSTAT * C * 4, tp + (L.NSTAT + L.NQP) * C * 4
    ctx = xr.ctx(st)
    rc = lib.cnnq_pc_aciq_fused_xrank(x.data_ptr(), y.data_ptr(), N, C, HW, ctypes.byref(cfg), ws.data_ptr(), gws,
                                      GROUP_WS_BYTES if gws is not None else 0, sp, mp, qp, dp, ctypes.byref(ctx), int(flags), st)
    if rc:
        L.check(rc, 'cnnq_pc_aciq_fused_xrank')
    if want_parts:
        return y, dict(stats=tabs[:L.NSTAT], qp=tabs[L.NSTAT:L.NSTAT + L.NQP], diag=tabs[L.NSTAT + L.NQP:], mom=mom)
    return y


def _mid_tread_qdq_xrank(x, N, C, HW, target, sym, tabs_mt, group, want_entropy, want_parts, flags=0):
    """Config 5 of a batch shard: as _aciq_qdq_xrank with the bin allocation and MODE 1 of the fused kernels
    (cnnq_pc_midtread_fused_xrank); the ranks' code counts are summed before the entropy (the one collective left: an integer
    all-reduce of the count table).  Returns what mid_tread_qdq returns, or None when the group has no in-launch exchange."""
    lib = L.load()
    st = _raw_stream(x.device.index)
    plan = _xrank_plan('aciq', x, N, C, HW, group, st,
                       lambda: _aciq_ws_bytes(x, N, C, HW) + (L.NSTAT + L.NQP + L.NDIAG) * C * 4 + L.NMOM * C * 8)
    if plan is None:
        return None
    xr, _, ws, gws = plan
    y = torch.empty_like(x)
    tabs = torch.empty((L.NSTAT + L.NMT, C), dtype=torch.float32, device=x.device)
    stats, mt = tabs[:L.NSTAT], tabs[L.NSTAT:]
    mom = torch.empty((L.NMOM, C), dtype=torch.float64, device=x.device)
    hist = torch.empty(L.mt_hist_words(C), dtype=torch.int64, device=x.device) if want_entropy else None     # zeroed by the call
    ctx = xr.ctx(st)
    rc = lib.cnnq_pc_midtread_fused_xrank(x.data_ptr(), y.data_ptr(), N, C, HW, float(target), int(bool(sym)), tabs_mt.data_ptr(),
                                          tabs_mt.shape[1], ws.data_ptr(), gws, GROUP_WS_BYTES if gws is not None else 0, stats.data_ptr(),
                                          mom.data_ptr(), mt.data_ptr(), _ptr(hist), ctypes.byref(ctx), int(flags), st)
    if rc:
        L.check(rc, 'cnnq_pc_midtread_fused_xrank')
    entropy = None
    if want_entropy:
        D.all_reduce_sum_(hist, group)
        ent = torch.empty(1, dtype=torch.float32, device=x.device)
        # (the global batch's element count from the merged moment record: shards may differ by a sample)
        L.check(lib.cnnq_midtread_entropy_count(_ptr(hist), _ptr(mt), C, mom[L.MOM_COUNT].data_ptr(), _ptr(ent), st), 'cnnq_midtread_entropy')
        entropy = ent[0]
    res = [y, entropy]
    if want_parts:
        res.append(dict(stats=stats, mt=mt, hist=hist, mom=mom))
    return tuple(res)


def pc_stats_single(x, N, C, HW, need_b=False, need_kurt=False, need_relu=False, flags=8):
    """The statistics table and the merged moment record from ONE launch that reads x once (cnnq_pc_stats_single), or None
    when the shape has no flat-tile plan.  flags: bit 3 (default here) also takes channels of more than 256 tiles, which
    ops.pc_stats leaves to the chain; bit 0 forces the recompute path (tests)."""
    lib = L.load()
    x = _dev_f32(x, 'x')
    st = _raw_stream(x.device.index)
    gws = _group_workspace(x, st)
    if gws is None:
        return None
    stats = torch.empty((L.NSTAT, C), dtype=torch.float32, device=x.device)
    mom = torch.empty((L.NMOM, C), dtype=torch.float64, device=x.device)
    rc = lib.cnnq_pc_stats_single(_ptr(x), N, C, HW, int(bool(need_b)), int(bool(need_kurt)), int(bool(need_relu)), gws,
                                  GROUP_WS_BYTES, _ptr(mom), _ptr(stats), int(flags), st)
    if rc == L.ENOTSUP:
        return None
    L.check(rc, 'cnnq_pc_stats_single')
    return stats, mom


# ------------------------------------------------------------------------------------- parameters
CLIP_CODES = {'no': 0, 'laplace': 1, 'gaus': 2}


def _params_cfg(num_bits, positive, clip, bit_alloc, prior_is_b, target, round_mode, direct_range):
    cfg = L.ParamsCfg()
    cfg.num_bits = int(num_bits)
    cfg.positive = int(bool(positive))
    if clip in CLIP_CODES:
        if clip != 'no' and int(num_bits) > 8:
            # iq.py:14-41: alpha_laplace / alpha_gaus have keys 0..8 only (the reference raises KeyError)
            raise L.CnnqError('%s clipping is tabulated for at most 8 bits, got %d' % (clip, int(num_bits)))
        cfg.clip, cfg.pstd = CLIP_CODES[clip], 0.
    elif 'std' in clip:
        cfg.clip, cfg.pstd = 3, float(clip.replace('std', ''))
    elif clip == 'mix':
        # iq.py:310-323 picks per channel between three clipping values from error columns of a statistics file: that is
        # act_qdq_mix (three parameter tables merged per channel), not one configuration of this kernel
        raise L.CnnqError("clipping 'mix' goes through ops.act_qdq_mix (it needs the mse_* columns of a statistics file)")
    else:
        raise L.CnnqError('unsupported clipping %r' % (clip,))
    cfg.bit_alloc = int(bool(bit_alloc))
    cfg.prior_is_b = int(bool(prior_is_b))
    cfg.target = float(num_bits if target is None else target)
    cfg.round_mode = int(bool(round_mode))
    cfg.direct_range = int(bool(direct_range))
    return cfg


def pc_params(stats, num_bits, positive=False, clip='no', bit_alloc=False, prior_is_b=False, target=None,
              round_mode=True, direct_range=False):
    """stats [NSTAT, C] -> (qp [NQP, C], diag [NDIAG, C]); see cnnq_pc_params."""
    lib = L.load()
    C = stats.shape[1]
    cfg = _params_cfg(num_bits, positive, clip, bit_alloc, prior_is_b, target, round_mode, direct_range)
    qp = torch.empty((L.NQP, C), dtype=torch.float32, device=stats.device)
    diag = torch.empty((L.NDIAG, C), dtype=torch.float32, device=stats.device)
    L.check(lib.cnnq_pc_params(_ptr(stats), C, ctypes.byref(cfg), _ptr(qp), _ptr(diag), _stream(stats)),
            'cnnq_pc_params')
    return qp, diag


def act_qdq_mix(x, num_bits, stats, mse, positive=False, bit_alloc=False, prior_is_b=False, target=None, round_mode=True,
                whole_tensor=False, want_codes=False, want_entropy=False, out=None):
    """clip_type == 'mix' of the `-sm use` route (iq.py:310-323 + 327-359): per channel the Gaussian clipping value where
    mse_gaus < mse_laplace, else the Laplace one, and the min/max half range (max - min) / 2 where mse_lowp < mse_gaus;
    comparisons with NaN are False, so a file whose error columns are NaN - all the reference's own collection writes -
    gives plain Laplace clipping.  stats [NSTAT, C]: the file's mean_{min, max, mean, b, std} rows; mse [3, C]: rows
    laplace, gaus, lowp.  Everything downstream of the clipping value is per channel and elementwise, so the three
    candidates' parameter tables are merged per channel: two cnnq_pc_params launches, the min/max candidate in a few
    [C]-sized fp32 torch ops that repeat that kernel's arithmetic (iq.py:284-300, 351, 443, 559-572), one fused Q/DQ."""
    x = _dev_f32(x, 'x')
    N, C, HW = (1, 1, x.numel()) if whole_tensor else geometry(x)
    stats = stats.to(device=x.device, dtype=torch.float32)
    mse = torch.as_tensor(mse, dtype=torch.float32, device=x.device).view(3, C)
    kw = dict(positive=positive, bit_alloc=bit_alloc, prior_is_b=prior_is_b, target=target, round_mode=round_mode,
              direct_range=whole_tensor)
    qp_l, dg_l = pc_params(stats, num_bits, clip='laplace', **kw)
    qp_g, _ = pc_params(stats, num_bits, clip='gaus', **kw)
    mn, mx, mean = stats[L.STAT_MIN], stats[L.STAT_MAX], stats[L.STAT_MEAN]
    alpha = (mx - mn) / 2
    if positive:
        rng, off = torch.clamp_min(mean, 0.) + alpha, torch.zeros_like(alpha)
    else:
        rng, off = 2 * alpha, torch.maximum(mn, mean - alpha)
    delta = rng if whole_tensor else (off + rng) - off
    if bit_alloc and num_bits <= 4:
        qmax = torch.exp2(dg_l[L.DIAG_BITS]) - 1.
        scale = torch.where(qmax > 0, delta / qmax, torch.zeros_like(delta))
    else:
        qmax = torch.full_like(delta, float(2 ** int(num_bits) - 1))
        scale = delta / qmax
    scale = torch.where(scale < 1e-8, torch.full_like(scale, 1e-8), scale)
    zp = torch.round(0. - off / scale)
    qp_p = torch.stack([scale, zp, qmax])
    pick_g = (mse[1] < mse[0]).view(1, C)
    pick_p = (mse[2] < mse[1]).view(1, C)
    qp = torch.where(pick_p, qp_p, torch.where(pick_g, qp_g, qp_l)).contiguous()
    hist = torch.zeros(256, dtype=torch.int64, device=x.device) if want_entropy else None
    res = pc_qdq(x, N, C, HW, qp, want_codes=want_codes, out=out, hist=hist)
    if want_entropy:
        res = (res + (entropy_from_hist(hist),)) if want_codes else (res, entropy_from_hist(hist))
    return res


# ------------------------------------------------------------------------------------- Q/DQ
def pc_qdq(x, N, C, HW, qp, want_codes=False, out=None, hist=None, reverse=False):
    """y = dequant(quant(x)) with per-channel parameters; optionally the uint8 codes; `hist`
    (optional zeroed int64[256] tensor) receives the code histogram; reverse: descending addresses."""
    lib = L.load()
    x = _dev_f32(x, 'x')
    y = _out_like(x, out)
    codes = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if want_codes else None
    L.check(lib.cnnq_pc_qdq(_ptr(x), _ptr(y), N, C, HW, _ptr(qp), _ptr(codes), _ptr(hist), int(bool(reverse)),
                            _stream(x)), 'cnnq_pc_qdq')
    return (y, codes) if want_codes else y


_GROUP_WS = {}
GROUP_WS_BYTES = 18 << 20


_GROUP_POOL = {}
GROUP_POOL_SIZE = 4


def _group_workspace(x, st=None):
    """The exchange workspace of the group single launch (cnnq_pc_minmax_qdq_group): fine-grained device memory
    from cnnq_group_ws_alloc, zeroed ONCE, one per (device, stream) - launches on one stream are ordered, so they
    share it; the kernel re-arms its counters.  Allocation synchronises the device, which a stream capture does
    not survive: the first call on a device allocates a small pool, later streams (a capture stream included)
    take from it; with the pool empty under capture there is no workspace (None: the caller takes the chain).
    Returns the raw device pointer (a ctypes.c_void_p)."""
    dev = x.device.index
    key = (dev, _raw_stream(dev) if st is None else st)
    ws = _GROUP_WS.get(key)
    if ws is not None:
        return ws
    pool = _GROUP_POOL.setdefault(dev, [])
    if not pool:
        if torch.cuda.is_current_stream_capturing():
            return None
        lib = L.load()
        for _ in range(GROUP_POOL_SIZE):
            w = ctypes.c_void_p()
            L.check(lib.cnnq_group_ws_alloc(GROUP_WS_BYTES, ctypes.byref(w)), 'cnnq_group_ws_alloc')
            pool.append(w)
    ws = _GROUP_WS[key] = pool.pop()
    return ws


GROUP_WAIT_EXPIRED, GROUP_TEST_HOOK = 1, 2


def group_status(x, clear=None):
    """Status word of this stream's group workspace (synchronises).  Bit 0 (GROUP_WAIT_EXPIRED): a bounded wait of the
    in-launch exchange ran out and its workgroup recomputed the extrema from x - results are unaffected, but the
    launch was slow (a group's members were not co-resident: another kernel held the CUs).  Bit 1 (GROUP_TEST_HOOK):
    the recompute path was forced by the test flag.  The word is sticky on the device, and while bit 0 is up every later
    wait is bounded by 0.5 ms instead of 20 ms (csrc/cnnq_group.hip.h): reading it ACKNOWLEDGES it - a non-zero word is
    zeroed after the read unless clear=False - so one transient expiry (a profiler attaching, a co-tenant's kernel) does
    not leave the short bound in force for the rest of the process.  clear=True zeroes unconditionally."""
    ws = _GROUP_WS.get((x.device.index, _raw_stream(x.device.index)))
    if ws is None:
        return 0
    v = ctypes.c_uint32()
    lib = L.load()
    L.check(lib.cnnq_group_ws_status(ws, ctypes.byref(v)), 'cnnq_group_ws_status')
    if clear or (clear is None and v.value != 0):
        L.check(lib.cnnq_group_ws_status_clear(ws), 'cnnq_group_ws_status_clear')
    return int(v.value)


def minmax_qdq_group(x, N, C, HW, num_bits, positive=False, out=None, want_parts=False, flags=0):
    """Config 2 in ONE launch and ONE read of x for tensors whose channels span several workgroups
    (cnnq_pc_minmax_qdq_group).  Returns None when the shape is not supported (the caller takes the chain)."""
    lib = L.load()
    x = _dev_f32(x, 'x')
    nbytes = lib.cnnq_pc_group_workspace(N, C, HW)
    if nbytes == 0 or nbytes > GROUP_WS_BYTES:
        return None
    gws = _group_workspace(x)
    if gws is None:
        return None
    y = _out_like(x, out)
    qp = torch.empty((L.NQP + 2, C), dtype=torch.float32, device=x.device)
    rc = lib.cnnq_pc_minmax_qdq_group(_ptr(x), _ptr(y), N, C, HW, int(num_bits), int(bool(positive)),
                                      gws, _ptr(qp), _ptr(qp[L.NQP:]) if want_parts else None,
                                      int(flags), _stream(x))
    if rc == L.ENOTSUP:
        return None
    L.check(rc, 'cnnq_pc_minmax_qdq_group')
    if want_parts:
        stats = torch.zeros((L.NSTAT, C), dtype=torch.float32, device=x.device)
        stats[L.STAT_MIN] = qp[L.NQP]
        stats[L.STAT_MAX] = qp[L.NQP + 1]
        return y, dict(stats=stats, qp=qp[:L.NQP], diag=None)
    return y


def minmax_qdq_resident(x, N, C, HW, num_bits, positive=False, out=None, want_parts=False):
    """Config 2 in ONE launch and ONE read of x (cnnq_pc_minmax_qdq_resident): the bits of the three-launch chain
    at 8 instead of 12 bytes per element.  Returns None when the shape has no resident kernel (a channel's batch
    population does not fit a workgroup's registers, unaligned pointers, H*W % 4 != 0 without a straddling
    layout) - the caller then takes the chain."""
    lib = L.load()
    x = _dev_f32(x, 'x')
    y = _out_like(x, out)
    qp = torch.empty((L.NQP + 2, C), dtype=torch.float32, device=x.device)     # parameters + {min, max} rows
    rc = lib.cnnq_pc_minmax_qdq_resident(_ptr(x), _ptr(y), N, C, HW, int(num_bits), int(bool(positive)), _ptr(qp),
                                         _ptr(qp[L.NQP:]) if want_parts else None, _stream(x))
    if rc == L.ENOTSUP:
        return None
    L.check(rc, 'cnnq_pc_minmax_qdq_resident')
    if want_parts:
        stats = torch.zeros((L.NSTAT, C), dtype=torch.float32, device=x.device)
        stats[L.STAT_MIN] = qp[L.NQP]
        stats[L.STAT_MAX] = qp[L.NQP + 1]
        return y, dict(stats=stats, qp=qp[:L.NQP], diag=None)
    return y


_HIST_REP = {}


def _hist_replicas(x, st):
    """The replica histogram of the single-launch kernels (cnnq_hist_replica_bytes), one per (device, stream), zeroed
    ONCE: cnnq_entropy_replicas leaves it zero."""
    key = (x.device.index, st)
    t = _HIST_REP.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            # first use under a stream capture: the zero fill would only be RECORDED and the block would belong to the
            # graph's pool - an eager call on this stream later would count into a table that was never zeroed.  The
            # caller takes the chain for this call instead (its histogram is a per-call buffer).
            return None
        t = _HIST_REP[key] = torch.zeros(L.load().cnnq_hist_replica_bytes() // 8, dtype=torch.int64, device=x.device)
    return t


_ENT_BATCH = None
_ENT_TABLES = {}        # per (device, stream): ENT_BATCH_CAP sets of replica tables, zeroed once (the batch kernel leaves them zero)
ENT_BATCH_CAP = 64


class entropy_batch:
    """`with ops.entropy_batch(): ...` - the entropies of the codes (want_entropy=True: -me, iq.py:445,179,217) of every tensor
    quantized inside the block are computed by ONE launch at its end instead of a dependent one-workgroup launch behind every
    tensor (8-25 us each: 0.76 ms of the 9.5 ms ResNet-50 b512 step with -me in round 5).  Nothing consumes an entropy
    mid-forward - the reference logs it - so the 0-dim tensors the calls hand out are only filled when the block ends (or at
    flush()); reading one earlier gives whatever the buffer held.  Each tensor counts into its own set of replica tables (128 KB;
    64 sets per stream).  One stream per block; not for use under a stream capture on its first use (the tables are zero-filled
    once) - the calls then take the per-tensor launch."""

    def __init__(self):
        self.n, self.out, self.tables, self.st, self.mt, self.pending = 0, None, None, None, [], []

    def after(self, fn):
        """run fn() once the entropies of the block are there (IntQuantizer's logger calls: float(entropy) is a host read)"""
        self.pending.append(fn)

    def __enter__(self):
        global _ENT_BATCH
        self._outer, _ENT_BATCH = _ENT_BATCH, self
        return self

    def __exit__(self, *exc):
        global _ENT_BATCH
        _ENT_BATCH = self._outer
        if exc[0] is None:
            self.flush()
        return False

    def slot(self, x, st):
        """(replica tables of this tensor, the 0-dim result) or None when no table set can be had (first use under a capture)"""
        if self.st is None:
            self.st = st
            key = (x.device.index, st)
            self.tables = _ENT_TABLES.get(key)
            if self.tables is None:
                if torch.cuda.is_current_stream_capturing():
                    self.st = None
                    return None
                words = L.load().cnnq_hist_replica_bytes() // 8
                self.tables = _ENT_TABLES[key] = torch.zeros((ENT_BATCH_CAP, words), dtype=torch.int64, device=x.device)
        elif st != self.st:
            raise L.CnnqError('entropy_batch: one stream per block')
        if self.n == ENT_BATCH_CAP:
            self.flush()
        if self.out is None:
            self.out = torch.empty(ENT_BATCH_CAP, dtype=torch.float32, device=x.device)
        i = self.n
        self.n += 1
        return self.tables[i], self.out[i]

    def add_midtread(self, hist, mt, C, total, device):
        """the mid-tread path's histogram of one tensor (cnnq_midtread_entropy's arguments); returns the 0-dim result"""
        if self.st is None:
            self.st = _raw_stream(device.index)
        if len(self.mt) == 16:
            self._flush_midtread()
        if not self.mt:
            self.mt_out = torch.empty(16, dtype=torch.float32, device=device)
        self.mt.append((hist, mt, int(C), int(total)))      # (the references keep the tables alive until the launch)
        return self.mt_out[len(self.mt) - 1]

    def _flush_midtread(self):
        n = len(self.mt)
        if n:
            hp = (ctypes.c_void_p * n)(*[h.data_ptr() for h, _, _, _ in self.mt])
            mp = (ctypes.c_void_p * n)(*[m.data_ptr() for _, m, _, _ in self.mt])
            cs = (ctypes.c_int64 * n)(*[c for _, _, c, _ in self.mt])
            ts = (ctypes.c_int64 * n)(*[t for _, _, _, t in self.mt])
            L.check(L.load().cnnq_midtread_entropy_batch(n, hp, mp, cs, ts, self.mt_out.data_ptr(), self.st), 'cnnq_midtread_entropy_batch')
            self.mt = []

    def flush(self):
        if self.n:
            L.check(L.load().cnnq_entropy_replicas_batch(self.tables.data_ptr(), self.n, self.out.data_ptr(), self.st),
                    'cnnq_entropy_replicas_batch')
            self.n, self.out = 0, None          # (the views handed out keep the old result buffer alive)
        self._flush_midtread()
        pending, self.pending = self.pending, []
        for fn in pending:
            fn()


def _entropy_slot(x, st):
    """Where a want_entropy launch counts: (tables, result or None).  Inside an entropy_batch block its next table set and the
    result it will fill at the end; else the stream's shared replica tables (None: no result yet - the caller launches
    cnnq_entropy_replicas behind its tensor)."""
    if _ENT_BATCH is not None:
        s = _ENT_BATCH.slot(x, st)
        if s is not None:
            return s
    return _hist_replicas(x, st), None


def minmax_qdq_single(x, N, C, HW, num_bits, positive=False, want_codes=False, want_entropy=False, out=None,
                      want_parts=False):
    """Config 2 in ONE launch also when the codes and / or the entropy of the codes are wanted
    (cnnq_pc_minmax_qdq_single; the entropy is one more tiny launch).  Returns None when the shape has no
    single-launch kernel or the codes do not fit a byte (the caller takes the chain)."""
    if num_bits > 8 and (want_codes or want_entropy):
        return None
    lib = L.load()
    x = _dev_f32(x, 'x')
    st = _raw_stream(x.device.index)
    gws = _group_workspace(x, st)
    y = _out_like(x, out)
    qp = torch.empty((L.NQP + 2, C), dtype=torch.float32, device=x.device)
    codes = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if want_codes else None
    hist, ent_batched = _entropy_slot(x, st) if want_entropy else (None, None)
    if want_entropy and hist is None:
        return None
    rc = lib.cnnq_pc_minmax_qdq_single(_ptr(x), _ptr(y), N, C, HW, int(num_bits), int(bool(positive)), gws,
                                       GROUP_WS_BYTES if gws is not None else 0, _ptr(qp), _ptr(qp[L.NQP:]), _ptr(codes),
                                       _ptr(hist), None, st)
    if rc == L.ENOTSUP:
        return None
    L.check(rc, 'cnnq_pc_minmax_qdq_single')
    res = [y]
    if want_codes:
        res.append(codes)
    if want_entropy:
        if ent_batched is not None:
            res.append(ent_batched)              # filled by the one launch at the end of the entropy_batch block
        else:
            ent = torch.empty(1, dtype=torch.float32, device=x.device)
            L.check(lib.cnnq_entropy_replicas(_ptr(hist), _ptr(ent), st), 'cnnq_entropy_replicas')
            res.append(ent[0])
    if want_parts:
        stats = torch.zeros((L.NSTAT, C), dtype=torch.float32, device=x.device)
        stats[L.STAT_MIN] = qp[L.NQP]
        stats[L.STAT_MAX] = qp[L.NQP + 1]
        res.append(dict(stats=stats, qp=qp[:L.NQP], diag=None))
    return res[0] if len(res) == 1 else tuple(res)


def aciq_qdq_single(x, N, C, HW, num_bits, positive=False, bit_alloc=False, target=None, round_mode=True, want_codes=False,
                    want_entropy=False, want_parts=False, out=None, flags=0):
    """Config 3 (Laplace clipping, dynamic statistics, optional bit allocation on the 'gaus' prior) with pass B, the
    parameters and the Q/DQ in ONE launch and one read of x (cnnq_pc_aciq_qdq_single: pass A, merge, bit allocation, the
    single launch).  Returns y [, codes] [, entropy] [, parts], or None when the shape has no single-launch plan (the
    caller takes the chain)."""
    lib = L.load()
    x = _dev_f32(x, 'x')
    st = _raw_stream(x.device.index)
    gws = _group_workspace(x, st)
    if gws is None:
        return None
    hist, ent_batched = _entropy_slot(x, st) if want_entropy else (None, None)
    if want_entropy and hist is None:
        return None
    cfg = _params_cfg(num_bits, positive, 'laplace', bit_alloc, False, target, round_mode, False)
    al = int(x.data_ptr() % 16 == 0)
    key = ('aciq', N, C, HW, al)
    nbytes = _WS_BYTES.get(key)
    if nbytes is None:
        nbytes = _WS_BYTES[key] = (lib.cnnq_pc_aciq_workspace(N, C, HW, al) + 15) // 16 * 16
    ws = _scratch(x, 'aciq', nbytes + (L.NQP + L.NDIAG) * C * 4, st)
    y = _out_like(x, out)
    tabs = torch.empty((L.NSTAT + L.NQP + L.NDIAG, C), dtype=torch.float32, device=x.device)
    stats, qp, diag = tabs[:L.NSTAT], tabs[L.NSTAT:L.NSTAT + L.NQP], tabs[L.NSTAT + L.NQP:]
    codes = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if want_codes else None
    rc = lib.cnnq_pc_aciq_qdq_single(_ptr(x), _ptr(y), N, C, HW, ctypes.byref(cfg), ws.data_ptr(), gws, GROUP_WS_BYTES, _ptr(stats),
                                     _ptr(qp), _ptr(diag), _ptr(codes), _ptr(hist), int(flags), st)
    if rc == L.ENOTSUP:
        return None
    L.check(rc, 'cnnq_pc_aciq_qdq_single')
    res = [y]
    if want_codes:
        res.append(codes)
    if want_entropy:
        if ent_batched is not None:
            res.append(ent_batched)
        else:
            ent = torch.empty(1, dtype=torch.float32, device=x.device)
            L.check(lib.cnnq_entropy_replicas(_ptr(hist), _ptr(ent), st), 'cnnq_entropy_replicas')
            res.append(ent[0])
    if want_parts:
        res.append(dict(stats=stats, qp=qp, diag=diag))
    return res[0] if len(res) == 1 else tuple(res)


def minmax_quantize_pack4(x, num_bits=4, positive=False, out=None):
    """Dynamic per-channel min/max quantization of x [N, C, H, W] straight to the STORED format: two 4-bit codes per
    byte instead of the dequantized floats, in one launch and one read of x (4.5 bytes per element).
    Returns (packed uint8 [numel / 2], qp [NQP, C]); dequantize_pack4(packed, x.shape, qp) gives back exactly what
    act_qdq_per_channel(x, num_bits) returns.  None when the shape has no single-launch kernel."""
    lib = L.load()
    x = _dev_f32(x, 'x')
    if num_bits > 4 or x.numel() % 2:
        raise L.CnnqError('packed 4-bit storage needs num_bits <= 4 and an even number of elements')
    N, C, HW = geometry(x)
    st = _raw_stream(x.device.index)
    gws = _group_workspace(x, st)
    packed = torch.empty(x.numel() // 2, dtype=torch.uint8, device=x.device) if out is None else out
    if out is not None and not (out.is_cuda and out.dtype == torch.uint8 and out.is_contiguous() and out.numel() >= x.numel() // 2):
        raise L.CnnqError('out must be a contiguous uint8 device buffer of numel / 2 bytes')
    qp = torch.empty((L.NQP, C), dtype=torch.float32, device=x.device)
    rc = lib.cnnq_pc_minmax_qdq_single(_ptr(x), None, N, C, HW, int(num_bits), int(bool(positive)), gws,
                                       GROUP_WS_BYTES if gws is not None else 0, _ptr(qp), None, None, None, _ptr(packed), st)
    if rc == L.ENOTSUP:
        return None
    L.check(rc, 'cnnq_pc_minmax_qdq_single')
    return packed, qp


def _cfg2_workspace_bytes(lib, N, C, HW):
    wplan = _WS_BYTES.get((N, C, HW))
    nbytes = wplan[0] if wplan is not None else lib.cnnq_pc_minmax_qdq_workspace(N, C, HW)
    if nbytes == 0:
        L.check(min(lib.cnnq_pc_groups(N, C, HW, 1), -1), 'cnnq_pc_groups(%d,%d,%d)' % (N, C, HW))
    return nbytes


def minmax_qdq_fused(x, N, C, HW, num_bits, positive=False, want_codes=False, want_entropy=False, out=None,
                     want_parts=False, group=None, _checked=False, chain=False, _xrank=None):
    """Config 2 (iq.py:409-451, 557-603): dynamic per-channel min/max -> scale / zero point -> Q/DQ; no host sync.
    Three routes, one function each:

    * one GPU (_minmax_qdq_local): a single launch that reads x once where the shape has one, else the three-launch
      chain (k_minmax -> k_minmax_params -> k_qdq);
    * the batch sharded over the ranks of `group`, x this rank's shard (world size > 1, or CNNQ_FORCE_EXCHANGE=1 on a
      1-rank group): the collective route (_minmax_qdq_collective, default: statistics launch -> all_gather of the local
      extrema [2, C] -> parameters + Q/DQ launch), or - opt-in, D.xrank_mode - the exchange inside the single launch
      (_minmax_qdq_xrank).  Extrema are exact, so every rank gets the bits of one GPU holding the whole batch.

    chain=True forces the three-launch chain where a single-launch kernel would otherwise run: the reference form the
    single-launch kernels are tested against.  _xrank: an XRankExchange to use (its verify()), False: never.  group=False:
    replicated data, the one-GPU route."""
    if not _checked:
        x = _dev_f32(x, 'x')
    # group=False: replicated data (weights) - never exchanged, whatever process group the job runs in.  (Until round 6 this
    # arrived here as None = the default group: every rank's identical weights went through the exchange - the same bits, one
    # needless collective per layer, and a wait that expired there left NaN WEIGHTS behind, outside any forward a checkpoint redoes.)
    world = 1 if group is False else D.world_size(group)
    if group is False or not (world > 1 or D.forced_exchange()):
        return _minmax_qdq_local(x, N, C, HW, num_bits, positive, want_codes, want_entropy, out, want_parts, chain)
    resident = _RESIDENT and not chain
    if _xrank is not False and (_xrank is not None or _XRANK_ON) and not ((want_codes or want_entropy) and num_bits > 8):
        res = _minmax_qdq_xrank(x, N, C, HW, num_bits, positive, want_codes, want_entropy, out, want_parts, group, resident, _xrank)
        if res is not None:
            return res
    return _minmax_qdq_collective(x, N, C, HW, num_bits, positive, want_codes, want_entropy, out, want_parts, group, world, resident)


def _minmax_qdq_local(x, N, C, HW, num_bits, positive, want_codes, want_entropy, out, want_parts, chain):
    """Config 2 on one GPU."""
    lib = L.load()
    resident = _RESIDENT and not chain
    if not (want_codes or want_entropy or want_parts):
        # the hot call: one C entry point, one cached workspace, no torch allocation besides the result
        key = (N, C, HW)
        plan = _WS_BYTES.get(key)
        if plan is None:
            nbytes = lib.cnnq_pc_minmax_qdq_workspace(N, C, HW)
            if nbytes == 0:
                L.check(min(lib.cnnq_pc_groups(N, C, HW, 1), -1), 'cnnq_pc_groups(%d,%d,%d)' % (N, C, HW))
            d = (ctypes.c_int32 * 8)()
            # the group workspace is needed when there is no whole-channel kernel, or one with too few workgroups
            # to fill the chip (RES_MIN_WGS in csrc/cnnq_plan.hip.h: the C side routes, this only decides whether
            # to hand it the workspace)
            wants_group = ((lib.cnnq_pc_resident_describe(N, C, HW, d) != 0 or d[6] < 192)
                           and 0 < lib.cnnq_pc_group_workspace(N, C, HW) <= GROUP_WS_BYTES)
            plan = _WS_BYTES[key] = (nbytes, wants_group)
        nbytes, wants_group = plan
        y = _out_like(x, out)
        st = _raw_stream(x.device.index)          # looked up once: workspace keys and the launch stream
        gws = _group_workspace(x, st) if (resident and wants_group) else None
        rc = lib.cnnq_pc_minmax_qdq_auto(x.data_ptr(), y.data_ptr(), N, C, HW, int(num_bits), 1 if positive else 0,
                                         _scratch(x, 'cfg2', nbytes, st).data_ptr(), gws,
                                         GROUP_WS_BYTES if gws is not None else 0, 1 if resident else 0, st)
        if rc:
            L.check(rc, 'cnnq_pc_minmax_qdq_auto')
        return y
    if resident and not want_codes and not want_entropy:
        res = minmax_qdq_resident(x, N, C, HW, num_bits, positive, out=out, want_parts=want_parts)
        if res is None:
            res = minmax_qdq_group(x, N, C, HW, num_bits, positive, out=out, want_parts=want_parts)
        if res is not None:
            return res
    if resident and _SINGLE_CODES:
        # codes / entropy out of the single launch too (round 3): no chain, no memset
        res = minmax_qdq_single(x, N, C, HW, num_bits, positive, want_codes, want_entropy, out=out, want_parts=want_parts)
        if res is not None:
            return res
    y = _out_like(x, out)
    G = max(lib.cnnq_pc_groups(N, C, HW, 1), lib.cnnq_pc_groups(N, C, HW, 0))
    if G <= 0:
        L.check(G, 'cnnq_pc_groups(%d,%d,%d)' % (N, C, HW))
    pmm = torch.empty((G, 2, C), dtype=torch.float32, device=x.device)
    qp = torch.empty((L.NQP, C), dtype=torch.float32, device=x.device)
    codes = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if want_codes else None
    hist = torch.zeros(256, dtype=torch.int64, device=x.device) if want_entropy else None
    L.check(lib.cnnq_pc_minmax_qdq(_ptr(x), _ptr(y), N, C, HW, int(num_bits), int(bool(positive)), _ptr(pmm),
                                   _ptr(qp), _ptr(codes), _ptr(hist), _stream(x)), 'cnnq_pc_minmax_qdq')
    res = [y]
    if want_codes:
        res.append(codes)
    if want_entropy:
        res.append(entropy_from_hist(hist))
    if want_parts:
        g_used = lib.cnnq_pc_groups(N, C, HW, int(x.data_ptr() % 16 == 0))
        stats = torch.zeros((L.NSTAT, C), dtype=torch.float32, device=x.device)
        stats[L.STAT_MIN] = pmm[:g_used, 0].min(dim=0)[0]
        stats[L.STAT_MAX] = pmm[:g_used, 1].max(dim=0)[0]
        res.append(dict(stats=stats, qp=qp, diag=None))
    return res[0] if len(res) == 1 else tuple(res)


def _minmax_qdq_xrank(x, N, C, HW, num_bits, positive, want_codes, want_entropy, out, want_parts, group, resident, _xrank):
    """Config 2 of a batch shard with the cross-rank exchange INSIDE the single launch (csrc/cnnq_xrank.hip.h; opt-in,
    verified against the collective at first use): x is read once; every rank takes this route or none does.  Also with
    the codes / the entropy of the codes / the parameters wanted (the ranks' code counts are summed afterwards).  None: this
    group has no (verified) in-launch exchange - the caller takes the collective."""
    lib = L.load()
    st = _raw_stream(x.device.index)
    if _xrank is None and resident and not (want_codes or want_entropy or want_parts):
        # the sharded hot call: everything that does not change from call to call is looked up once (the exchange of the
        # group, the workspaces); D.disable_xrank / release_plans() drop the plans
        key = ('xr', id(group), x.device.index, st, N, C, HW)
        plan = _XPLAN.get(key)
        if plan is None and not torch.cuda.is_current_stream_capturing():    # (a capture-time scratch buffer is never cached)
            xr = D.xrank_exchange(group)
            plan = False
            if xr is not None and xr.fits(C):
                plan = (xr, group, _scratch(x, 'cfg2', _cfg2_workspace_bytes(lib, N, C, HW), st), _group_workspace(x, st))
            _XPLAN[key] = plan
        if plan:
            y = _out_like(x, out)
            plan[0].minmax_qdq(x, y, N, C, HW, num_bits, positive, plan[2].data_ptr(), plan[3], GROUP_WS_BYTES, st)
            return y
        xr = None if plan is False else D.xrank_exchange(group)
    else:
        xr = _xrank if _xrank is not None else D.xrank_exchange(group)
    hist_rep = _hist_replicas(x, st) if (want_entropy and xr is not None) else None
    if xr is None or not xr.fits(C) or (want_entropy and hist_rep is None):
        return None
    y = _out_like(x, out)
    gws = _group_workspace(x, st) if resident else None
    ws = _scratch(x, 'cfg2', _cfg2_workspace_bytes(lib, N, C, HW), st)
    codes = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if want_codes else None
    xr.minmax_qdq(x, y, N, C, HW, num_bits, positive, ws.data_ptr(), gws,
                  GROUP_WS_BYTES if gws is not None else 0, st, codes=codes, hist_rep=hist_rep)
    if not (want_codes or want_entropy or want_parts):
        return y
    res = [y]
    if want_codes:
        res.append(codes)
    if want_entropy:
        hist = torch.zeros(256, dtype=torch.int64, device=x.device)
        L.check(lib.cnnq_hist_replicas_fold(_ptr(hist_rep), _ptr(hist), st), 'cnnq_hist_replicas_fold')
        D.all_reduce_sum_(hist, group)                       # the global batch's code counts
        res.append(entropy_from_hist(hist))
    if want_parts:
        tab = ws[:4 * (L.NQP + 2) * C].view(torch.float32).view(L.NQP + 2, C).clone()    # qp rows, then the global {min, max}
        stats = torch.zeros((L.NSTAT, C), dtype=torch.float32, device=x.device)
        stats[L.STAT_MIN] = tab[L.NQP]
        stats[L.STAT_MAX] = tab[L.NQP + 1]
        res.append(dict(stats=stats, qp=tab[:L.NQP], diag=None))
    return res[0] if len(res) == 1 else tuple(res)


def _minmax_qdq_collective(x, N, C, HW, num_bits, positive, want_codes, want_entropy, out, want_parts, group, world, resident):
    """Config 2 of a batch shard around a collective: local extrema (one launch) -> all_gather of [2, C] (ncclAllGather
    enqueued directly on the compute stream, rccl.py; torch.distributed on gloo rigs) -> parameters from the W gathered
    records + Q/DQ (one launch).  x is read twice (12 bytes per element); host-side skew between the ranks is harmless."""
    lib = L.load()
    if not (want_codes or want_entropy or want_parts):
        # the hot call: two C calls and the collective, all in cached workspaces
        st = _raw_stream(x.device.index)
        key = (id(group), x.device.index, st, N, C, HW, world, _DIRECT_RCCL)
        plan = _XPLAN.get(key)
        if plan is None:
            # everything that does not change from call to call: workspace slices, the gathered buffer, the direct
            # RCCL communicator (created collectively at first use).  The plan keeps `group` alive, so its id stays its
            # own, and its scratch buffers too (a later, larger tensor may make _scratch hand out new ones).
            ws = _scratch(x, 'cfg2', _cfg2_workspace_bytes(lib, N, C, HW), st)
            gbuf = _scratch(x, 'gath', 8 * C * world, st)
            dc = None
            if x.is_cuda:
                from . import rccl
                dc = rccl.direct_comm(group)
            plan = _XPLAN[key] = dict(
                group=group, ws=ws, gbuf=gbuf, base=ws.data_ptr(), dc=dc,
                local=ws[12 * C:20 * C].view(torch.float32).view(2, C),            # the mm[2][C] slot of the workspace
                gathered=gbuf[:8 * C * world].view(torch.float32).view(world, 2, C))
        base = plan['base']
        y = _out_like(x, out)
        gws = _group_workspace(x, st) if resident else None      # one launch for the local extrema when the plan allows
        rc = lib.cnnq_pc_minmax_local_auto(x.data_ptr(), N, C, HW, base + 20 * C, gws,
                                           GROUP_WS_BYTES if gws is not None else 0, base + 12 * C, st)
        if rc:
            L.check(rc, 'cnnq_pc_minmax_local_auto')
        dc = plan['dc']
        gathered = plan['gathered']
        if dc is not None:
            dc.all_gather_raw(base + 12 * C, gathered.data_ptr(), 8 * C, st)
        else:
            gathered = D.all_gather_records(plan['local'], group, out=gathered)
        rc = lib.cnnq_pc_gathered_qdq(x.data_ptr(), y.data_ptr(), N, C, HW, gathered.data_ptr(), world, int(num_bits),
                                      1 if positive else 0, base, st)
        if rc:
            L.check(rc, 'cnnq_pc_gathered_qdq')
        return y
    y = _out_like(x, out)
    G = max(lib.cnnq_pc_groups(N, C, HW, 1), lib.cnnq_pc_groups(N, C, HW, 0))
    if G <= 0:
        L.check(G, 'cnnq_pc_groups(%d,%d,%d)' % (N, C, HW))
    pmm = torch.empty((G, 2, C), dtype=torch.float32, device=x.device)
    qp = torch.empty((L.NQP, C), dtype=torch.float32, device=x.device)
    codes = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if want_codes else None
    hist = torch.zeros(256, dtype=torch.int64, device=x.device) if want_entropy else None
    g_used = lib.cnnq_pc_groups(N, C, HW, int(x.data_ptr() % 16 == 0))
    L.check(lib.cnnq_pc_minmax(_ptr(x), N, C, HW, _ptr(pmm), _stream(x)), 'cnnq_pc_minmax')
    local = torch.empty((2, C), dtype=torch.float32, device=x.device)
    L.check(lib.cnnq_pc_minmax_reduce(_ptr(pmm), g_used, C, _ptr(local), _stream(x)), 'cnnq_pc_minmax_reduce')
    pmm = D.all_gather_records(local, group)                     # [W, 2, C]
    L.check(lib.cnnq_pc_minmax_params(_ptr(pmm), world, C, int(num_bits), int(bool(positive)), _ptr(qp),
                                      _stream(x)), 'cnnq_pc_minmax_params')
    L.check(lib.cnnq_pc_qdq(_ptr(x), _ptr(y), N, C, HW, _ptr(qp), _ptr(codes), _ptr(hist), 1, _stream(x)),
            'cnnq_pc_qdq')
    if want_entropy:
        D.all_reduce_sum_(hist, group)
    res = [y]
    if want_codes:
        res.append(codes)
    if want_entropy:
        res.append(entropy_from_hist(hist))
    if want_parts:
        stats = torch.zeros((L.NSTAT, C), dtype=torch.float32, device=x.device)
        stats[L.STAT_MIN] = pmm[:world, 0].min(dim=0)[0]
        stats[L.STAT_MAX] = pmm[:world, 1].max(dim=0)[0]
        res.append(dict(stats=stats, qp=qp, diag=None))
    return res[0] if len(res) == 1 else tuple(res)


def _slice_ptr(t, c0, HW):
    return ctypes.c_void_p(t.data_ptr() + 4 * c0 * HW)


def _slice_aligned(t, c0, HW, stride):
    return int((t.data_ptr() + 4 * c0 * HW) % 16 == 0 and stride % 4 == 0)


def minmax_qdq_channel_slice(x, c0, c1, num_bits, positive=False, out=None):
    """Config 2 on the channel slice x[:, c0:c1] of a contiguous NCHW tensor, read and written in place of
    the parent (no slice copy): e.g. the branches of a concatenated output, each with its own quantizer
    settings.  Returns `out` (default: a new tensor shaped like x; only the slice is written)."""
    lib = L.load()
    x = _dev_f32(x, 'x')
    N, C, HW = geometry(x)
    if not 0 <= c0 < c1 <= C:
        raise L.CnnqError('bad channel slice [%d, %d) of %d' % (c0, c1, C))
    y = _out_like(x, out)
    Cs, stride = c1 - c0, C * HW
    G = lib.cnnq_pc_groups(N, Cs, HW, _slice_aligned(x, c0, HW, stride))
    if G <= 0:
        L.check(G, 'cnnq_pc_groups(%d,%d,%d)' % (N, Cs, HW))
    pmm = torch.empty((G, 2, Cs), dtype=torch.float32, device=x.device)
    qp = torch.empty((L.NQP, Cs), dtype=torch.float32, device=x.device)
    st = _stream(x)
    L.check(lib.cnnq_pc_minmax_strided(_slice_ptr(x, c0, HW), N, Cs, HW, stride, _ptr(pmm), st), 'cnnq_pc_minmax_strided')
    L.check(lib.cnnq_pc_minmax_params(_ptr(pmm), G, Cs, int(num_bits), int(bool(positive)), _ptr(qp), st),
            'cnnq_pc_minmax_params')
    L.check(lib.cnnq_pc_qdq_strided(_slice_ptr(x, c0, HW), _slice_ptr(y, c0, HW), N, Cs, HW, stride, _ptr(qp), None,
                                    None, 1, st), 'cnnq_pc_qdq_strided')
    return y


def quantize_pack4(x, qp):
    """x [N, C, H, W] + parameter table -> packed int4 codes (uint8, numel/2 bytes, two codes per byte)."""
    lib = L.load()
    x = _dev_f32(x, 'x')
    N, C, HW = geometry(x)
    packed = torch.empty(x.numel() // 2, dtype=torch.uint8, device=x.device)
    L.check(lib.cnnq_pc_quantize_pack4(_ptr(x), _ptr(packed), N, C, HW, _ptr(qp), _stream(x)), 'cnnq_pc_quantize_pack4')
    return packed


def dequantize_pack4(packed, shape, qp):
    """Inverse of quantize_pack4: the dequantized fp32 tensor (bit-identical to pc_qdq's output)."""
    lib = L.load()
    y = torch.empty(shape, dtype=torch.float32, device=packed.device)
    N, C, HW = geometry(y)
    L.check(lib.cnnq_pc_dequantize_pack4(_ptr(packed), _ptr(y), N, C, HW, _ptr(qp), _stream(y)),
            'cnnq_pc_dequantize_pack4')
    return y


def quantize_u8(x, qp):
    """x [N, C, H, W] + parameter table -> uint8 codes (one byte each, numel bytes): the stored format for
    quantizations of up to 8 bits."""
    lib = L.load()
    x = _dev_f32(x, 'x')
    N, C, HW = geometry(x)
    codes = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    L.check(lib.cnnq_pc_quantize_u8(_ptr(x), _ptr(codes), N, C, HW, _ptr(qp), _stream(x)), 'cnnq_pc_quantize_u8')
    return codes


def dequantize_u8(codes, qp):
    """Inverse of quantize_u8: the dequantized fp32 tensor (bit-identical to pc_qdq's output)."""
    lib = L.load()
    y = torch.empty(codes.shape, dtype=torch.float32, device=codes.device)
    N, C, HW = geometry(y)
    L.check(lib.cnnq_pc_dequantize_u8(_ptr(codes), _ptr(y), N, C, HW, _ptr(qp), _stream(y)), 'cnnq_pc_dequantize_u8')
    return y


def packed_capacity(shape):
    """Worst-case bytes of the packed format for an activation of this shape (every channel at 8 bits)."""
    N, C = shape[0], shape[1]
    HW = 1
    for d in shape[2:]:
        HW *= d
    return N * C * ((HW * 8 + 31) // 32 * 4)


def packed_layout(bits, HW):
    """rowoff int32 [C + 1] of the packed format for per-channel widths `bits` and H*W elements per row
    (cnnq_pc_packed_layout): row (n, c) starts at byte n * rowoff[C] + rowoff[c].  A function of the parameters only - with
    a fixed bit allocation it is computed once and handed to quantize_packed / dequantize_packed."""
    lib = L.load()
    bits = bits.contiguous()
    C = bits.numel()
    rowoff = torch.empty(C + 1, dtype=torch.int32, device=bits.device)
    L.check(lib.cnnq_pc_packed_layout(_ptr(bits), C, int(HW), _ptr(rowoff), _stream(bits)), 'cnnq_pc_packed_layout')
    return rowoff


def quantize_packed(x, qp, bits, out=None, form=0, rowoff=None):
    """x [N, C, H, W] + parameter table + per-channel bit widths (diag[DIAG_BITS] of pc_params with bit
    allocation) -> (packed uint8 [N * bytes_per_sample], rowoff int32 [C + 1]): bits[c] bits per code, i.e.
    sum(bits)/8 bytes per spatial position (cnnq_pc_quantize_packed).  Sizing the buffer exactly takes one host
    read of rowoff[C]; with `out` (a uint8 buffer of at least packed_capacity(x.shape) bytes) nothing synchronises
    and the whole buffer is returned (the used prefix is N * rowoff[C] bytes).  form: 0 = the library's choice, 1 = the
    general kernel, 2 = the lean kernel (cnnq_pc_quantize_packed_form); every form writes the same bytes.  rowoff: the
    layout of packed_layout(bits, H * W) when the caller already has it (one launch less)."""
    lib = L.load()
    x = _dev_f32(x, 'x')
    N, C, HW = geometry(x)
    bits = bits.contiguous()
    if rowoff is None:
        rowoff = torch.empty(C + 1, dtype=torch.int32, device=x.device)
        L.check(lib.cnnq_pc_packed_layout(_ptr(bits), C, HW, _ptr(rowoff), _stream(x)), 'cnnq_pc_packed_layout')
    elif not (rowoff.is_cuda and rowoff.dtype == torch.int32 and rowoff.is_contiguous() and rowoff.numel() == C + 1):
        raise L.CnnqError('rowoff must be the int32 device tensor [C + 1] of packed_layout(bits, H * W)')
    if out is not None:
        if not (out.is_cuda and out.dtype == torch.uint8 and out.is_contiguous() and out.numel() >= packed_capacity(x.shape)):
            raise L.CnnqError('out must be a contiguous uint8 device buffer of at least packed_capacity(x.shape) bytes')
        packed, plane = out, None
    else:
        plane = int(rowoff[C].item())
        packed = torch.empty(max(N * plane, 4), dtype=torch.uint8, device=x.device)
    L.check(lib.cnnq_pc_quantize_packed_form(_ptr(x), _ptr(packed), N, C, HW, _ptr(qp), _ptr(bits), _ptr(rowoff), int(form),
                                             _stream(x)), 'cnnq_pc_quantize_packed')
    return (packed if plane is None else packed[:N * plane]), rowoff


def dequantize_packed(packed, shape, qp, bits, rowoff, out=None, form=0):
    """Inverse of quantize_packed: the dequantized fp32 tensor (bit-identical to pc_qdq's output).  form as in
    quantize_packed (cnnq_pc_dequantize_packed_form)."""
    lib = L.load()
    y = torch.empty(shape, dtype=torch.float32, device=packed.device) if out is None else out
    if out is not None and not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
                                and tuple(out.shape) == tuple(shape)):
        raise L.CnnqError('out must be a contiguous float32 device tensor of the given shape')
    N, C, HW = geometry(y)
    L.check(lib.cnnq_pc_dequantize_packed_form(_ptr(packed), _ptr(y), N, C, HW, _ptr(qp), _ptr(bits.contiguous()),
                                               _ptr(rowoff), int(form), _stream(y)), 'cnnq_pc_dequantize_packed')
    return y


def entropy_from_hist(hist):
    """Shannon entropy (bits) of an int64 histogram tensor -> 0-dim float32 tensor on the device."""
    lib = L.load()
    out = torch.empty(1, dtype=torch.float32, device=hist.device)
    L.check(lib.cnnq_entropy(_ptr(hist), hist.numel(), _ptr(out), _stream(hist)), 'cnnq_entropy')
    return out[0]


def pt_setup(device, num_bits, range_offset=None, stats=None, rows=0, rows_mode=0, zero_min=False,
             int_exp=False, enforce_true_zero=True):
    """Per-tensor parameters ptp[8] on the device from host scalars or from per-row stats."""
    lib = L.load()
    ptp = torch.empty(8, dtype=torch.float32, device=device)
    ro = None
    if range_offset is not None:
        ro = (ctypes.c_float * 2)(float(range_offset[0]), float(range_offset[1]))
    stride = stats.shape[1] if stats is not None else 0
    L.check(lib.cnnq_pt_setup(ro, _ptr(stats), stride, int(rows), int(rows_mode), int(bool(zero_min)),
                              int(num_bits), int(bool(int_exp)), int(bool(enforce_true_zero)), _ptr(ptp),
                              _stream(ptp)), 'cnnq_pt_setup')
    return ptp


def pt_qdq(x, ptp, noise=None, out=None):
    lib = L.load()
    x = _dev_f32(x, 'x')
    y = _out_like(x, out)
    if noise is not None:
        noise = _dev_f32(noise, 'noise')
    if x.numel() == 0:
        return y
    L.check(lib.cnnq_pt_qdq(_ptr(x), _ptr(y), x.numel(), _ptr(ptp), _ptr(noise), _stream(x)), 'cnnq_pt_qdq')
    return y


# ------------------------------------------------------------------------------------- pipelines
def act_qdq_per_channel(x, num_bits, positive=False, clip='no', bit_alloc=False, prior_is_b=False, target=None,
                        round_mode=True, per_channel_dim=1, group=None, want_codes=False, want_parts=False,
                        stats=None, want_entropy=False, whole_tensor=False, out=None, bcorr=None):
    """The dynamic per-channel hot path end to end: statistics (one or two coalesced reads of
    x) -> parameters (one workgroup) -> fused Q/DQ (one read, one write).  Covers iq.py:409-451
    (clip='no'), iq.py:327-352 (ACIQ) and, with per_channel_dim=0, the weights of iq.py:453-476;
    whole_tensor=True treats the tensor as ONE channel (per-tensor clipping, iq.py:353-357).
    `stats` (optional [NSTAT, C] table, e.g. from a calibration file) replaces the dynamic
    statistics.  group=False: never exchange (replicated data such as weights).
    bcorr (None, or the relu-first flag): also apply the activation bias correction of iqm.py:180-196,
    fused into the passes where the parameter table is at hand (qdq_bias_corrected).
    Returns y [, codes] [, entropy (0-dim device tensor)] [, parts].  No host synchronisation."""
    x = _dev_f32(x, 'x')
    N, C, HW = (1, 1, x.numel()) if whole_tensor else geometry(x, per_channel_dim)
    use_ba = bool(bit_alloc) and num_bits <= 4 and not whole_tensor
    world = 1 if group is False else D.world_size(group)
    if bcorr is not None and (want_codes or want_entropy or want_parts or whole_tensor or per_channel_dim != 1):
        raise L.CnnqError('bcorr combines only with the plain per-channel activation Q/DQ')
    if stats is None and clip == 'no' and not use_ba and not whole_tensor:
        res = minmax_qdq_fused(x, N, C, HW, num_bits, positive, want_codes, want_entropy, out=out,
                               want_parts=want_parts, group=group, _checked=True)
        if bcorr is not None:
            res = act_bias_correction_(x, res, bool(bcorr), group=None if group is False else group)
        return res
    # Laplace clipping with dynamic statistics on one GPU: pass B, the parameters and the Q/DQ in ONE launch that reads
    # x once (cnnq_pc_aciq_qdq_single: 12 instead of 16 bytes per element) when the shape has a single-launch plan
    exchanging = group is not False and (world > 1 or D.forced_exchange())     # x is this rank's shard of the batch
    single = (_ACIQ_SINGLE and _RESIDENT and stats is None and bcorr is None and clip == 'laplace'
              and not whole_tensor and not (use_ba and prior_is_b) and num_bits <= 8)
    if single and exchanging and not (want_codes or want_entropy):
        # sharded: the ranks' sums meet inside the single launch (round 6); None: no in-launch exchange for this group - the chain
        cfg = _params_cfg(num_bits, positive, clip, use_ba, prior_is_b, target, round_mode, whole_tensor)
        res = _aciq_qdq_xrank(x, N, C, HW, cfg, group, want_parts, out)
        if res is not None:
            return res
    single = single and not exchanging
    if (stats is None and not exchanging and bcorr is None and not (want_codes or want_entropy or want_parts)):
        # one host call for the whole pipeline (cnnq_pc_aciq_qdq_auto: four launches through the single-launch kernel,
        # else the five of the chain), one cached workspace (statistics partials, then the parameter and diagnostic
        # tables, which nobody outside the call reads)
        lib = L.load()
        cfg = _params_cfg(num_bits, positive, clip, use_ba, prior_is_b, target, round_mode, whole_tensor)
        y = _out_like(x, out)
        al = int(x.data_ptr() % 16 == 0)
        key = ('aciq', N, C, HW, al)
        nbytes = _WS_BYTES.get(key)
        if nbytes is None:
            nbytes = _WS_BYTES[key] = (lib.cnnq_pc_aciq_workspace(N, C, HW, al) + 15) // 16 * 16
        st = _raw_stream(x.device.index)
        base = _scratch(x, 'aciq', nbytes + (L.NQP + L.NDIAG) * C * 4, st).data_ptr()
        qd = base + nbytes
        gws = _group_workspace(x, st) if single else None
        if gws is not None:
            rc = lib.cnnq_pc_aciq_qdq_auto(x.data_ptr(), y.data_ptr(), N, C, HW, ctypes.byref(cfg), base, gws, GROUP_WS_BYTES, qd,
                                           qd + L.NQP * C * 4, st)
        else:
            rc = lib.cnnq_pc_aciq_qdq(x.data_ptr(), y.data_ptr(), N, C, HW, ctypes.byref(cfg), base, qd, qd + L.NQP * C * 4, st)
        if rc:
            L.check(rc, 'cnnq_pc_aciq_qdq')
        return y
    if single:
        res = aciq_qdq_single(x, N, C, HW, num_bits, positive, use_ba, target, round_mode, want_codes, want_entropy, want_parts,
                              out=out)
        if res is not None:
            return res
    if stats is None:
        need_b = (clip == 'laplace') or (use_ba and prior_is_b)
        stats, _ = pc_stats(x, N, C, HW, need_b=need_b, group=None if group is False else group,
                            local_only=group is False)
    qp, diag = pc_params(stats, num_bits, positive, clip, use_ba, prior_is_b, target, round_mode,
                         direct_range=whole_tensor)
    if bcorr is not None:
        return qdq_bias_corrected(x, N, C, HW, qp, bool(bcorr), group=None if group is False else group, out=out)
    hist = torch.zeros(256, dtype=torch.int64, device=x.device) if want_entropy else None
    res = pc_qdq(x, N, C, HW, qp, want_codes, out=out, hist=hist)
    out = list(res) if want_codes else [res]
    if want_entropy:
        if D.world_size(None if group is False else group) > 1 and group is not False:
            D.all_reduce_sum_(hist, group)
        out.append(entropy_from_hist(hist))
    if want_parts:
        out.append(dict(stats=stats, qp=qp, diag=diag))
    return out[0] if len(out) == 1 else tuple(out)


def weight_correction(w, w_q, vcorr=False, bcorr=False):
    """Per-output-channel variance / mean correction of quantized weights (iqm.py:374-391).
    Returns a new tensor; weights are replicated across ranks, so no exchange."""
    if not (vcorr or bcorr):
        return w_q
    lib = L.load()
    w = _dev_f32(w, 'w')
    out = _dev_f32(w_q, 'w_q').clone()
    C, HW = w.shape[0], w.numel() // w.shape[0]
    st_w, _ = pc_stats(w, 1, C, HW, local_only=True)
    st_q, _ = pc_stats(out, 1, C, HW, local_only=True)
    L.check(lib.cnnq_pc_weight_correct(_ptr(out), C, HW, _ptr(st_w), _ptr(st_q), int(bool(vcorr)), int(bool(bcorr)),
                                       _stream(out)), 'cnnq_pc_weight_correct')
    return out.view(w_q.shape)


def act_bias_correction_(out, out_q, relu_first, group=None):
    """Activation bias correction (iqm.py:180-196), IN PLACE on out_q; `out` is the unquantized
    activation.  With world size > 1 the per-channel sums are exchanged so the bias is global."""
    lib = L.load()
    x = _dev_f32(out, 'out')
    if not (isinstance(out_q, torch.Tensor) and out_q.is_cuda and out_q.is_contiguous() and out_q.dtype == torch.float32):
        raise L.CnnqError('out_q must be a contiguous float32 device tensor')
    N, C, HW = geometry(x)
    G = lib.cnnq_pc_groups(N, C, HW, int(x.data_ptr() % 16 == 0 and out_q.data_ptr() % 16 == 0))
    part3 = torch.empty((G, 3, C), dtype=torch.float64, device=x.device)
    L.check(lib.cnnq_pc_bcorr_sums(_ptr(x), _ptr(out_q), N, C, HW, int(bool(relu_first)), _ptr(part3), _stream(x)),
            'cnnq_pc_bcorr_sums')
    bias = torch.empty(C, dtype=torch.float32, device=x.device)
    if D.world_size(group) > 1:
        sums = torch.empty((3, C), dtype=torch.float64, device=x.device)
        L.check(lib.cnnq_pc_bcorr_bias(_ptr(part3), G, C, _ptr(sums), None, _stream(x)), 'cnnq_pc_bcorr_bias')
        part3 = D.all_gather_records(sums, group)
        G = part3.shape[0]
    L.check(lib.cnnq_pc_bcorr_bias(_ptr(part3), G, C, None, _ptr(bias), _stream(x)), 'cnnq_pc_bcorr_bias')
    L.check(lib.cnnq_pc_bcorr_apply(_ptr(out_q), N, C, HW, _ptr(bias), _stream(x)), 'cnnq_pc_bcorr_apply')
    return out_q


def qdq_bias_corrected(x, N, C, HW, qp, relu_first, group=None, out=None):
    """Q/DQ with the parameter table qp followed by the activation bias correction, without ever
    storing the uncorrected tensor: one read-only pass over x for the per-channel sums (the quantized
    value is recomputed on the fly) and one fused quantize+correct pass - 12 B/elem instead of 24, the
    same floats as pc_qdq + act_bias_correction_."""
    lib = L.load()
    x = _dev_f32(x, 'x')
    y = _out_like(x, out)
    G = lib.cnnq_pc_groups(N, C, HW, int(x.data_ptr() % 16 == 0))
    if G <= 0:
        L.check(G, 'cnnq_pc_groups(%d,%d,%d)' % (N, C, HW))
    part3 = torch.empty((G, 3, C), dtype=torch.float64, device=x.device)
    L.check(lib.cnnq_pc_qdq_bcorr_sums(_ptr(x), N, C, HW, _ptr(qp), int(bool(relu_first)), _ptr(part3), _stream(x)),
            'cnnq_pc_qdq_bcorr_sums')
    bias = torch.empty(C, dtype=torch.float32, device=x.device)
    if D.world_size(group) > 1:
        sums = torch.empty((3, C), dtype=torch.float64, device=x.device)
        L.check(lib.cnnq_pc_bcorr_bias(_ptr(part3), G, C, _ptr(sums), None, _stream(x)), 'cnnq_pc_bcorr_bias')
        part3 = D.all_gather_records(sums, group)
        G = part3.shape[0]
    L.check(lib.cnnq_pc_bcorr_bias(_ptr(part3), G, C, None, _ptr(bias), _stream(x)), 'cnnq_pc_bcorr_bias')
    L.check(lib.cnnq_pc_qdq_bcorr(_ptr(x), _ptr(y), N, C, HW, _ptr(qp), _ptr(bias), 1, _stream(x)), 'cnnq_pc_qdq_bcorr')
    return y


_MT_TABLES = {}


def _midtread_tables(device):
    """The (omega, alpha) interpolation tables of iq.py:41-51 as a device fp64 [2, 101] tensor."""
    key = str(device)
    if key not in _MT_TABLES:
        from .qtypes._midtread_tables import ALPHA_TABLE, OMEGA_TABLE
        _MT_TABLES[key] = torch.tensor([OMEGA_TABLE, ALPHA_TABLE], dtype=torch.float64, device=device)
    return _MT_TABLES[key]


def mid_tread_qdq(x, target, clip, sym, per_channel_dim=1, whole_tensor=False, group=None, want_entropy=False,
                  want_codes=False, want_parts=False):
    """Mid-tread quantization with per-channel bin allocation (iq.py:147-225): statistics ->
    cnnq_pc_midtread_params -> cnnq_pc_midtread_qdq (+ histogram -> entropy).  Returns
    (y, entropy or None [, codes] [, parts])."""
    lib = L.load()
    x = _dev_f32(x, 'x')
    N, C, HW = (1, 1, x.numel()) if whole_tensor else geometry(x, per_channel_dim)
    local = group is False
    grp = None if local else group
    tabs = _midtread_tables(x.device)
    exchanging = not local and (D.world_size(grp) > 1 or D.forced_exchange())
    if _ACIQ_SINGLE and _RESIDENT and clip and not whole_tensor and per_channel_dim == 1 and not want_codes and exchanging:
        # sharded: pass A through the collective, the ranks' sums of |x - mean| inside the single launch (round 6)
        res = _mid_tread_qdq_xrank(x, N, C, HW, target, sym, tabs, grp, want_entropy, want_parts)
        if res is not None:
            return res
    if (_ACIQ_SINGLE and _RESIDENT and clip and not whole_tensor and per_channel_dim == 1 and not want_codes
            and not exchanging):
        # pass B, the step sizes / clamp bounds and the quantization in ONE launch that reads x once
        # (cnnq_pc_midtread_qdq_single: 12 instead of 16 bytes per element) when the shape has a single-launch plan
        res = mid_tread_qdq_single(x, N, C, HW, target, sym, tabs, want_entropy, want_parts)
        if res is not None:
            return res
    stats, mom = pc_stats(x, N, C, HW, need_b=bool(clip), group=grp, local_only=local)
    mt = torch.empty((L.NMT, C), dtype=torch.float32, device=x.device)
    L.check(lib.cnnq_pc_midtread_params(_ptr(stats), C, float(target), int(bool(clip)), int(bool(sym)), _ptr(tabs),
                                        tabs.shape[1], _ptr(mt), _stream(x)), 'cnnq_pc_midtread_params')
    y = torch.empty_like(x)
    codes = torch.empty_like(x) if want_codes else None
    hist = torch.zeros(L.mt_hist_words(C), dtype=torch.int64, device=x.device) if want_entropy else None
    L.check(lib.cnnq_pc_midtread_qdq(_ptr(x), _ptr(y), N, C, HW, _ptr(mt), int(bool(clip)), _ptr(codes), _ptr(hist),
                                     _stream(x)), 'cnnq_pc_midtread_qdq')
    entropy = None
    if want_entropy:
        world = 1 if local else D.world_size(grp)
        if world > 1:
            D.all_reduce_sum_(hist, grp)
        ent = torch.empty(1, dtype=torch.float32, device=x.device)
        if world > 1:       # the global batch's element count from the merged moment record (shards may differ by a sample)
            L.check(lib.cnnq_midtread_entropy_count(_ptr(hist), _ptr(mt), C, mom[L.MOM_COUNT].data_ptr(), _ptr(ent), _stream(x)),
                    'cnnq_midtread_entropy')
        else:
            L.check(lib.cnnq_midtread_entropy(_ptr(hist), _ptr(mt), C, x.numel(), _ptr(ent), _stream(x)), 'cnnq_midtread_entropy')
        entropy = ent[0]
    res = [y, entropy]
    if want_codes:
        res.append(codes)
    if want_parts:
        res.append(dict(stats=stats, mt=mt, hist=hist))
    return tuple(res)


def mid_tread_qdq_single(x, N, C, HW, target, sym, tabs, want_entropy=False, want_parts=False, flags=0):
    """Config 5 with clipping in four launches (cnnq_pc_midtread_qdq_single) + the entropy kernel.  Returns what mid_tread_qdq
    returns, or None when the shape has no single-launch plan."""
    lib = L.load()
    st = _raw_stream(x.device.index)
    gws = _group_workspace(x, st)
    if gws is None:
        return None
    al = int(x.data_ptr() % 16 == 0)
    key = ('aciq', N, C, HW, al)
    nbytes = _WS_BYTES.get(key)
    if nbytes is None:
        nbytes = _WS_BYTES[key] = (lib.cnnq_pc_aciq_workspace(N, C, HW, al) + 15) // 16 * 16
    ws = _scratch(x, 'aciq', nbytes + (L.NQP + L.NDIAG) * C * 4, st)
    y = torch.empty_like(x)
    tabs_out = torch.empty((L.NSTAT + L.NMT, C), dtype=torch.float32, device=x.device)
    stats, mt = tabs_out[:L.NSTAT], tabs_out[L.NSTAT:]
    hist = torch.empty(L.mt_hist_words(C), dtype=torch.int64, device=x.device) if want_entropy else None     # zeroed by the call
    rc = lib.cnnq_pc_midtread_qdq_single(_ptr(x), _ptr(y), N, C, HW, float(target), int(bool(sym)), _ptr(tabs), tabs.shape[1],
                                         ws.data_ptr(), gws, GROUP_WS_BYTES, _ptr(stats), _ptr(mt), _ptr(hist), int(flags), st)
    if rc == L.ENOTSUP:
        return None
    L.check(rc, 'cnnq_pc_midtread_qdq_single')
    entropy = None
    if want_entropy:
        if _ENT_BATCH is not None:
            entropy = _ENT_BATCH.add_midtread(hist, mt, C, x.numel(), x.device)      # one launch for the whole block, at its end
        else:
            ent = torch.empty(1, dtype=torch.float32, device=x.device)
            L.check(lib.cnnq_midtread_entropy(_ptr(hist), _ptr(mt), C, x.numel(), _ptr(ent), st), 'cnnq_midtread_entropy')
            entropy = ent[0]
    res = [y, entropy]
    if want_parts:
        res.append(dict(stats=stats, mt=mt, hist=hist))
    return tuple(res)


def tensor_row_stats(x, rows):
    """Per-row MIN/MAX (and friends) of x viewed as [rows, numel/rows]: the per-sample statistics
    of iq.py:510-517 (rows = batch) - the per-channel kernels with N = 1, C = rows."""
    lib = L.load()
    x = _dev_f32(x, 'x')
    hw = x.numel() // rows
    G = lib.cnnq_pc_groups(1, rows, hw, int(x.data_ptr() % 16 == 0))
    if G <= 0:
        L.check(G, 'cnnq_pc_groups(1,%d,%d)' % (rows, hw))
    pmm = torch.empty((G, 2, rows), dtype=torch.float32, device=x.device)
    L.check(lib.cnnq_pc_minmax(_ptr(x), 1, rows, hw, _ptr(pmm), _stream(x)), 'cnnq_pc_minmax')
    # rows MIN (0) and MAX (1) of a stats table with stride `rows`: what cnnq_pt_setup reads
    table = torch.empty((2, rows), dtype=torch.float32, device=x.device)
    L.check(lib.cnnq_pc_minmax_reduce(_ptr(pmm), G, rows, _ptr(table), _stream(x)), 'cnnq_pc_minmax_reduce')
    return table


def minmax_qdq_per_tensor(x, num_bits, avg_over_batch, zero_min=False, int_exp=False, enforce_true_zero=True,
                          group=None, fused=None):
    """iq.py:361-379 + 605-614 with dynamic statistics: per-sample min/max, their batch mean
    (or the whole-tensor min/max), then the GEMMLOWP kernel - all on the device, four launches.
    fused=True (default: CNNQ_PT_FUSED=1) takes the ONE-launch form on a single GPU (cnnq_pt_minmax_qdq_fused): the same
    bits, but measured SLOWER than the chain - 70 against 54 us on the [32,64,112,112] tensor of BASELINE config 1: its
    second sweep is not served by the Infinity Cache once 103 MB of y are written next to it - so
    it is opt-in."""
    x = _dev_f32(x, 'x')
    rows = x.shape[0] if x.dim() > 1 else 1
    if (_PT_FUSED if fused is None else fused) and D.world_size(group) == 1 and x.numel() > 0:
        # one launch (k_pt_fused): two sweeps with a tile count in between, the second served by the Infinity Cache
        st = _raw_stream(x.device.index)
        gws = _group_workspace(x, st)
        if gws is not None:
            y = torch.empty_like(x)
            rc = L.load().cnnq_pt_minmax_qdq_fused(x.data_ptr(), y.data_ptr(), x.numel(), rows, 0 if avg_over_batch else 1,
                                                   int(bool(zero_min)), int(num_bits), int(bool(int_exp)),
                                                   int(bool(enforce_true_zero)), gws, GROUP_WS_BYTES, None, st)
            if rc == 0:
                return y
            if rc != L.ENOTSUP:
                L.check(rc, 'cnnq_pt_minmax_qdq_fused')
    stats = tensor_row_stats(x, rows)
    if D.world_size(group) > 1:
        stats = D.merge_row_minmax(stats, rows, avg_over_batch, group)
        rows = stats.shape[1]
    ptp = pt_setup(x.device, num_bits, stats=stats, rows=rows, rows_mode=0 if avg_over_batch else 1,
                   zero_min=zero_min, int_exp=int_exp, enforce_true_zero=enforce_true_zero)
    return pt_qdq(x, ptp)


def kld_thresholds(x, rows=None, want_parts=False):
    """`-kld` calibration (inference/kld_threshold.py:6-84 per sample, statistic_manager.py:80-82):
    x viewed as [rows, numel/rows] (rows = batch samples) -> float64 tensor [rows, 3] =
    {optimal clipping threshold, its KL divergence, candidate index}; the `kld_th` statistic of
    the batch is out[:, 0].max().  want_parts adds (hist [rows, 2001] int32, div [rows, 994])."""
    lib = L.load()
    x = _dev_f32(x, 'x')
    rows = int(rows if rows is not None else (x.shape[0] if x.dim() > 1 else 1))
    length = x.numel() // rows
    rowmm = tensor_row_stats(x, rows)
    hist = torch.empty((rows, L.KLD_BINS), dtype=torch.int32, device=x.device)
    div = torch.empty((rows, L.KLD_NCAND), dtype=torch.float64, device=x.device)
    out = torch.empty((rows, 3), dtype=torch.float64, device=x.device)
    L.check(lib.cnnq_kld_hist(_ptr(x), rows, length, _ptr(rowmm), _ptr(hist), _stream(x)), 'cnnq_kld_hist')
    L.check(lib.cnnq_kld_search(_ptr(hist), rows, _ptr(rowmm), _ptr(div), _ptr(out), _stream(x)), 'cnnq_kld_search')
    if want_parts:
        return out, hist, div
    return out


def row_sumsq(x, rows=None):
    """Per-sample sum of squares, the runtime distance measure of distance_stats.py:22-33
    (`torch.sum(t**2, dim=-1)` on [N, -1]): the moments kernel with N = 1, C = rows (fp64 sums)."""
    x = _dev_f32(x, 'x')
    rows = int(rows if rows is not None else x.shape[0])
    _, mom = pc_stats(x, 1, rows, x.numel() // rows, local_only=True)
    return mom[L.MOM_SUMSQ].to(torch.float32)
