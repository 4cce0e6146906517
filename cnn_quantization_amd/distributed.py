"""Multi-GPU layer: one process per GPU (torchrun), activations sharded along the batch, and ONE
small exchange per statistics pass over RCCL/xGMI (torch.distributed backend "nccl" is RCCL on
ROCm; the CPU tests drive the same code over gloo).  Two routes for config 2, and only two: the collective (default:
statistics launch -> ncclAllGather on the compute stream -> Q/DQ launch) and, opt-in, the exchange inside the single launch
(XRankExchange: x is read once).

The reference has no counterpart: its DataParallel replicas each quantize with the statistics
of their own sub-batch (inference/inference_sim.py:196-200).  Here every rank ends up with the
statistics of the GLOBAL batch, so an N-GPU run reproduces the single-GPU result; min/max are
exact, hence config 2 is bit-identical for any world size.

What travels: mergeable fp64 records [K, C] (min, max, sum, sumsq, count, ...) - at most
7 * 2048 * 8 B = 115 KB per rank, latency-bound; the Q/DQ itself needs no communication.
An all_gather followed by a merge in rank order (cnnq_pc_combine) is used instead of typed
all-reduces so that the result does not depend on the collective's reduction order."""
import torch
import torch.distributed as dist


def world_size(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size(group)


def forced_exchange():
    """CNNQ_FORCE_EXCHANGE=1 with an initialised (1-rank) process group: take the multi-GPU launch sequence even
    for world size 1, so that the real collective (RCCL `nccl` backend) runs and can be timed on a 1-GPU box."""
    import os
    return os.environ.get('CNNQ_FORCE_EXCHANGE', '0') == '1' and dist.is_available() and dist.is_initialized()


def rank(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    return dist.get_rank(group)


def shard_batch(n, rank_, world):
    """[n0, n1) of the batch owned by `rank_`: contiguous, sizes differ by at most one."""
    base, rem = divmod(n, world)
    n0 = rank_ * base + min(rank_, rem)
    return n0, n0 + base + (1 if rank_ < rem else 0)


def all_gather_records(rec, group=None, out=None):
    """rec [K, C] on every rank -> [W, K, C] in rank order (the G axis cnnq_pc_combine merges): RCCL's all_gather enqueued
    directly on the caller's stream when the backend is `nccl` (rccl.py), else torch.distributed's (gloo rigs).  `out`:
    optional preallocated [W, K, C] result."""
    return collective_all_gather(rec, group, out)


def collective_all_gather(rec, group=None, out=None):
    w = world_size(group)
    rec = rec.contiguous()
    if out is None:
        out = torch.empty((w,) + tuple(rec.shape), dtype=rec.dtype, device=rec.device)
    if w == 1 and not forced_exchange():
        out[0].copy_(rec)
        return out
    if rec.is_cuda:
        from . import rccl
        dc = rccl.direct_comm(group)                 # RCCL on the caller's stream, ~3 us of host time (rccl.py)
        if dc is not None:
            return dc.all_gather(rec, out)
    try:
        dist.all_gather_into_tensor(out.view(-1), rec.view(-1), group=group)   # flat: same on RCCL and gloo
    except RuntimeError:
        # gloo with device tensors (single-GPU test rigs): list form
        dist.all_gather([out[i] for i in range(w)], rec, group=group)
    return out


class XRankExchange:
    """The windows of the IN-LAUNCH cross-rank exchange of config 2 (csrc/cnnq_xrank.hip.h, cnnq_pc_minmax_qdq_xrank_dev): with
    the batch sharded over the ranks of `group` (one process per GPU, one node), the single-launch kernels push their
    channels' extrema into every rank's window and wait for the others' inside the launch - x is read once, 8 instead of
    12 bytes per element.

    OPT-IN (xrank_mode: CNNQ_XRANK=1 / auto, or set_xrank_mode): a rank that waits for a peer spins INSIDE the kernel, bounded
    by `timeout` (default 60 s of the 100 MHz clock: the order of a collective's timeout, since host-side skew between the
    ranks - a data loader, a checkpoint - turns into waiting time here); when the bound expires that rank's outputs of the
    launch are NaN, the status word is raised and every later wait gives up at once.  The caller must then agree over the
    group and fall back to the collective (disable_xrank, as bench.py does); minmax_qdq raises CnnqError at its periodic
    check, and healthy() is the synchronising check to run at the program's own synchronisation points and AFTER REPLAYS
    of a captured graph (a replay never passes through this class, so nothing else would notice).  `verify()` cross-checks
    the exchange against the collective path on every rank and must pass before use.  One stream per group, every rank issues
    the same launches in the same order.  Launches CAN be captured into a HIP graph (the sequence number is a device word);
    the windows of an exchange whose launches were captured stay mapped after close() so that a stale replay cannot fault."""
    CMAX = 16384                                        # slots per (parity, rank) block: 4 MB at 8 ranks (config 2 and the fused clipping
                                                        # kernels use C slots per launch, the statistics kernel 8 C: 2048 channels fit)
    TIMEOUT_TICKS = 6000000000                          # 60 s of the 100 MHz clock
    CHECK_EVERY = 64                                    # launches between two host checks of the status word (about one per
                                                        # ResNet-50 forward)

    def __init__(self, group=None):
        """Collective: every rank of `group` must call it.  Local failures (allocation, IPC export / import)
        never skip a collective step; `self.ok` is the group-wide verdict (all ranks agree on it)."""
        import ctypes
        import os
        from . import _lib as L
        self.group, self.L, self.lib = group, L, L.load()
        self.world, self.rank = world_size(group), rank(group)
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.mapped, self.own, self.why = [], None, ''
        own = ctypes.c_void_p()
        handle = ctypes.create_string_buffer(64)
        local_ok = True
        try:
            L.check(self.lib.cnnq_xrank_alloc(self.world, self.CMAX, ctypes.byref(own), handle), 'cnnq_xrank_alloc')
            self.own = own
        except Exception as e:
            local_ok, self.why = False, str(e)
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(handle.raw) if local_ok else b'', group=group)
        ptrs = []
        if local_ok and all(len(h) == 64 for h in handles):
            try:
                for r, h in enumerate(handles):
                    if r == self.rank:
                        ptrs.append(own.value)
                        continue
                    w = ctypes.c_void_p()
                    L.check(self.lib.cnnq_xrank_open(ctypes.create_string_buffer(h, 64), ctypes.byref(w)), 'cnnq_xrank_open')
                    self.mapped.append(w)
                    ptrs.append(w.value)
            except Exception as e:
                local_ok, self.why = False, str(e)
        else:
            local_ok = False
        self.ok = self._all_agree(local_ok)             # also: every window is mapped before anyone pushes
        if self.ok:
            self.windows = torch.tensor(ptrs, dtype=torch.int64, device=self.device)
            self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.seq = 0
        self.calls = 0
        self.captured = 0                               # launches recorded into HIP graphs
        self.stream = None
        ms = os.environ.get('CNNQ_XRANK_TIMEOUT_MS')
        self.timeout = int(float(ms) * 1e5) if ms else self.TIMEOUT_TICKS
        self.seq_dev = None                             # eight device words: [0] the sequence word, [4..7] the slots in use per parity
                                                        # (kept by the launches themselves; allocated at the first launch)
        fa = os.environ.get('CNNQ_XRANK_TEST_FAIL_AT')    # tests: rank 0 reports an expired wait at its n-th launch
        self.fail_at = int(fa) if (fa and self.rank == 0) else 0
        ea = os.environ.get('CNNQ_XRANK_TEST_EXPIRE_AT')  # tests: rank 0's status word reports an expired wait from its n-th launch on
        self.expire_at = int(ea) if (ea and self.rank == 0) else 0

    def _all_agree(self, flag):
        """Group-wide AND of a local boolean (a collective)."""
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.device)
        try:
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        except RuntimeError:                             # gloo without device tensors
            t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    def fits(self, C, words=1):
        """A launch over C channels with `words` slots per channel fits the windows."""
        return 0 < C * words <= self.CMAX

    def healthy(self):
        """Host check (synchronises): no wait for a peer has expired so far.  Call it at synchronisation points and after
        replays of a graph that holds exchange launches."""
        return int(self.status.item()) == 0

    def healthy_so_far(self):
        """The periodic check of the hot path WITHOUT a synchronisation: the status word is copied to pinned host memory
        behind the launches enqueued so far, and what the PREVIOUS copy brought back is looked at once its event has
        completed - an expired wait is noticed one interval later than with healthy(), the GPU never drains for it."""
        snap = getattr(self, '_snap', None)
        ok = True
        if snap is not None and snap[1].query():
            ok = int(snap[0].item()) == 0
            snap = None
        if snap is None:
            if getattr(self, '_host', None) is None:
                self._host, self._ev = torch.empty(1, dtype=torch.int32, pin_memory=True), torch.cuda.Event()
            self._host.copy_(self.status, non_blocking=True)
            self._ev.record()
            snap = (self._host, self._ev)
        self._snap = snap
        return ok

    def _next_launch(self, st):
        """The bookkeeping every exchanging launch shares (all four kinds draw from ONE sequence per stream): returns the host's
        launch number, or 0 from the first captured launch on (device numbering: a replay advances the device word, not
        self.seq, and a one-workgroup kernel behind each launch zeroes its slots).  Which slots a launch leaves behind is
        recorded on the device by the launch itself (round 6; ADVICE r5: the host's memory of the last two channel counts was
        wrong for a graph captured after eager launches with more channels, replayed out of order, ...)."""
        capturing = torch.cuda.is_current_stream_capturing()
        if self.stream is None:
            self.stream = st
        elif self.stream != st:
            raise self.L.CnnqError('XRankExchange is bound to the stream of its first launch; use one stream per group')
        if self.seq_dev is None:
            if capturing:
                raise self.L.CnnqError('XRankExchange: run one launch eagerly before capturing (the sequence word is allocated at first use)')
            self.seq_dev = torch.zeros(8, dtype=torch.int32, device=self.device)
        self.seq += 1                                   # launches ENQUEUED here
        self.calls += 1
        self.captured += 1 if capturing else 0
        if (_RECOVERY == 'raise' and not capturing and self.calls % self.CHECK_EVERY == 0
                and not self.healthy_so_far()):         # periodic host check (no synchronisation); 'checkpoint': the program checks itself
            raise self.L.CnnqError('XRankExchange: a wait for a peer expired; results since the last check are invalid')
        if self.fail_at and self.calls == self.fail_at:
            raise self.L.CnnqError('XRankExchange: CNNQ_XRANK_TEST_FAIL_AT (test hook)')
        if self.expire_at and self.calls == self.expire_at and not capturing:
            self.status.fill_(4)                        # test hook: as if a wait for a peer had expired in the previous launch
        return 0 if self.captured else self.seq

    def ctx(self, st):
        """The cnnq_xrank_ctx of the NEXT launch on stream st (consumes a launch number)."""
        c = self.L.XRankCtx()
        c.windows = self.windows.data_ptr()
        c.rank, c.world, c.cmax = self.rank, self.world, self.CMAX
        c.seq = self._next_launch(st)
        c.seq_dev = self.seq_dev.data_ptr()
        c.status = self.status.data_ptr()
        c.timeout_ticks = self.timeout
        return c

    def minmax_qdq(self, x, y, N, C, HW, num_bits, positive, ws_ptr, gws, gws_bytes, st, codes=None, hist_rep=None):
        """Enqueue config 2 of this rank's shard x -> y with the in-launch exchange; qp / mm (the GLOBAL extrema) land at the
        start of the workspace at `ws_ptr` (cnnq_pc_minmax_qdq_workspace bytes).  codes / hist_rep: this rank's codes and
        code counts (the replica tables of the single-launch kernels).  ONE launch per tensor (cnnq_pc_minmax_qdq_xrank_seq:
        the host numbers the launches, workgroup 0 cleans up behind the launch two back).  The call may be captured into a
        HIP graph: from the first captured launch on the sequence number lives in the device word and a one-workgroup kernel
        behind each launch advances it - every rank then replays its graph the same number of times, and checks healthy()
        after its replays."""
        host_seq = self._next_launch(st)
        rc = self.lib.cnnq_pc_minmax_qdq_xrank_seq(x.data_ptr(), y.data_ptr(), N, C, HW, int(num_bits), 1 if positive else 0,
                                                   ws_ptr, gws, gws_bytes, self.windows.data_ptr(), self.rank, self.world,
                                                   self.CMAX, host_seq, self.seq_dev.data_ptr(), 0, self.status.data_ptr(),
                                                   self.timeout, codes.data_ptr() if codes is not None else None,
                                                   hist_rep.data_ptr() if hist_rep is not None else None, st)
        if rc:
            self.L.check(rc, 'cnnq_pc_minmax_qdq_xrank_seq')

    def verify(self, rounds=12):
        """Config 2 of random shards through the in-launch exchange and through the collective path, bit for bit, on
        every rank: tile shapes of all three kernels and one without a single-launch kernel."""
        from . import ops
        g = torch.Generator(device=self.device).manual_seed(4321 + self.rank)
        shapes = [(6, 8, 14, 14), (40, 6, 56, 56), (70, 40, 7, 7), (37, 24, 14, 14), (3, 16, 5, 9), (8, 64, 7, 7)]
        ok = True
        for i in range(rounds):
            n, c, h, w = shapes[i % len(shapes)]
            x = torch.randn((n, c, h, w), generator=g, device=self.device) * (1 + self.rank) + 0.1 * i
            half = bool(i & 1)
            ref = ops.minmax_qdq_fused(x, n, c, h * w, 4, half, group=self.group, _xrank=False)
            got = ops.minmax_qdq_fused(x, n, c, h * w, 4, half, group=self.group, _xrank=self)
            ok = ok and bool(torch.equal(ref, got))
        # ... and the SUMS (round 6): the seven statistics of the global batch through the windows (eight words per channel; a
        # flat-tile shard and one without a plan) against the chain around the collective - extrema and count exact, the sums
        # within the tier of fp32 sums of 8 (the single launch) against sums of 4 (the chain) accumulated in fp64
        from . import _lib as L
        for (n, c, h, w) in ((40, 6, 56, 56), (37, 24, 14, 14)):
            x = torch.randn((n, c, h, w), generator=g, device=self.device) * (1 + self.rank) - 0.3
            plan = (self, self.group, ops._scratch(x, 'stats', L.load().cnnq_pc_stats_workspace(n, c, h * w, 1), ops._raw_stream(self.device.index)),
                    ops._group_workspace(x))
            ops._XPLAN[('stats', id(self.group), x.device.index, ops._raw_stream(self.device.index), n, c, h * w, x.data_ptr() % 16 == 0)] = plan
            st, mom = ops._pc_stats_xrank(x, n, c, h * w, True, True, True, self.group)
            ops.release_plans()
            part = ops.pc_moments(x, n, c, h * w, True)
            mom_local, _ = ops.pc_combine(part, True)
            mom_ref, st_ref = ops.pc_combine(all_gather_records(mom_local, self.group), True)
            ok = ok and bool(torch.equal(st[:2], st_ref[:2])) and bool(torch.equal(mom[L.MOM_COUNT], mom_ref[L.MOM_COUNT]))
            ok = ok and bool(torch.allclose(st[2:4], st_ref[2:4], rtol=1e-5, atol=1e-6)) and bool(torch.allclose(mom[2:4], mom_ref[2:4], rtol=1e-6, atol=1e-3))
            ok = ok and bool(torch.isfinite(st[L.STAT_B]).all()) and bool((st[L.STAT_B] > 0).all())
        return self._all_agree(ok and self.healthy())

    def _unmap(self):
        for w in self.mapped:
            self.lib.cnnq_xrank_close(w)
        if self.own is not None:
            self.lib.cnnq_xrank_free(self.own)
        self.mapped, self.own = [], None

    def close(self):
        """Collective: unmap the peers' windows and free the own one once nobody uses them any more.  When launches of this
        exchange were captured into HIP graphs the windows stay mapped (a few MB, for the life of the process): a replay of
        such a graph after close() must not fault - it finds no peer, its waits expire and it reports through its status
        word like any other expired wait."""
        torch.cuda.synchronize()
        dist.barrier(group=self.group)
        if not self.captured:
            self._unmap()
        else:
            _RETIRED.append(self)                        # release_retired_exchanges() once the graphs are gone
        self.ok = False
        for k in [k for k, v in _XRANK.items() if v is self]:    # a closed exchange must not be handed out again
            del _XRANK[k]
        from . import ops
        ops.release_plans()                              # ... nor stay in a cached plan of the hot call


_XRANK = {}
_XRANK_MODE = None
_RECOVERY = 'raise'
_RETIRED = []       # closed exchanges whose launches live on in captured graphs: their windows stay mapped


def release_retired_exchanges():
    """Unmap / free the windows of closed exchanges that were kept because HIP graphs held launches of them.  Call it once
    those graphs have been destroyed (a replay afterwards would fault); local, synchronises the device."""
    torch.cuda.synchronize()
    while _RETIRED:
        _RETIRED.pop()._unmap()


def set_xrank_mode(mode):
    """'0' / '1' / 'auto' for this process (overrides CNNQ_XRANK; None: back to the environment).  A program that sets
    '1' or 'auto' takes on the recovery: agree over the group when a wait expired, disable_xrank, redo the work."""
    global _XRANK_MODE
    if mode not in (None, '0', '1', 'auto'):
        raise ValueError(mode)
    _XRANK_MODE = mode
    from . import ops
    ops.reload_switches()


def xrank_mode():
    """Which exchange the batch-sharded single-launch paths use (config 2's extrema; since round 6 the sums of configs 3 / 4 / 5).
    'auto' (DEFAULT since round 6): the in-launch exchange when the group has several ranks and one GPU per rank (backend nccl =
    RCCL), after it reproduced the collective's bits on every rank (XRankExchange.verify) - ranks that share a device (the gloo
    rigs of the tests) cannot count on their launches running together and keep the collective.  '0': the collective -
    statistics launch, RCCL all_gather on the compute stream, Q/DQ launch; host-side skew between the ranks is harmless there.
    '1': the in-launch exchange whenever a group exchanges (also ranks that share a GPU, also a forced 1-rank exchange).  From
    set_xrank_mode, else CNNQ_XRANK.  The recovery that makes 'auto' a safe default is xrank_checkpoint()."""
    import os
    m = _XRANK_MODE if _XRANK_MODE is not None else os.environ.get('CNNQ_XRANK', 'auto')
    return m if m in ('1', 'auto') else '0'


def xrank_exchange(group=None):
    """The process-wide XRankExchange of `group` when xrank_mode allows it and it verified against the collective
    path on every rank; else None (the collective).  World size 1 only with mode '1' under CNNQ_FORCE_EXCHANGE=1 (timing
    the protocol on a 1-GPU box)."""
    import os
    mode = xrank_mode()
    if mode == '0' or not (dist.is_available() and dist.is_initialized()):
        return None
    if world_size(group) == 1 and not (mode == '1' and forced_exchange()):
        return None
    if mode == 'auto' and dist.get_backend(group) != 'nccl' and os.environ.get('CNNQ_XRANK_SHARED_OK', '0') != '1':
        return None                                       # (CNNQ_XRANK_SHARED_OK=1: tests of the auto path on a one-GPU rig)
    key = tuple(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD))
    if key not in _XRANK:
        ex = XRankExchange(group)                       # collective; never raises for a local failure
        good = ex.ok and ex.verify()
        if not good:
            if rank(group) == 0:
                print('cnn_quantization_amd: in-launch cross-rank exchange unavailable or not verified (%s); using the '
                      'collective' % (ex.why or 'see other ranks',))
            ex.close()
        _XRANK[key] = ex if good else None
    return _XRANK[key]


def set_xrank_recovery(mode):
    """How an expired wait of the in-launch exchange surfaces.  'raise' (default): the exchange looks at its status word every
    CHECK_EVERY launches without synchronising and raises CnnqError on the rank that saw the expiry - like a collective's
    timeout, the job ends.  'checkpoint': the program calls xrank_checkpoint() at its own synchronisation points (every rank,
    together) and redoes the work since the last one when it returns False; nothing raises in between, so no rank leaves the
    lock-step of the collectives on its own.  harness/inference_sim.py uses 'checkpoint'."""
    global _RECOVERY
    if mode not in ('raise', 'checkpoint'):
        raise ValueError(mode)
    _RECOVERY = mode


def xrank_checkpoint(group=None):
    """The recovery of the in-launch exchange, for the program's own synchronisation points (the end of a forward pass, after
    the replays of a captured graph).  COLLECTIVE and synchronising: every rank of `group` calls it at the same point.  True:
    no wait for a peer has expired on any rank since the last checkpoint - everything computed since then stands.  False: some
    rank's wait expired (its outputs of that launch and of its later launches are NaN): the group's exchange has been closed on
    every rank - the collective serves the group from now on - and the caller redoes the work since the last checkpoint
    (harness/inference_sim.py re-runs the batch).  Without an exchange (mode '0', one rank, ranks sharing a GPU) it returns
    True without synchronising."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    key = tuple(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD))
    ex = _XRANK.get(key)
    if ex is None:
        return True
    ok = ex._all_agree(ex.healthy())
    if not ok:
        if rank(group) == 0:
            print('cnn_quantization_amd: a wait of the in-launch exchange expired on some rank; the group continues on the collective')
        disable_xrank(group)
    return ok


def disable_xrank(group=None):
    """Every rank of `group` calls this together (after a wait for a peer expired, say): the group's in-launch exchange is
    closed and the collective path serves the group from now on."""
    key = tuple(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD))
    ex = _XRANK.get(key)
    if ex is not None:
        ex.close()
    _XRANK[key] = None
    from . import ops
    ops.release_plans()                                  # the hot call's cached plans hold the exchange


def gather_counts(n, device, group=None):
    """The integer `n` of every rank as a host list in rank order (one tiny all_gather + a host sync; used only by
    the per-tensor path, whose tables have one row per local sample and shards may differ by a sample)."""
    t = torch.tensor([[int(n)]], dtype=torch.int64, device=device)
    return [int(v) for v in collective_all_gather(t, group).flatten().tolist()]


def merge_row_minmax(stats, rows, avg_over_batch, group=None):
    """Per-sample MIN/MAX rows of every rank's shard -> one table whose first two rows list all samples of the
    global batch in rank order, ready for cnnq_pt_setup (which takes their batch mean, or their extrema).
    Shards may hold different numbers of samples (shard_batch: sizes differ by at most one): the row counts
    travel first, the tables are padded to the largest shard for the all_gather and un-padded afterwards."""
    w = world_size(group)
    counts = gather_counts(rows, stats.device, group)
    rmax = max(counts)
    local = torch.zeros((2, rmax), dtype=stats.dtype, device=stats.device)
    local[:, :rows] = stats[:2, :rows]
    allr = all_gather_records(local, group)                     # [W, 2, rmax]
    merged = torch.zeros((stats.shape[0], sum(counts)), dtype=stats.dtype, device=stats.device)
    merged[:2] = torch.cat([allr[r, :, :counts[r]] for r in range(w)], dim=1)
    return merged


def all_reduce_sum_(t, group=None):
    """In-place integer sum (code histograms for the global entropy)."""
    if world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
