"""RCCL called directly (ctypes on the librccl.so that PyTorch-ROCm already has loaded) for the one collective of
the hot path: the all_gather of the small per-channel records between the statistics and the Q/DQ pass.

Why not torch.distributed for it: `dist.all_gather_into_tensor` costs ~15-20 us of HOST time per call and runs on
the process group's own stream (two event hand-offs with the compute stream).  At the batch-64 shard a tensor's whole
chain takes ~35 us, so 53 such calls per step made the multi-GPU step host-bound (tools/rccl_latency.py,
DESIGN.md section 6).  `ncclAllGather` enqueued on the caller's stream is one ~3 us ctypes call and no stream hop.

torch.distributed stays the control plane: it carries the unique id at set-up, is the reference the first exchanges
are verified against, and is the fallback whenever this path is unavailable (library not found, set-up failed or
timed out, verification mismatch) - the decision is taken collectively so every rank uses the same path."""
import ctypes
import os
import threading

import torch
import torch.distributed as dist

NCCL_UINT8 = 1
SETUP_TIMEOUT_S = 120.


class _UniqueId(ctypes.Structure):
    _fields_ = [('internal', ctypes.c_char * 128)]


def _load():
    path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
    lib = ctypes.CDLL(path if os.path.exists(path) else 'librccl.so')
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
    lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                  ctypes.c_void_p]
    lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    lib.ncclCommCuDevice.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    lib.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllGather, lib.ncclCommDestroy, lib.ncclCommCuDevice,
              lib.ncclCommCount):
        f.restype = ctypes.c_int
    return lib


class DirectComm:
    """An RCCL communicator over the ranks of `group`, created collectively (every rank must construct it).
    `ok` is the group-wide verdict; when False the caller keeps torch.distributed."""

    def __init__(self, group=None):
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.comm, self.lib, self.why = ctypes.c_void_p(), None, ''
        local_ok = True
        try:
            self.lib = _load()
        except (OSError, AttributeError) as e:        # library not found, or a symbol this module binds is missing
            local_ok, self.why = False, 'librccl.so: %s' % e
        uid = _UniqueId()
        if local_ok and self.rank == 0:
            rc = self.lib.ncclGetUniqueId(ctypes.byref(uid))
            if rc != 0:
                local_ok, self.why = False, 'ncclGetUniqueId: %d' % rc
        box = [ctypes.string_at(ctypes.byref(uid), 128) if (self.rank == 0 and local_ok) else b'']   # raw: the id holds NULs
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if len(box[0]) != 128:
            local_ok, self.why = False, self.why or 'no unique id from rank 0'
        if self._all_agree(local_ok):                # nobody enters the blocking set-up unless everybody can
            ctypes.memmove(ctypes.byref(uid), box[0], 128)
            err = []
            # the HIP current device is a property of the HOST THREAD and a new thread starts on device 0:
            # without this every rank >= 1 of a one-process-per-GPU job would create its communicator on GPU 0
            dev_index = torch.cuda.current_device()
            self.device_index = dev_index

            def init():
                torch.cuda.set_device(dev_index)
                err.append(self.lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank))
            th = threading.Thread(target=init, daemon=True)
            th.start()
            th.join(SETUP_TIMEOUT_S)
            if th.is_alive() or not err or err[0] != 0:
                local_ok, self.why = False, 'ncclCommInitRank %s' % ('timed out' if th.is_alive() else err)
            if local_ok:
                # the communicator must live on THIS rank's device and span the whole group
                d, n = ctypes.c_int(-1), ctypes.c_int(-1)
                rc1 = self.lib.ncclCommCuDevice(self.comm, ctypes.byref(d))
                rc2 = self.lib.ncclCommCount(self.comm, ctypes.byref(n))
                if rc1 != 0 or rc2 != 0 or d.value != dev_index or n.value != self.world:
                    local_ok, self.why = False, 'communicator on device %d with %d ranks, expected device %d with %d ranks' % (
                        d.value, n.value, dev_index, self.world)
        else:
            local_ok = False
        self.ok = self._all_agree(local_ok) and self._verify()
        if not self.ok and not self.why:
            self.why = 'set-up or verification failed on some rank'

    def _all_agree(self, flag):
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=torch.device('cuda', torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    def all_gather(self, rec, out):
        """rec (contiguous device tensor) of every rank -> out [W, ...] in rank order, enqueued on the current stream."""
        st = ctypes.c_void_p(torch.cuda.current_stream(rec.device).cuda_stream)
        rc = self.lib.ncclAllGather(ctypes.c_void_p(rec.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                    rec.numel() * rec.element_size(), NCCL_UINT8, self.comm, st)
        if rc != 0:
            raise RuntimeError('ncclAllGather failed: %s' % self.lib.ncclGetErrorString(rc).decode())
        return out

    def all_gather_raw(self, src, dst, nbytes, stream):
        """The same on raw device addresses and a raw stream handle (the hot path's cached plan: ~2 us of host time)."""
        rc = self.lib.ncclAllGather(src, dst, nbytes, NCCL_UINT8, self.comm, stream)
        if rc != 0:
            raise RuntimeError('ncclAllGather failed: %s' % self.lib.ncclGetErrorString(rc).decode())

    def _verify(self, rounds=8):
        """A few records through both paths; True iff every rank sees identical results."""
        dev = torch.device('cuda', torch.cuda.current_device())
        g = torch.Generator(device=dev).manual_seed(4321 + self.rank)
        ok = True
        for i in range(rounds):
            rec = torch.randn((2, (1, 64, 2048, 4096)[i % 4]), generator=g, device=dev)
            if i % 2:
                rec = rec.double()
            ref = torch.empty((self.world,) + tuple(rec.shape), dtype=rec.dtype, device=dev)
            dist.all_gather_into_tensor(ref.view(-1), rec.view(-1), group=self.group)
            got = self.all_gather(rec, torch.empty_like(ref))
            torch.cuda.synchronize()
            ok = ok and bool(torch.equal(ref, got))
        return self._all_agree(ok)

    def close(self):
        if self.comm:
            torch.cuda.synchronize()
            self.lib.ncclCommDestroy(self.comm)
            self.comm = ctypes.c_void_p()


_COMMS = {}


def direct_comm(group=None):
    """The process-wide DirectComm of `group` (created at first use, collectively), or None: backend is not nccl,
    CNNQ_DIRECT_RCCL=0, or the set-up / verification failed somewhere in the group."""
    if os.environ.get('CNNQ_DIRECT_RCCL', '1') == '0' or not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_backend(group) != 'nccl':
        return None
    key = tuple(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD))
    if key not in _COMMS:
        c = DirectComm(group)
        if not c.ok and dist.get_rank(group) == 0:
            print('cnn_quantization_amd: direct RCCL path unavailable (%s); using torch.distributed' % (c.why or 'see other ranks'))
        _COMMS[key] = c if c.ok else None
    return _COMMS[key]


def comm_ranks(group=None):
    """Ranks of the direct communicator of `group` as RCCL itself reports them (ncclCommCount), 0 when the direct
    path is not in use - reported by bench.py as `rccl_ranks`."""
    c = direct_comm(group)
    if c is None:
        return 0
    n = ctypes.c_int(0)
    return n.value if c.lib.ncclCommCount(c.comm, ctypes.byref(n)) == 0 else 0


def close_all():
    """Destroy every direct communicator; cached exchange plans that hold one are dropped with it."""
    for c in _COMMS.values():
        if c is not None:
            c.close()
    _COMMS.clear()
    try:
        from . import ops
        ops.release_plans()
    except ImportError:
        pass
