"""Multi-GPU layer: one process per GPU (torchrun), activations sharded along the batch, and ONE
small exchange per statistics pass over RCCL/xGMI (torch.distributed backend "nccl" is RCCL on
ROCm; the CPU tests drive the same code over gloo).

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


def rank(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    return dist.get_rank(group)


def shard_batch(n, rank_, world):
    """[n0, n1) of the batch owned by `rank_`: contiguous, sizes differ by at most one."""
    base, rem = divmod(n, world)
    n0 = rank_ * base + min(rank_, rem)
    return n0, n0 + base + (1 if rank_ < rem else 0)


def all_gather_records(rec, group=None):
    """rec [K, C] on every rank -> [W, K, C] in rank order (the G axis cnnq_pc_combine merges)."""
    w = world_size(group)
    rec = rec.contiguous()
    out = torch.empty((w,) + tuple(rec.shape), dtype=rec.dtype, device=rec.device)
    if w == 1:
        out[0].copy_(rec)
        return out
    try:
        dist.all_gather_into_tensor(out.view(-1), rec.view(-1), group=group)   # flat: same on RCCL and gloo
    except RuntimeError:
        # gloo with device tensors (single-GPU test rigs): list form
        dist.all_gather([out[i] for i in range(w)], rec, group=group)
    return out


def all_gather_records_async(rec, group=None):
    """Non-blocking all_gather_records: returns (out [W, K, C], work).  The collective runs on the backend's
    own stream (RCCL) while the caller keeps enqueuing kernels; `work.wait()` orders the caller's stream
    behind it."""
    w = world_size(group)
    rec = rec.contiguous()
    out = torch.empty((w,) + tuple(rec.shape), dtype=rec.dtype, device=rec.device)
    try:
        work = dist.all_gather_into_tensor(out.view(-1), rec.view(-1), group=group, async_op=True)
    except RuntimeError:
        work = dist.all_gather([out[i] for i in range(w)], rec, group=group, async_op=True)
    return out, work


def merge_row_minmax(stats, rows, avg_over_batch, group=None):
    """Per-sample MIN/MAX rows of every rank's shard -> one table whose first two rows list all
    samples of the global batch (equal shard sizes), ready for cnnq_pt_setup."""
    w = world_size(group)
    local = stats[:2, :rows].contiguous()                       # [2, rows]
    allr = all_gather_records(local, group)                     # [W, 2, rows]
    merged = torch.zeros((stats.shape[0], w * rows), dtype=stats.dtype, device=stats.device)
    merged[:2] = allr.permute(1, 0, 2).reshape(2, w * rows)
    return merged


def all_reduce_sum_(t, group=None):
    """In-place integer sum (code histograms for the global entropy)."""
    if world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
