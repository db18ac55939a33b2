"""Data-parallel sharding of the hot path across the GPUs of one node (one process per GPU, RCCL over xGMI).

The path is embarrassingly parallel over clips: rank r of R owns the contiguous rows
[r*B/R, (r+1)*B/R) of the batch and a full replica of the (<= 3 MB) packed weights.  There are exactly two
collectives, both on torch.distributed's "nccl" backend (= RCCL on ROCm):
  * a 256-byte in-place MAX all-reduce of the 64 batch-amax slots between the two kernels (the slots are armed with -inf,
    so there is no other kernel between the two HIP kernels), so the top_db floor is the one the
    reference computes for the WHOLE batch (models/uit.py:307 via torchaudio's packed-channel reshape) and the
    result does not depend on R;
  * one all-gather of the [B/R, outputdim] fp32 outputs into [B, outputdim] on every rank (what evaluate.py's
    metric accumulation consumes).
Nothing else crosses GPUs; there is no collective inside either kernel.
"""
import torch
import torch.distributed as dist


def shard_range(B, rank, world):
    """Contiguous, equal shards (the all-gather needs equal sizes): B must divide by world."""
    if B % world:
        raise ValueError(f"global batch {B} must be divisible by the number of ranks {world}")
    per = B // world
    return rank * per, (rank + 1) * per


def amax_allreduce(slots, group=None):
    """In-place MAX all-reduce of the 64 batch-amax slots (256 bytes): ONE collective on the device slots themselves, nothing
    else between the two HIP kernels.  Untouched slots hold -inf (the front-end arms them so), which MAX ignores; the encoder
    takes the max over the 64 slots itself."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(slots, op=dist.ReduceOp.MAX, group=group)
    return slots


def amax_allreduce_async(slots, group=None):
    """The same collective, asynchronous on RCCL's stream.  The caller may enqueue unrelated kernels (the next batch's
    front-end on another replica) before amax_allreduce_finish."""
    work = None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        work = dist.all_reduce(slots, op=dist.ReduceOp.MAX, group=group, async_op=True)
    return slots, work


def amax_allreduce_finish(pending):
    """Second half: the current stream waits for the collective (no host sync, no extra kernel)."""
    slots, work = pending
    if work is not None:
        work.wait()
    return slots


def gather_outputs(local, group=None):
    """[B/R, C] on every rank -> [B, C] on every rank (rank order == row order)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


class GatherHandle:
    """Result of an asynchronous all-gather: `.wait()` makes the current stream wait for it and returns [B, C]."""

    def __init__(self, work, out, keep):
        self.work, self.out, self._keep = work, out, keep      # `keep` pins the send buffer until completion

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.out


def gather_outputs_async(local, group=None):
    """Like gather_outputs, but returns immediately with a GatherHandle: the collective runs on RCCL's own stream, so
    the kernels of the NEXT batch (enqueued on the caller's stream) overlap with it.  At 8 ranks x 8192 clips the
    gather moves 141 MB per step -- about half a compute step if it were serialised."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return GatherHandle(None, local, None)
    world = dist.get_world_size(group)
    local = local.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    work = dist.all_gather_into_tensor(out, local, group=group, async_op=True)
    return GatherHandle(work, out, local)


class ShardedUiT:
    """model: any callable `model(x, amax_sync=fn) -> probs` (uit_mobile_amd.models.UITBase on a HIP device)."""

    def __init__(self, model, group=None, global_floor=True):
        self.model, self.group, self.global_floor = model, group, global_floor

    def __call__(self, local_x, async_gather=False):
        sync = (lambda slots: amax_allreduce(slots, self.group)) if self.global_floor else None
        probs = self.model(local_x, amax_sync=sync)
        if async_gather:
            return gather_outputs_async(probs, self.group)      # -> GatherHandle, overlap with the next batch
        return gather_outputs(probs, self.group)
