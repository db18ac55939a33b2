"""Data-parallel sharding of the hot path across the GPUs of one node (one process per GPU, RCCL over xGMI).

The path is embarrassingly parallel over clips: rank r of R owns the contiguous rows
[r*B/R, (r+1)*B/R) of the batch and a full replica of the (<= 3 MB) packed weights.  There are exactly two
collectives, both on torch.distributed's "nccl" backend (= RCCL on ROCm):
  * a 256-byte in-place MAX all-reduce of the 64 batch-amax slots between the two kernels (the slots are armed with -inf,
    so there is no other kernel between the two HIP kernels), so the top_db floor is the one the
    reference computes for the WHOLE batch (models/uit.py:307 via torchaudio's packed-channel reshape) and the
    result does not depend on R;
  * one all-gather of the [B/R, outputdim] fp32 outputs into [B, outputdim] on every rank (what evaluate.py's
    metric accumulation consumes): RCCL's all_gather_into_tensor, or -- selectable -- a copy-engine fan-out into IPC-mapped
    peer buffers (PeerGather) that uses no CU.
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


def _backend_is_device(group=None):
    """True when the group's collectives run on device tensors in stream order (nccl = RCCL); gloo works on host tensors."""
    try:
        return dist.get_backend(group) == "nccl"
    except Exception:
        return False


def amax_allreduce_host(slots, group=None):
    """The same MAX all-reduce of the 64 slots staged through the host -- for groups whose backend cannot take device tensors
    (gloo: the two-processes-on-one-GPU tests).  Synchronises the stream; never used with RCCL."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        h = slots.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
        slots.copy_(h)
    return slots


class PeerGather:
    """Copy-engine all-gather over xGMI: every rank copies its [rows, C] block straight INTO
    every peer's output buffer with device-to-device copies (hipMemcpyPeerAsync through IPC-mapped buffers), so no RCCL
    gather kernel competes for the CUs that the two persistent HIP kernels hold.  Completion fence: a 4-byte all-reduce
    enqueued behind the copies -- a rank's contribution is stream-ordered after its own copies, so when the all-reduce has
    completed on a rank, every peer's block has landed in that rank's buffer.

    Buffers are allocated once (`nbuf` of them, rotated per call: a result stays valid until `nbuf` further gathers) and
    exchanged once through torch's IPC handles (hipIpcGetMemHandle / dmabuf; needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this
    driver).  All ranks must live on one node and see the same device numbering."""

    def __init__(self, rows, cols, device, group=None, dtype=torch.float32, nbuf=2):
        from torch.multiprocessing.reductions import reduce_tensor
        self.group, self.rows, self.cols = group, rows, cols
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device(device)
        self.out = [torch.empty((self.world * rows, cols), device=self.device, dtype=dtype) for _ in range(nbuf)]
        self.fence = torch.zeros(1, device=self.device, dtype=torch.int32)
        self.copy_stream = torch.cuda.Stream(self.device)
        self.step = 0
        self.peers = [[None] * nbuf for _ in range(self.world)]
        if self.world > 1:
            mine = [reduce_tensor(t) for t in self.out]
            everyone = [None] * self.world
            dist.all_gather_object(everyone, mine, group=group)
            for r, handles in enumerate(everyone):
                if r != self.rank:
                    self.peers[r] = [fn(*args) for fn, args in handles]     # peer buffers mapped into this process
            self._keep = everyone

    def gather_async(self, local):
        """Enqueue the fan-out of `local` [rows, C] (ready on the current stream) on the copy stream; returns a GatherHandle
        whose wait() makes the current stream wait for the fence and returns this rank's full [world*rows, C] buffer."""
        b = self.step % len(self.out)
        self.step += 1
        lo = self.rank * self.rows
        cur = torch.cuda.current_stream(self.device)
        self.copy_stream.wait_stream(cur)
        with torch.cuda.stream(self.copy_stream):
            self.out[b][lo:lo + self.rows].copy_(local, non_blocking=True)
            for r in range(self.world):
                if r != self.rank:
                    self.peers[r][b][lo:lo + self.rows].copy_(local, non_blocking=True)   # device-to-device over xGMI
            work, done = None, None
            if self.world > 1:
                if _backend_is_device(self.group):
                    work = dist.all_reduce(self.fence, group=self.group, async_op=True)   # fence behind the copies
                else:                                       # host-side backend (tests): drain the copies, then rendezvous
                    self.copy_stream.synchronize()
                    dist.barrier(group=self.group)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        return _PeerHandle(work, done, self.out[b], local)

    def gather(self, local):
        return self.gather_async(local).wait()


class _PeerHandle:
    def __init__(self, work, done, out, keep):
        self.work, self.done, self.out, self._keep = work, done, out, keep

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.done is not None:
            torch.cuda.current_stream(self.out.device).wait_event(self.done)
            self.done = None
        return self.out


class ShardedUiT:
    """model: any callable `model(x, amax_sync=fn) -> probs` (uit_mobile_amd.models.UITBase on a HIP device).

    gather: 'rccl' (default: torch.distributed all_gather_into_tensor on the nccl = RCCL backend) or 'peer' (PeerGather:
    copy-engine fan-out, no collective kernel).  R-invariance: results do not depend on the rank count -- bit for bit -- as
    long as every shard starts on an even row (rows-per-rank even): a clip's arithmetic depends on the PARITY of its row
    within the call (the two clips of a wave are pooled in mirrored lane order), on nothing else."""

    def __init__(self, model, group=None, global_floor=True, gather="rccl"):
        if gather not in ("rccl", "peer"):
            raise ValueError(f"Unknown gather ({gather})")
        self.model, self.group, self.global_floor, self.gather = model, group, global_floor, gather
        self._peer = None

    def _amax_sync(self, slots):
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        if multi and not _backend_is_device(self.group):
            return amax_allreduce_host(slots, self.group)
        return amax_allreduce(slots, self.group)

    def __call__(self, local_x, async_gather=False):
        probs = self.model(local_x, amax_sync=self._amax_sync if self.global_floor else None)
        if self.gather == "peer" and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            if self._peer is None or self._peer.rows != probs.shape[0] or self._peer.cols != probs.shape[1]:
                self._peer = PeerGather(probs.shape[0], probs.shape[1], probs.device, self.group, probs.dtype)
            h = self._peer.gather_async(probs)
            return h if async_gather else h.wait()
        if async_gather:
            return gather_outputs_async(probs, self.group)      # -> GatherHandle, overlap with the next batch
        return gather_outputs(probs, self.group)
