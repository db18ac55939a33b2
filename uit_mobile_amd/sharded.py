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
    peer buffers (PeerGather) that uses no CU.  The payload can be cut before it crosses xGMI: float16 probabilities, the
    first n classes only, or the device-side top-k only (ShardedUiT).
Nothing else crosses GPUs; there is no collective inside either kernel.
"""
import torch
import torch.distributed as dist


_HOST_STAGED = True      # collectives over a host-side backend are staged through the host by this module (False: bench.py's lab switch
                         # --gloo-device-tensors, the path as it was before round 6's hardening -- tools/wd_repeat.sh)


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
        if _HOST_STAGED and not _backend_is_device(group):
            # host-side backend (gloo: several ranks on ONE GPU, tests only): staged through the host by this module, synchronously, in
            # stream order -- not handed to gloo as device tensors (round 6: one run in ~45 of the 8-ranks-on-one-GPU scenario failed its
            # parity check, never reproduced; the path now depends on nothing but `.cpu()` / `.copy_()` stream semantics)
            return amax_allreduce_host(slots, group), None
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
    if _HOST_STAGED and local.is_cuda and not _backend_is_device(group):
        return gather_outputs_async(local, group).wait()       # host-side backend: staged through the host (see there)
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
    if _HOST_STAGED and not _backend_is_device(group):
        # host-side backend (gloo, tests only): `.cpu()` waits for the producing kernels, the collective runs on host tensors, the result
        # returns with a stream-ordered copy -- synchronous, and free of any assumption about a host backend's handling of device tensors
        h = local.detach().cpu()
        out_h = torch.empty((world * h.shape[0],) + tuple(h.shape[1:]), dtype=h.dtype)
        dist.all_gather_into_tensor(out_h, h, group=group)
        return GatherHandle(None, out_h.to(local.device), None)
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


class PeerGatherUnavailable(RuntimeError):
    """The IPC precondition of PeerGather does not hold on this node (raised on EVERY rank, so callers can fall back together)."""


class PeerGather:
    """Copy-engine all-gather over xGMI: every rank copies its [rows, C] block straight INTO
    every peer's output buffer with device-to-device copies (hipMemcpyPeerAsync through IPC-mapped buffers), so no RCCL
    gather kernel competes for the CUs that the two persistent HIP kernels hold.

    Fences (one 4-byte all-reduce per gather, enqueued on the copy stream BEHIND the copies and behind everything the caller's
    stream held when gather_async was called):
      * arrival: when the fence of gather s has completed on a rank, every peer's block of gather s has landed in that rank's
        buffer (a rank contributes only after its own copies);
      * consumption: the copies of gather s+1 wait for the fence of gather s, and a rank contributes to that fence only after
        the work its caller had enqueued before gather_async(s) -- so a peer overwrites buffer b only after this rank has
        finished every read of b that was enqueued before its LATEST-BUT-ONE gather_async call.
    Contract: `nbuf` buffers rotate; the reads of a result must be enqueued on the current stream before the (nbuf-1)-th
    following gather_async call.  nbuf = 3 (default) covers the pipelined pattern "wait handle N after enqueuing gather N+1";
    nbuf = 2 is enough when every result is waited and consumed before the next gather.

    Buffers are allocated once and exchanged once through torch's IPC handles (hipIpcGetMemHandle / dmabuf; this pool's driver needs
    HSA_ENABLE_IPC_MODE_LEGACY=0).  The precondition is CHECKED by doing it: if any rank cannot export or map a buffer, every rank raises
    PeerGatherUnavailable (ShardedUiT then falls back to the RCCL gather with a warning).  All ranks must live on one node."""

    def __init__(self, rows, cols, device, group=None, dtype=torch.float32, nbuf=3):
        from torch.multiprocessing.reductions import reduce_tensor
        if nbuf < 2:
            raise ValueError("PeerGather needs nbuf >= 2")
        self.group, self.rows, self.cols = group, rows, cols
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device(device)
        self.out = [torch.empty((self.world * rows, cols), device=self.device, dtype=dtype) for _ in range(nbuf)]
        self.fence = torch.zeros(1, device=self.device, dtype=torch.int32)
        self.copy_stream = torch.cuda.Stream(self.device)
        self.step = 0
        self._last_fence = None                     # work handle of the previous gather's fence (device backends)
        self.peers = [[None] * nbuf for _ in range(self.world)]
        if self.world > 1:
            import os
            # the actual export / map decides; the environment only explains a failure (drivers with legacy IPC work without the variable)
            hint = "" if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "") == "0" else \
                " (HSA_ENABLE_IPC_MODE_LEGACY=0 is not set: a host driver that only supports dmabuf IPC needs it)"
            err, mine = None, None
            try:
                mine = [reduce_tensor(t) for t in self.out]
            except Exception as e:                  # hipIpcGetMemHandle refused
                err = f"cannot export the output buffers: {e}{hint}"
            everyone = [None] * self.world
            dist.all_gather_object(everyone, (mine, err), group=group)
            if all(e is None for _, e in everyone):
                try:
                    for r, (handles, _) in enumerate(everyone):
                        if r != self.rank:
                            self.peers[r] = [fn(*args) for fn, args in handles]     # peer buffers mapped into this process
                except Exception as e:              # hipIpcOpenMemHandle refused
                    err = f"cannot map a peer's buffers: {e}{hint}"
            verdict = [None] * self.world
            dist.all_gather_object(verdict, err, group=group)                       # every rank takes the same decision
            bad = [(r, e) for r, e in enumerate(verdict) if e is not None] + [(r, e) for r, (_, e) in enumerate(everyone) if e is not None]
            if bad:
                raise PeerGatherUnavailable("PeerGather precondition failed: " + "; ".join(f"rank {r}: {e}" for r, e in sorted(set(bad))))
            self._keep = everyone

    def gather_async(self, local):
        """Enqueue the fan-out of `local` [rows, C] (ready on the current stream) on the copy stream; returns a handle whose
        wait() makes the current stream wait for the arrival fence and returns this rank's full [world*rows, C] buffer."""
        b = self.step % len(self.out)
        self.step += 1
        lo = self.rank * self.rows
        cur = torch.cuda.current_stream(self.device)
        device_backend = self.world > 1 and _backend_is_device(self.group)
        self.copy_stream.wait_stream(cur)           # `local` is ready; the caller's earlier reads of older results are ordered before our fence
        with torch.cuda.stream(self.copy_stream):
            if device_backend and self._last_fence is not None:
                self._last_fence.wait()             # consumption fence: the previous gather's all-reduce has completed on this stream
            self.out[b][lo:lo + self.rows].copy_(local, non_blocking=True)
            for r in range(self.world):
                if r != self.rank:
                    self.peers[r][b][lo:lo + self.rows].copy_(local, non_blocking=True)   # device-to-device over xGMI
            work, done = None, None
            if self.world > 1:
                if device_backend:
                    work = dist.all_reduce(self.fence, group=self.group, async_op=True)   # fence behind the copies
                    self._last_fence = work
                else:                                       # host-side backend (tests): drain the copies, then rendezvous
                    self.copy_stream.synchronize()
                    dist.barrier(group=self.group)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        return _PeerHandle(work, done, self.out[b], local)

    def gather(self, local):
        return self.gather_async(local).wait()

    def self_test(self):
        """One gather of a known pattern through every buffer, verified on EVERY rank: rank r sends rows filled with r + 1; each rank
        checks that block r of its result holds r + 1, and the verdicts are MIN-all-reduced, so all ranks agree.  Catches what the
        export / map check at construction cannot: peer copies that land nowhere (no P2P route) or in the wrong place.  Raises
        PeerGatherUnavailable on all ranks together.  Costs `nbuf` small gathers; call it once at set-up."""
        if self.world == 1:
            return True
        ok = True
        try:
            for _ in range(len(self.out)):
                send = torch.full((self.rows, self.cols), float(self.rank + 1), device=self.device).to(self.out[0].dtype)
                got = self.gather_async(send).wait()
                torch.cuda.current_stream(self.device).synchronize()
                want = torch.arange(1, self.world + 1, device=self.device).repeat_interleave(self.rows).to(got.dtype)
                ok = ok and bool((got == want[:, None]).all())
        except Exception:
            ok = False
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if _backend_is_device(self.group) else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()) != 1:
            raise PeerGatherUnavailable("PeerGather self-test failed: a peer copy did not arrive where it should (no usable P2P route?)")
        return True


class _PeerHandle:
    def __init__(self, work, done, out, keep):
        self.work, self.done, self.out, self._keep = work, done, out, keep

    def wait(self):
        if self.work is not None:
            self.work.wait()                        # idempotent: PeerGather itself may wait on the same work for the next gather
            self.work = None
        if self.done is not None:
            torch.cuda.current_stream(self.out.device).wait_event(self.done)
            self.done = None
        return self.out


class ShardedUiT:
    """model: any callable `model(x, amax_sync=fn) -> probs` (uit_mobile_amd.models.UITBase on a HIP device).

    gather: 'rccl' (default: torch.distributed all_gather_into_tensor on the nccl = RCCL backend) or 'peer' (PeerGather:
    copy-engine fan-out, no collective kernel; falls back to 'rccl' with a warning when its IPC precondition fails).
    What is gathered (the payload per clip decides the xGMI ingress per rank: fp32 [537] = 2 148 B):
      * gather_dtype=None: the fp32 probabilities (bit-for-bit R-invariant);  torch.float16: converted before the collective
        (1 074 B per clip; |error| <= 2.5e-4 on a probability, what evaluate.py's metrics see) and returned as float16;
      * slice_classes=n: only the first n classes (evaluate.py:92 consumes `[..., :527]`);
      * topk=k: only `output.topk(k)` (inference.py:57) computed on the device by uitk_topk BEFORE the collective -- 8 k bytes
        per clip; the call then returns (values [B,k] float32, indices [B,k] int64).
    R-invariance: results do not depend on the rank count -- bit for bit -- as long as every shard starts on an even row
    (rows-per-rank even): a clip's arithmetic depends on the PARITY of its row within the call (the two clips of a wave are
    pooled in mirrored lane order), on nothing else."""

    def __init__(self, model, group=None, global_floor=True, gather="rccl", gather_dtype=None, slice_classes=None, topk=None,
                 topk_fn=None, peer_nbuf=3):
        if gather not in ("rccl", "peer"):
            raise ValueError(f"Unknown gather ({gather})")
        self.peer_nbuf = int(peer_nbuf)             # PeerGather's rotating buffers: 3 = "wait handle n after enqueuing gather n+1"
        if gather_dtype not in (None, torch.float32, torch.float16):
            raise ValueError("gather_dtype must be None / torch.float32 / torch.float16")
        if topk is not None and (gather_dtype is torch.float16 or slice_classes is not None):
            raise ValueError("topk replaces the probability payload: do not combine it with gather_dtype / slice_classes")
        self.model, self.group, self.global_floor, self.gather = model, group, global_floor, gather
        self.gather_dtype = None if gather_dtype is torch.float32 else gather_dtype
        self.slice_classes, self.topk = slice_classes, topk
        self._topk_fn = topk_fn                     # tests inject a host function; the product uses postprocess.topk (uitk_topk)
        self._peer = None

    def _amax_sync(self, slots):
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        if multi and not _backend_is_device(self.group):
            return amax_allreduce_host(slots, self.group)
        return amax_allreduce(slots, self.group)

    def payload_bytes_per_clip(self, outputdim):
        """bytes each rank RECEIVES per remote clip (sizing the xGMI ingress: DESIGN.md section 5)"""
        if self.topk is not None:
            return 8 * self.topk
        n = outputdim if self.slice_classes is None else min(outputdim, self.slice_classes)
        return n * (2 if self.gather_dtype is torch.float16 else 4)

    def _payload(self, probs):
        if self.topk is not None:
            fn = self._topk_fn
            if fn is None:
                from .postprocess import topk as fn
            vals, idx = fn(probs, self.topk)
            # one [rows, 2k] int32 tensor: values bit-cast next to the indices -> ONE collective, 8 k bytes per clip
            return torch.cat([vals.contiguous().view(torch.int32), idx.to(torch.int32)], 1)
        if self.slice_classes is not None:
            probs = probs[:, :self.slice_classes]
        if self.gather_dtype is torch.float16:
            probs = probs.to(torch.float16)
        return probs.contiguous()

    def _unpack(self, gathered):
        if self.topk is None:
            return gathered
        k = self.topk
        return gathered[:, :k].contiguous().view(torch.float32), gathered[:, k:].to(torch.int64)

    def __call__(self, local_x, async_gather=False):
        probs = self.model(local_x, amax_sync=self._amax_sync if self.global_floor else None)
        send = self._payload(probs)
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        if self.gather == "peer" and multi:
            if self._peer is None or self._peer.rows != send.shape[0] or self._peer.cols != send.shape[1] or self._peer.out[0].dtype != send.dtype:
                try:
                    self._peer = PeerGather(send.shape[0], send.shape[1], send.device, self.group, send.dtype, nbuf=self.peer_nbuf)
                    self._peer.self_test()
                except PeerGatherUnavailable as e:
                    import warnings
                    warnings.warn(f"{e} -- falling back to gather='rccl'")
                    self.gather, self._peer = "rccl", None
        if self.gather == "peer" and multi:
            h = self._peer.gather_async(send)
        else:
            h = gather_outputs_async(send, self.group)      # -> GatherHandle, overlap with the next batch
        h = _Unpacked(h, self._unpack) if self.topk is not None else h
        return h if async_gather else h.wait()


class _Unpacked:
    """A gather handle whose wait() splits the packed top-k payload back into (values, indices)."""

    def __init__(self, handle, fn):
        self.handle, self.fn = handle, fn

    def wait(self):
        return self.fn(self.handle.wait())
