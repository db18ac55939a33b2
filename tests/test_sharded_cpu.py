"""N>1 host logic on CPU: world_size-2 gloo processes drive the sharding wrapper (shard ranges, amax MAX
all-reduce semantics incl. NaN = untouched slot, all-gather row order) around a stand-in model function.
The HIP model itself cannot run here; what is under test is the distributed plumbing in uit_mobile_amd/sharded.py."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from uit_mobile_amd.sharded import (ShardedUiT, amax_allreduce, amax_allreduce_async, amax_allreduce_finish, gather_outputs,
                                    gather_outputs_async, shard_range)


def test_shard_range():
    assert [shard_range(8, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)


def test_amax_allreduce_single_process_is_identity_on_armed_slots():
    """Slots are armed with -inf (no NaN): the single-rank path leaves them alone, and their max is the batch amax."""
    s = torch.full((64,), float("-inf"))
    s[3], s[40] = -12.5, 7.25
    before = s.clone()
    assert amax_allreduce(s) is s and torch.equal(s, before)
    assert float(s.max()) == 7.25


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, C = 8, 5
        full = torch.arange(B * 4, dtype=torch.float32).reshape(B, 4)
        lo, hi = shard_range(B, rank, world)

        def fake_model(x, amax_sync=None):
            slots = torch.full((64,), float("-inf"))          # armed, as the front-end kernel leaves untouched slots
            slots[rank] = float(x.max())                      # this shard's "amax"
            amax_sync(slots)                                  # in-place MAX all-reduce of the 64 slots
            # output depends on the (now global) amax = max over the slots, like the encoder's top_db floor does
            return torch.cat([x, slots.max().reshape(1, 1).expand(x.shape[0], 1)], 1)

        out = ShardedUiT(fake_model)(full[lo:hi])
        expect = torch.cat([full, torch.full((B, 1), float(full.max()))], 1)
        ok = out.shape == (B, C) and torch.equal(out, expect)
        g = gather_outputs(torch.full((2, 3), float(rank)))
        ok = ok and torch.equal(g, torch.tensor([[0.] * 3] * 2 + [[1.] * 3] * 2))
        # asynchronous gathers: several in flight, waited out of band (what bench.py does to overlap with compute)
        hs = [gather_outputs_async(torch.full((2, 2), float(10 * i + rank))) for i in range(3)]
        hs.append(ShardedUiT(fake_model)(full[lo:hi], async_gather=True))
        for i, h in enumerate(hs[:3]):
            ok = ok and torch.equal(h.wait(), torch.tensor([[10. * i] * 2] * 2 + [[10. * i + 1] * 2] * 2))
        ok = ok and torch.equal(hs[3].wait(), expect)
        # split amax all-reduce (bench.py pipelines the next batch's front-end between the two halves): two in flight
        sl = [torch.full((64,), float("-inf")) for _ in range(2)]
        sl[0][5 + rank] = 1.0 + rank
        sl[1][rank] = -3.0 - rank
        pend = [amax_allreduce_async(t) for t in sl]
        for t, pnd in zip(sl, pend):
            amax_allreduce_finish(pnd)
        ok = ok and float(sl[0].max()) == 2.0 and float(sl[1].max()) == -3.0
        ok = ok and float(sl[0][5]) == 1.0 and float(sl[0][6]) == 2.0 and float(sl[0][7]) == float("-inf")   # slot-wise MAX
        # reduced payloads (round 3: what crosses xGMI per clip can be cut before the collective)
        h16 = ShardedUiT(fake_model, gather_dtype=torch.float16)(full[lo:hi])
        ok = ok and h16.dtype == torch.float16 and torch.equal(h16, expect.to(torch.float16))
        sl3 = ShardedUiT(fake_model, slice_classes=3)(full[lo:hi])
        ok = ok and sl3.shape == (B, 3) and torch.equal(sl3, expect[:, :3])
        host_topk = lambda p, k: tuple(t.contiguous() for t in p.topk(k))            # stands in for uitk_topk (device only)
        tv, ti = ShardedUiT(fake_model, topk=2, topk_fn=host_topk)(full[lo:hi])
        ev, ei = expect.topk(2)
        ok = ok and torch.equal(tv, ev) and torch.equal(ti, ei) and ti.dtype == torch.int64
        hh = ShardedUiT(fake_model, topk=2, topk_fn=host_topk)(full[lo:hi], async_gather=True)
        tv, ti = hh.wait()
        ok = ok and torch.equal(tv, ev) and torch.equal(ti, ei)
        ok = ok and ShardedUiT(fake_model, topk=5).payload_bytes_per_clip(537) == 40
        ok = ok and ShardedUiT(fake_model, gather_dtype=torch.float16, slice_classes=527).payload_bytes_per_clip(537) == 1054
        # gather='peer' whose IPC precondition fails: EVERY rank falls back to the RCCL / gloo gather, loudly
        import warnings
        import uit_mobile_amd.sharded as S

        class Refuses:
            def __init__(self, *a, **k):
                raise S.PeerGatherUnavailable("HSA_ENABLE_IPC_MODE_LEGACY=0 is not set (test)")
        keep, S.PeerGather = S.PeerGather, Refuses
        try:
            sp = ShardedUiT(fake_model, gather="peer")
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                outp = sp(full[lo:hi])
            ok = ok and torch.equal(outp, expect) and sp.gather == "rccl" and any("falling back" in str(x.message) for x in w)
        finally:
            S.PeerGather = keep
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
