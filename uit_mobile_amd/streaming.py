"""Input pipeline for host-resident audio (the step BEFORE the hot path).

The reference feeds the model from DataLoader workers: int16 HDF5 clips -> float32 / 32768 on the CPU -> zero-pad
collate -> `x.to(DEVICE, non_blocking=True)` (dataset.py:40-56, 183-204; run.py:30-32).  That moves 4 bytes per
sample over PCIe and spends CPU time on the int16->float conversion.  Here the int16 PCM stays int16 end to end:

    pinned host staging (2 buffers) --async H2D on a copy stream--> device int16 --> uitk (x 1/32768 fused in the load)

`HostClipStreamer.run(batches)` overlaps the copy of batch i+1 with the two kernels of batch i and yields the
outputs in order.  PCIe Gen5 x16 (~63 GB/s) bounds this at ~2 M int16 clips/s per GPU (1 M for float32) -- far below
the device-resident rate, which is why bench.py's `value` is measured with the input already in HBM.
"""
import torch


class HostClipStreamer:
    def __init__(self, model, batch_size, clip_samples, device=None, dtype=torch.int16):
        self.model = model
        self.device = torch.device(device or "cuda")
        self.B, self.T, self.dtype = batch_size, clip_samples, dtype
        self.copy_stream = torch.cuda.Stream(self.device)
        self.host = [torch.empty((batch_size, clip_samples), dtype=dtype).pin_memory() for _ in range(2)]
        self.dev = [torch.empty((batch_size, clip_samples), dtype=dtype, device=self.device) for _ in range(2)]
        self.ready = [torch.cuda.Event() for _ in range(2)]     # H2D of slot i finished
        self.free = [torch.cuda.Event() for _ in range(2)]      # compute on slot i finished (slot reusable)

    def _stage(self, slot, batch):
        """CPU copy into pinned memory, then async H2D on the copy stream."""
        n = batch.shape[0]
        if batch.is_pinned() and batch.dtype == self.dtype:
            src = batch                                         # producer wrote straight into pinned memory: no CPU copy
        else:
            # the H2D copy issued two batches ago may still be reading this pinned buffer: the HOST must wait for it
            # (the `free` event below only orders the copy stream against the kernels)
            self.ready[slot].synchronize()
            self.host[slot][:n].copy_(batch)                    # CPU memcpy (+ dtype conversion) into the staging buffer
            src = self.host[slot][:n]
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.free[slot])        # do not overwrite a buffer the kernels still read
            self.dev[slot][:n].copy_(src, non_blocking=True)
            self.ready[slot].record(self.copy_stream)
        return n

    def run(self, batches):
        """batches: iterable of CPU tensors [b <= batch_size, clip_samples] (int16 PCM or float32).  Yields probs."""
        main = torch.cuda.current_stream(self.device)
        for e in self.free:
            e.record(main)
        it = iter(batches)
        try:
            nxt = next(it)
        except StopIteration:
            return
        slot = 0
        n = self._stage(slot, nxt)
        while True:
            try:
                nxt = next(it)
                n_next = self._stage(slot ^ 1, nxt)             # overlaps with the kernels below
            except StopIteration:
                nxt, n_next = None, 0
            main.wait_event(self.ready[slot])
            out = self.model(self.dev[slot][:n])
            self.free[slot].record(main)
            yield out
            if nxt is None:
                return
            slot, n = slot ^ 1, n_next
