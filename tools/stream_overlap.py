"""GPU: does running consecutive batches on TWO HIP streams (so that one batch's kernels can fill the other's tail) beat the bench's
single-stream order?  Same kernels, same batches; only the streams and their priorities differ.

    python tools/stream_overlap.py [steps]

 serial      one stream: FE(n+1), ENC(n), FE(n+2), ...                     (bench.py's order)
 two         chain A (even batches) on stream A, chain B on stream B, equal priority
 enc-high    per chain: front-ends on a LOW-priority stream, encoders on a HIGH-priority stream (events carry FE(n) -> ENC(n) and
             ENC(n) -> FE(n + R), the workspace reuse of the replica)
 fe-high     the other way round
Each with R = 2 and 3 replicas.  Prints ms per step (median of 5 repeats of `steps` steps, wall clock around a synchronize).

    python tools/stream_overlap.py [steps] coresident
Round 5: the same variants with kernels that CAN share a CU -- encoder on 4-wave workgroups (half a CU's registers and LDS each) and the
front-end on ONE workgroup per CU (the other half): a front-end wave and an encoder wave per SIMD, the matrix pipe and the VALU / LDS
path used by different instruction streams at the same time.  (Round 4 measured the 8-wave encoder, which owns its CU.)
"""
import statistics
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from bench import make_state_dict, synth_batch  # noqa: E402
from uit_mobile_amd import models  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60
    coresident = "coresident" in sys.argv
    dev = torch.device("cuda", 0)
    B, T = 8192, 16000
    sd = make_state_dict("uit_xs", 13)
    reps = []
    for _ in range(4):
        r = models.uit_xs(outputdim=537, target_length=102)
        r.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        reps.append(r.to(dev).eval())
    x = synth_batch(B, T, 0, dev, False)
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    print(f"stream priority range: least {lo}, greatest {hi}", flush=True)
    with torch.no_grad():
        for r in reps:
            r.reserve(B, T, dev)
            r(x[:64])
        reps[0].tune_frontend(x)
        if coresident:
            for r in reps:
                r.set_option("enc_waves", 4)
                r.set_option("fe_wg_per_cu", 1)
            print("co-resident shapes: encoder 4-wave workgroups, front-end one workgroup per CU", flush=True)

        def serial(R, n):
            st = reps[0].frontend(x)
            for i in range(n):
                nxt = reps[(i + 1) % R].frontend(x) if i + 1 < n else None
                reps[i % R].encode(st)
                st = nxt

        def make_streamed(fe_prio, enc_prio, split):
            def run(R, n):
                main_s = torch.cuda.current_stream(dev)
                if split == 2:                             # ONE front-end stream and ONE encoder stream for all replicas
                    fs = [torch.cuda.Stream(dev, priority=fe_prio)] * R
                    es = [torch.cuda.Stream(dev, priority=enc_prio)] * R
                elif split:
                    fs = [torch.cuda.Stream(dev, priority=fe_prio) for _ in range(R)]
                    es = [torch.cuda.Stream(dev, priority=enc_prio) for _ in range(R)]
                else:
                    fs = [torch.cuda.Stream(dev, priority=fe_prio) for _ in range(R)]
                    es = fs
                start = torch.cuda.Event()
                start.record(main_s)
                for s in set(fs + es):
                    s.wait_event(start)
                enc_done = [None] * R
                for i in range(n):
                    k = i % R
                    with torch.cuda.stream(fs[k]):
                        if split and enc_done[k] is not None:
                            fs[k].wait_event(enc_done[k])
                        st = reps[k].frontend(x)
                        fe_done = torch.cuda.Event()
                        fe_done.record(fs[k])
                    with torch.cuda.stream(es[k]):
                        if split:
                            es[k].wait_event(fe_done)
                        reps[k].encode(st)
                        enc_done[k] = torch.cuda.Event()
                        enc_done[k].record(es[k])
                for s in set(fs + es):
                    e = torch.cuda.Event()
                    e.record(s)
                    main_s.wait_event(e)
            return run

        variants = [("serial", serial), ("two (equal priority)", make_streamed(0, 0, False)),
                    ("enc-high / fe-low", make_streamed(lo, hi, True)), ("fe-high / enc-low", make_streamed(hi, lo, True)),
                    ("split streams, equal priority", make_streamed(0, 0, True)),
                    ("FE stream + ENC stream, equal", make_streamed(0, 0, 2)), ("FE stream low + ENC stream high", make_streamed(lo, hi, 2)),
                    ("FE stream high + ENC stream low", make_streamed(hi, lo, 2))]
        for R in ((2, 3) if coresident else (2, 3, 4)):
            for name, fn in variants:
                fn(R, 20)
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    fn(R, steps)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) / steps * 1e3)
                print(f"R={R} {name:32s} {statistics.median(ts):.4f} ms/step  (min {min(ts):.4f} max {max(ts):.4f})  "
                      f"{B / statistics.median(ts) / 1e3:.2f} M clips/s", flush=True)


if __name__ == "__main__":
    main()
