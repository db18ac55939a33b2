"""GPU box experiment: does step n+1's front-end overlap step n's encoder when the two kernels are launched on two HIP
streams?  python tools/overlap_bench.py [steps]   (knobs by environment: UITK_ENC_WAVES=4|8, UITK_FE_WG_PER_CU=1|2)

Modes timed back to back on the same box, same process:
  serial : one stream, FE(n+1) enqueued before enc(n)  (what bench.py did in round 1)
  2stream: FE on stream B, encoder on stream A, events for the per-batch dependency FE(n) -> enc(n) and for the reuse
           of a replica's operand workspace enc(n) -> FE(n+2)
"""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from bench import synth_batch  # noqa: E402
from synth_weights import make_state_dict  # noqa: E402
from uit_mobile_amd import models  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, T = int(os.environ.get("OB_BATCH", 8192)), 16000
dev = torch.device("cuda", 0)
sd = make_state_dict("uit_xs", 13)
NREP = int(os.environ.get("OB_REPLICAS", 2))
reps = []
for _ in range(NREP):
    r = models.uit_xs(outputdim=537, target_length=102)
    r.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    r = r.to(dev).eval()
    r.reserve(B, T, dev)
    reps.append(r)
x = synth_batch(B, T, 0, dev, False)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def serial(n):
    st = reps[0].frontend(x)
    for i in range(n):
        nxt = reps[(i + 1) % NREP].frontend(x) if i + 1 < n else None
        out = reps[i % NREP].encode(st)
        st = nxt
    return out


def two_stream(n):
    fe_done = [torch.cuda.Event() for _ in range(n)]
    enc_done = [torch.cuda.Event() for _ in range(n)]
    states = [None] * n
    out = None
    for i in range(n + 1):
        if i < n:
            with torch.cuda.stream(sb):
                if i >= NREP:
                    sb.wait_event(enc_done[i - NREP])          # this replica's operands were consumed
                states[i] = reps[i % NREP].frontend(x)
                fe_done[i].record(sb)
        j = i - 1
        if j >= 0:
            with torch.cuda.stream(sa):
                sa.wait_event(fe_done[j])
                out = reps[j % NREP].encode(states[j])
                enc_done[j].record(sa)
    return out


with torch.no_grad():
    for mode, fn in (("serial", serial), ("2stream", two_stream), ("serial", serial), ("2stream", two_stream)):
        fn(60)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"{mode:8s} enc_waves={os.environ.get('UITK_ENC_WAVES', '8')} fe_wg_per_cu={os.environ.get('UITK_FE_WG_PER_CU', '2')} "
              f"replicas={NREP}: {el / steps * 1e3:.4f} ms/step  {B * steps / el / 1e6:.2f} M clips/s", flush=True)
