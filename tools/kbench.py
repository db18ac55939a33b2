"""GPU micro-bench: per-kernel time of the front-end and encoder for a few variants, interleaved in ONE process.
    python tools/kbench.py [model] [B] [rounds]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from synth_weights import make_state_dict  # noqa: E402
from uit_mobile_amd import models  # noqa: E402
from bench import synth_batch, GEMM_FLOP  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "uit_xs"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict(name, 13).items()}
x = synth_batch(B, 16000, 0, dev, False)
xi = synth_batch(B, 16000, 0, dev, True)
variants = {}
VARS = [("w8/f16", "8", "f16", "0", "0"), ("w4/f16", "4", "f16", "0", "0")]
for d in (4000, 8000, 12000, 16000, 24000):
    VARS.append((f"w4/slot{d//1000}k", "4", "f16", "0", str(d)))
os.environ["UITK_ENC_STAGGER_MODE"] = "2"
for tag, waves, prec, sched, stag in VARS:
    os.environ["UITK_ENC_WAVES"] = waves
    os.environ["UITK_ENC_SCHED"] = sched
    os.environ["UITK_ENC_STAGGER"] = stag
    m = getattr(models, name)(outputdim=537, target_length=102, precision=prec)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    m(x[:64]); torch.cuda.synchronize()
    variants[tag] = m
res = {k: ([], []) for k in variants}
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for r in range(rounds):
    for tag, m in variants.items():
        for _ in range(3):
            m(x)
        fe, en = [], []
        for _ in range(10):
            ev[0].record()
            m(x, amax_sync=lambda s: ev[1].record())
            ev[2].record()
            torch.cuda.synchronize()
            fe.append(ev[0].elapsed_time(ev[1])); en.append(ev[1].elapsed_time(ev[2]))
        res[tag][0].append(np.median(fe)); res[tag][1].append(np.median(en))
for tag, (fe, en) in res.items():
    e = float(np.median(en))
    print(f"{tag:8s} frontend {np.median(fe):.4f} ms (min {min(fe):.4f})  encoder {e:.4f} ms (min {min(en):.4f})  "
          f"-> {B * GEMM_FLOP[name] / e / 1e9:.0f} TFLOP/s, total {B / (np.median(fe) + e) * 1e3 / 1e6:.2f} M clips/s")
m = variants["w8/f16"]
fe = []
for _ in range(10):
    ev[0].record(); m(xi, amax_sync=lambda s: ev[1].record()); ev[2].record(); torch.cuda.synchronize()
    fe.append(ev[0].elapsed_time(ev[1]))
print(f"int16 input: frontend {np.median(fe):.4f} ms")
