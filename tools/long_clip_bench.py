"""GPU box: throughput on long (multi-crop) clips, e.g. AudioSet-style 10 s clips = 10 crops each."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from uit_mobile_amd.synth import make_state_dict
from uit_mobile_amd import models
from bench import synth_batch
sd = make_state_dict("uit_xs", 13)
m = models.uit_xs(outputdim=537, target_length=102); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.cuda().eval()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for B, T in ((8192, 16000), (819, 160000), (820, 160000), (2048, 64000)):      # 819 x 10 crops = 8190 rows: two full encoder rounds; 820: + an 8-row tail launch
    x = synth_batch(B, T, 0, "cuda", False)
    for _ in range(5): m(x)
    fe, en = [], []
    for _ in range(20):
        ev[0].record(); st = m.frontend(x); ev[1].record(); p = m.encode(st); ev[2].record(); torch.cuda.synchronize()
        fe.append(ev[0].elapsed_time(ev[1])); en.append(ev[1].elapsed_time(ev[2]))
    sec = B * T / 16000
    print(f"B {B:5d} x {T/16000:4.1f} s ({m.num_crops(T)} crops): front-end {np.median(fe):.3f} ms  encoder(+crop reduce) {np.median(en):.3f} ms  "
          f"-> {sec / (np.median(fe) + np.median(en)) * 1e3 / 1e6:.2f} M audio-seconds/s")
