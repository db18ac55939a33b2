"""GPU box: launch-to-launch distribution of the front-end time at UiT-XS x 8192 (one input buffer, back-to-back forward calls)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from uit_mobile_amd.synth import make_state_dict
from uit_mobile_amd import models
from bench import synth_batch
dev = torch.device("cuda", 0)
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xs", 13).items()}
x = synth_batch(8192, 16000, 0, dev, False)
m = models.uit_xs(outputdim=537, target_length=102).to(dev); m.load_state_dict(sd); m.eval()
for _ in range(150): m(x)
n = 200
evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
for e in evs:
    e[0].record(); m(x, amax_sync=lambda s, e=e: e[1].record()); e[2].record()
torch.cuda.synchronize()
fe = np.array([e[0].elapsed_time(e[1]) for e in evs]); en = np.array([e[1].elapsed_time(e[2]) for e in evs])
print("split", m.get_option("fe_asym_pm"))
print("front-end  percentiles 0/10/25/50/75/90/100:", np.round(np.percentile(fe, [0, 10, 25, 50, 75, 90, 100]), 4))
print("encoder    percentiles 0/10/25/50/75/90/100:", np.round(np.percentile(en, [0, 10, 25, 50, 75, 90, 100]), 4))
print("front-end sequence (first 60):", " ".join(f"{v:.3f}" for v in fe[:60]))
h, edges = np.histogram(fe, bins=12)
print("histogram:", list(zip(np.round(edges[:-1], 3), h)))
