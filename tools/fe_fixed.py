"""Front-end / encoder launch time against the batch size (the intercept is the fixed cost of a launch): prints per-B medians.
usage: python tools/fe_fixed.py   (GPU box; UITK_LIB_PATH selects a build for same-box A/B)"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from synth_weights import make_state_dict
from uit_mobile_amd import models
from bench import synth_batch
dev = torch.device("cuda", 0)
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xs", 13).items()}
m = models.uit_xs(outputdim=537, target_length=102).to(dev); m.load_state_dict(sd); m.eval()
xs = synth_batch(8192, 16000, 0, dev, False)
out = []
for B in (64, 1024, 4096, 8192):
    x = xs[:B].contiguous()
    N = 40
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(N)]
    for _ in range(10): m(x)
    for i in range(N):
        e = evs[i]
        e[0].record(); m(x, amax_sync=lambda s, e=e: e[1].record()); e[2].record()
    torch.cuda.synchronize()
    fe = np.median([e[0].elapsed_time(e[1]) for e in evs]); en = np.median([e[1].elapsed_time(e[2]) for e in evs])
    out.append((B, fe, en))
print("  ".join(f"B={B}: fe {fe*1e3:.1f} us enc {en*1e3:.1f} us" for B, fe, en in out))
(b1, f1, _), (b2, f2, _) = out[2], out[3]
print(f"front-end fixed cost (extrapolated from B=4096, 8192): {(2 * f1 - f2) * 1e3:.1f} us")
