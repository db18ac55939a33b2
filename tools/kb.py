"""One-variant kernel timing (precision by KB_PREC, library by UITK_LIB_PATH): prints front-end / encoder medians for UiT-XS B=8192."""
import os, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from uit_mobile_amd.synth import make_state_dict
from uit_mobile_amd import models
from bench import synth_batch
dev = torch.device("cuda", 0)
prec = os.environ.get("KB_PREC", "f16l6")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xs", 13).items()}
x = synth_batch(8192, 16000, 0, dev, False)
m = models.uit_xs(outputdim=537, target_length=102, precision=prec).to(dev); m.load_state_dict(sd); m.eval()
N = 60
evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(N)]
for _ in range(30): m(x)
for i in range(N):
    e = evs[i]
    e[0].record(); m(x, amax_sync=lambda s, e=e: e[1].record()); e[2].record()
torch.cuda.synchronize()
fe = [e[0].elapsed_time(e[1]) for e in evs]; en = [e[1].elapsed_time(e[2]) for e in evs]
print(f"{prec} frontend {np.median(fe):.4f}  encoder {np.median(en):.4f} (min {min(en):.4f})  total {np.median(fe)+np.median(en):.4f} ms")
