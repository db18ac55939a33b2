"""GPU: max |dlogit| / |dprob| of the HIP path vs the golden fixtures (reference outputs)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from synth_weights import make_state_dict
from uit_mobile_amd import models
for fix, name, prec in (("synth1s_uit_xxxs", "uit_xxxs", "f16"), ("synth1s_uit_xs", "uit_xs", "f16"),
                        ("synth1s_uit_xs_scale3", "uit_xs", "f16"), ("synth1s_uit_xs", "uit_xs", "bf16"),
                        ("synth1s_uit_xs_scale3", "uit_xs", "bf16")):
    g = dict(np.load(ROOT / "tests" / "golden" / f"{fix}.npz"))
    sd = make_state_dict(name, int(g["seed"]), float(g["scale"]))
    m = getattr(models, name)(outputdim=537, target_length=102, precision=prec)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.cuda().eval()
    p, l = m(torch.from_numpy(g["pcm"]).cuda(), return_logits=True)
    print(f"{fix:26s} {prec:5s} max|dlogit| {np.abs(l.cpu().numpy() - g['logits']).max():.2e}  max|dprob| {np.abs(p.cpu().numpy() - g['probs']).max():.2e}")
