"""GPU box: precision f16l6 against the C oracle next to f16 / f16x2 on trained-like checkpoints (quick check; the committed test is
tests/test_gpu_parity.py::test_parity_mode_all_sweep_seeds).   python tools/l6_check.py [model] [seed ...]"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from uit_mobile_amd.synth import make_trained_like_state_dict
from uit_mobile_amd import models
from oracle.oracle import Oracle
from bench import synth_batch

name = sys.argv[1] if len(sys.argv) > 1 else "uit_xs"
seeds = [int(s) for s in sys.argv[2:]] or [205, 207]
x = synth_batch(24, 16000, 0, "cpu", False).numpy()
for seed in seeds:
    sd = make_trained_like_state_dict(name, seed)
    r = Oracle(sd).forward(x); wp, wl = r["probs"], r["logits"]
    for prec in ("f16", "f16x2", "f16l6"):
        m = getattr(models, name)(outputdim=537, target_length=102, precision=prec)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.cuda().eval()
        p, l = m(torch.from_numpy(x).cuda(), return_logits=True)
        dl = float(np.abs(l.cpu().numpy().reshape(wl.shape) - wl).max()); dp = float(np.abs(p.cpu().numpy() - wp).max())
        print(f"{name} seed {seed} {prec:6s}: max|dlogit| {dl:.2e}  max|dprob| {dp:.2e}  |logit|max {np.abs(wl).max():.1f}", flush=True)
