"""GPU box: where do two runs of the f16l6 encoder first differ?  Dumps the residual stream after every block twice (debug tokens)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from uit_mobile_amd.synth import make_state_dict, synth_pcm_i16
from uit_mobile_amd import models
prec = sys.argv[1] if len(sys.argv) > 1 else "f16l6"
B = 1024
sd = make_state_dict("uit_xxxs", 5)
m = models.uit_xxxs(outputdim=537, target_length=102, precision=prec)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.cuda().eval()
x = torch.from_numpy(synth_pcm_i16(B, 16000, seed=41)).cuda()
with torch.no_grad():
    runs = []
    for r in range(4):
        p, t = m.debug_tokens(x)
        runs.append((p.cpu().numpy(), t.cpu().numpy()[:, 0]))          # [B, depth+1, 24, 128]
    pp = [m(x).cpu().numpy() for _ in range(4)]
print("plain forward run-to-run max diff:", max(float(np.abs(pp[0] - q).max()) for q in pp[1:]))
for r in range(1, 4):
    d = np.abs(runs[0][1] - runs[r][1])
    print(f"run 0 vs {r}: probs diff {np.abs(runs[0][0] - runs[r][0]).max():.3g}; token diffs per stage (max):", [float(d[:, s].max()) for s in range(d.shape[1])])
    if d.max() > 0:
        s = int(np.argmax(d.reshape(d.shape[0], d.shape[1], -1).max(2).max(0) > 0))
        rows, toks, chans = np.nonzero(d[:, s])
        print(f"   first differing stage {s}: clips {sorted(set(rows.tolist()))[:12]} (parity counts even/odd: {(rows % 2 == 0).sum()}/{(rows % 2 == 1).sum()}), "
              f"tokens {sorted(set(toks.tolist()))}, channels {sorted(set(chans.tolist()))[:40]} n={len(set(chans.tolist()))}")
