"""GPU diagnostic: HIP path vs oracle, stage by stage.  Run on the GPU box: python tools/diag.py [model] [prec]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from oracle.oracle import Oracle  # noqa: E402
from synth_weights import make_state_dict, synth_pcm_i16  # noqa: E402
from uit_mobile_amd import models  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "uit_xxxs"
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 6
T = int(sys.argv[4]) if len(sys.argv) > 4 else 16000
sd = make_state_dict(name, 11)
pcm = synth_pcm_i16(B, T, seed=0)
o = Oracle(sd)
ref = o.forward(pcm, want_stages=True)
m = getattr(models, name)(outputdim=537, target_length=102, precision=prec)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
m = m.eval()
x = torch.from_numpy(pcm).cuda()
mel = m.melspec_db(x).cpu().numpy()
d = np.abs(mel - ref["mel_db"])
print("mel_db max diff", d.max(), "per clip", d.reshape(B, -1).max(1), "nan", np.isnan(mel).sum())
probs, toks = m.debug_tokens(x)
toks = toks.cpu().numpy()
for s in range(toks.shape[2]):
    dd = np.abs(toks[:, :, s, : ref["stages"].shape[3]] - ref["stages"][:, :, s])
    print(f"stage {s}: max diff {dd.max():.3e} per clip {dd.reshape(B, -1).max(1)} nan {np.isnan(toks[:, :, s]).sum()}")
p2, lg = m(x, return_logits=True)
torch.cuda.synchronize()
lg = lg.cpu().numpy()
print("logits max diff", np.abs(lg - ref["logits"]).max(), "per clip", np.abs(lg - ref["logits"]).reshape(B, -1).max(1))
print("probs  max diff", np.abs(p2.cpu().numpy() - ref["probs"]).max())
xf = (torch.from_numpy(pcm).float() / 32768.0).cuda()
p3 = m(xf)
print("f32-input vs i16-input probs diff", (p3 - p2).abs().max().item())
