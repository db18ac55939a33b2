"""GPU box: is precision f16l6 deterministic run to run, and bitwise equivariant under permutations that keep clip pairs together?"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from uit_mobile_amd.synth import make_state_dict, synth_pcm_i16
from uit_mobile_amd import models
pcm = synth_pcm_i16(4096, 16000, seed=41)
perm = np.random.default_rng(3).permutation(2048) * 2
idx = np.stack([perm, perm + 1], 1).reshape(-1)
for prec in ("f16", "f16l6", "f16x2"):
    sd = make_state_dict("uit_xxxs", 5)
    m = models.uit_xxxs(outputdim=537, target_length=102, precision=prec)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.cuda().eval()
    x = torch.from_numpy(pcm).cuda(); xp = torch.from_numpy(pcm[idx]).cuda()
    with torch.no_grad():
        a = m(x).cpu().numpy(); b = m(x).cpu().numpy(); c = m(xp).cpu().numpy()
        mel_a = m.melspec_db(x[:256]).cpu().numpy(); 
        reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
        n_bad = sum(int(not np.array_equal(a, m(x).cpu().numpy())) for _ in range(reps))
    print(prec, f"repeats differing from the first run: {n_bad}/{reps}")
    d_rep = np.abs(a - b).max(); d_perm = np.abs(c - a[idx]).max()
    bad = np.argwhere(np.abs(c - a[idx]).max(1) > 0)[:, 0]
    print(prec, "run-to-run max diff", d_rep, "| pair-permutation max diff", d_perm, "| rows differing", len(bad), bad[:10], "orig rows", idx[bad[:10]])
