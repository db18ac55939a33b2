"""GPU box: logit/prob error of the HIP path against the C oracle over many weight seeds and scales (confidence in the
1e-3 parity margin beyond the committed fixtures).   python tools/seed_sweep.py [nseeds]"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from uit_mobile_amd.synth import make_state_dict, make_trained_like_state_dict, synth_pcm_i16
from uit_mobile_amd import models
from oracle.oracle import Oracle
from bench import synth_batch

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
x = synth_batch(24, 16000, 0, "cpu", False).numpy()
worst = {}
for name in ("uit_xs", "uit_xxs", "uit_xxxs"):
    for scale in (1.0, 2.0, 3.0):
        for prec in ("f16", "f16l6"):
            errs = []
            for seed in range(100, 100 + nseeds):
                sd = make_state_dict(name, seed, scale)
                m = getattr(models, name)(outputdim=537, target_length=102, precision=prec)
                m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.cuda().eval()
                p, l = m(torch.from_numpy(x).cuda(), return_logits=True)
                o = Oracle(sd, 102, "mean", 160, 2)
                r = o.forward(x); wp, wl = r["probs"], r["logits"]
                dl = float(np.abs(l.cpu().numpy().reshape(wl.shape) - wl).max()); dp = float(np.abs(p.cpu().numpy() - wp).max())
                errs.append((dl, dp, float(np.abs(wl).max())))
            e = np.array(errs)
            print(f"{name:9s} scale {scale:.0f} {prec}: max|dlogit| {e[:,0].max():.2e} (median {np.median(e[:,0]):.2e})  "
                  f"max|dprob| {e[:,1].max():.2e}  |logit|max {e[:,2].max():.1f}", flush=True)
# trained-checkpoint magnitudes (synth_weights.make_trained_like_state_dict): logits beyond -10.  Per family every fp16 precision
# it has: the uit_* family f16 / f16l6 (parity mode) / f16x2, the full-width-attention factories f16 / f16x2 (their parity mode).
families = [(n, "bneck", "relu", ("f16", "f16l6", "f16x2")) for n in ("uit_xs", "uit_xxs", "uit_xxxs")] + \
           [("audio_transformer_h128_d6_m3_relu", "full", "relu", ("f16", "f16x2")), ("audio_transformer_h128_d6_m3", "full", "gelu", ("f16", "f16x2")),
            ("audio_transformer_h128_d4_m3", "full", "gelu", ("f16", "f16x2"))]
for name, att, act, precs in families:
    errs = {p: [] for p in precs}
    for seed in range(200, 200 + nseeds):
        sd = make_trained_like_state_dict(name, seed, attention=att)
        r = Oracle(sd, act=act).forward(x); wp, wl = r["probs"], r["logits"]
        for tprec in precs:
            m = getattr(models, name)(outputdim=537, target_length=102, precision=tprec)
            m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.cuda().eval()
            p, l = m(torch.from_numpy(x).cuda(), return_logits=True)
            errs[tprec].append((float(np.abs(l.cpu().numpy().reshape(wl.shape) - wl).max()), float(np.abs(p.cpu().numpy() - wp).max()),
                                float(np.abs(wl).max())))
    for tprec in precs:
        e = np.array(errs[tprec])
        print(f"{name:34s} trained-like {tprec:5s}: max|dlogit| {e[:,0].max():.2e} (median {np.median(e[:,0]):.2e}, per seed "
              f"{' '.join('%.1e' % v for v in e[:,0])})  max|dprob| {e[:,1].max():.2e}  |logit|max {e[:,2].max():.1f}", flush=True)
