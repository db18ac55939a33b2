"""GPU: bit-stability soak of the encoder instantiations under load (round 6, after the hazard clean-up): for each of a dozen
(factory, pooling, precision, clip length, batch) cases, `repeats` forwards of the same batch while a second model keeps another stream
busy -- every output must equal the first run's bit for bit (NaN == NaN), be finite; a 48-clip batch of the same model sits at the oracle.

    python tools/soak.py [repeats=500]"""
import sys
import time
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tests" / "golden")]
from oracle.oracle import Oracle  # noqa: E402
from uit_mobile_amd import models  # noqa: E402
from uit_mobile_amd.synth import make_state_dict, make_trained_like_state_dict, synth_pcm_i16  # noqa: E402

warnings.simplefilter("ignore")
repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 500
CASES = [("uit_xs", "bneck", "relu", "mean", "auto", 16000, 8192), ("uit_xs", "bneck", "relu", "mean", "auto", 8000, 4096),
         ("uit_xs", "bneck", "relu", "mean", "auto", 16000, 512), ("uit_xs", "bneck", "relu", "mean", "f16l6", 16000, 4096),
         ("uit_xs", "bneck", "relu", "mean", "f16x2", 16000, 4096), ("uit_xs", "bneck", "relu", "mean", "fp8c", 16000, 4096),
         ("uit_xs", "bneck", "relu", "mean", "bf16", 8000, 4096), ("uit_xxs", "bneck", "relu", "dm", "auto", 16000, 2048),
         ("uit_xxs", "bneck", "relu", "token", "auto", 16000, 2048), ("uit_xxs", "bneck", "relu", "token", "f16ax", 8000, 2048),
         ("audio_transformer_h128_d4_m3", "full", "gelu", "mean", "auto", 16000, 4096), ("audio_transformer_h128_d4_m3", "full", "gelu", "mean", "auto", 8000, 4096),
         ("audio_transformer_h128_d4_m3", "full", "gelu", "dm", "auto", 8000, 2048), ("audio_transformer_h128_d6_m3_relu", "full", "relu", "mean", "f16x2", 16000, 4096),
         ("audio_transformer_h128_d4_m3", "full", "gelu", "token", "auto", 16000, 2048)]
noise = models.uit_xs(outputdim=537, target_length=102, precision="f16")
noise.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xs", 3).items()})
noise = noise.cuda().eval()
xn = torch.from_numpy(synth_pcm_i16(4096, 16000, seed=5)).cuda()
side = torch.cuda.Stream()
bad = 0
for name, att, act, pooling, prec, T, B in CASES:
    sd = make_trained_like_state_dict(name, 205, attention=att)
    m = getattr(models, name)(outputdim=537, target_length=102, pooling=pooling, precision=prec)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.cuda().eval()
    pcm = synth_pcm_i16(B, T, seed=11)
    x = torch.from_numpy(pcm).cuda()
    t0 = time.time()
    with torch.no_grad():
        noise(xn)
        first = m(x).clone()
        torch.cuda.synchronize()
        differ = 0
        for rep in range(repeats):
            if rep % 4 == 0:
                with torch.cuda.stream(side):
                    noise(xn)
            differ += int(not torch.equal(first.view(torch.int32), m(x).view(torch.int32)))
        torch.cuda.synchronize()
    # against the oracle: a batch of its own, small enough to hand over whole (the top_db floor is the amax of the WHOLE batch)
    with torch.no_grad():
        small = m(x[:48]).cpu().numpy()
    ref = Oracle(sd, act=act, pooling=pooling).forward(pcm[:48])["probs"]
    d = float(np.abs(small - ref).max())
    gate = {"bf16": 2e-2, "fp8c": 5e-2}.get(prec, 3e-4)     # (probabilities)
    ok = differ == 0 and bool(torch.isfinite(first).all()) and d < gate
    bad += int(not ok)
    print(f"{name:34s} {pooling:5s} {prec:6s}->{m.effective_precision():6s} T={T:5d} B={B:4d}: {repeats} repeats under load: {differ} differ; "
          f"max|dprob| of 48 clips vs oracle {d:.1e}  [{time.time() - t0:.0f} s] {'ok' if ok else 'FAIL'}", flush=True)
print(f"{len(CASES)} cases, {bad} failures")
sys.exit(1 if bad else 0)
