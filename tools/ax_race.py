"""GPU box, lab: the f16ax BNeckAttention race of round 5 (profiles/round5/f16ax_bisect.log), one library per process (UITK_LIB_PATH).

    UITK_LIB_PATH=$PWD/tools/ab/libuitk_ax_chained.so python tools/ax_race.py [repeats=20]

Checkpoint synth.make_structured_trained_like_state_dict("uit_xxs", 234), pooling 'token', precision f16ax on uitk_create's image
(no calibration).  (a) eight clips against the C oracle, per clip; (b) 1024 clips, `repeats` runs beside a second model on another
stream: how many runs differ from the first bit for bit, how many outputs are not finite."""
import os
import sys
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tests" / "golden")]
from oracle.oracle import Oracle  # noqa: E402
from uit_mobile_amd import models  # noqa: E402
import uit_mobile_amd.models.uit as U  # noqa: E402
from uit_mobile_amd.synth import make_state_dict, make_structured_trained_like_state_dict, synth_pcm_i16  # noqa: E402

warnings.simplefilter("ignore")
repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 20
factory = sys.argv[2] if len(sys.argv) > 2 else "uit_xxs"
lib = os.path.basename(os.environ.get("UITK_LIB_PATH", "libuitk.so"))
sd = make_structured_trained_like_state_dict(factory, 234)
U._CALIBRATING = tuple(p for p in U._CALIBRATING if p != "f16ax")        # uitk_create's image: the case that was wrong
m = getattr(models, factory)(outputdim=537, target_length=102, pooling="token", precision="f16ax")
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
m = m.cuda().eval()
orc = Oracle(sd, act="relu", pooling="token")

pcm8 = synth_pcm_i16(8, 16000, seed=3)
with torch.no_grad():
    lg = m(torch.from_numpy(pcm8).cuda(), return_logits=True)[1].cpu().numpy()
ref = orc.forward(pcm8)["logits"]
d8 = np.abs(lg.reshape(ref.shape) - ref).reshape(8, -1).max(1)
print(f"{lib:34s} 8 clips vs oracle: " + " ".join(f"{v:.1e}" for v in d8), flush=True)

pcm = synth_pcm_i16(1024, 16000, seed=97)
x = torch.from_numpy(pcm).cuda()
noise = models.uit_xs(outputdim=537, target_length=102, precision="f16")
noise.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xs", 3).items()})
noise = noise.cuda().eval()
xn = torch.from_numpy(synth_pcm_i16(4096, 16000, seed=5)).cuda()
side = torch.cuda.Stream()
with torch.no_grad():
    noise(xn)
    first = m(x, return_logits=True)[1].clone()
    torch.cuda.synchronize()
    differing, nonfinite = 0, int((~torch.isfinite(first)).any(-1).sum())
    for rep in range(repeats):
        with torch.cuda.stream(side):
            for _ in range(1 + rep % 3):
                noise(xn)
        out = m(x, return_logits=True)[1]
        differing += int(not torch.equal(first.view(torch.int32), out.view(torch.int32)))
        nonfinite = max(nonfinite, int((~torch.isfinite(out)).reshape(len(pcm), -1).any(-1).sum()))
        torch.cuda.synchronize()
ref = orc.forward(pcm)["logits"]
d = np.abs(first.cpu().numpy().reshape(ref.shape) - ref).reshape(len(pcm), -1).max(1)
bad = int((~(d < 8e-4)).sum())
print(f"{lib:34s} 1024 clips x {repeats} repeats under load: {differing} differ, clips with a non-finite logit (worst run) {nonfinite}, "
      f"first run vs oracle: worst finite {np.nanmax(np.where(np.isfinite(d), d, 0)):.2e}, clips over 8e-4 or NaN {bad}", flush=True)

if os.environ.get("AX_TOKENS"):                       # residual stream after every block of the eight clips -> gpurun_out/ax_tokens_<lib>.npy
    with torch.no_grad():
        _, tok = m.debug_tokens(torch.from_numpy(pcm8).cuda())
    Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
    np.save(ROOT / "gpurun_out" / f"ax_tokens_{lib.replace('.so', '')}.npy", tok.cpu().numpy())
