"""GPU: same-box A/B of front-end builds (UITK_LIB_PATH selects the library; one process per library so that each loads its own):
front-end launch time at the two bench shapes (UiT-XS x 8192 = config #3, UiT-3XS x 4096 = config #2) + a checksum of the operands the
front-end wrote and of the probabilities, so that a variant that changes a bit is seen.

    for lib in uit_mobile_amd/libuitk.so tools/ab/libuitk_fe_dpp.so; do UITK_LIB_PATH=$PWD/$lib python tools/fe_variant_ab.py; done
Round 5: the in-register (DPP) transpose between the two radix-16 FFT stages, built with tools/mklib.sh fe_dpp -DUITK_LAB -DFE_XPOSE_DPP."""
import hashlib
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT)]
from bench import synth_batch  # noqa: E402
from uit_mobile_amd import models  # noqa: E402
from uit_mobile_amd.synth import make_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
rounds = int(os.environ.get("AB_ROUNDS", "3"))
for name, B in (("uit_xs", 8192), ("uit_xxxs", 4096)):
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict(name, 13).items()}
    x = synth_batch(B, 16000, 0, dev, False)
    m = getattr(models, name)(outputdim=537, target_length=102, precision="f16").to(dev)
    m.load_state_dict(sd)
    m = m.eval().prepare(dev, (B, 16000))
    with torch.no_grad():
        mel = m.melspec_db(x[:64])
        p = m(x)
    torch.cuda.synchronize()
    h = hashlib.sha256(mel.cpu().numpy().tobytes() + p.cpu().numpy().tobytes()).hexdigest()[:16]
    meds = []
    for _ in range(rounds):
        N = 60
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(N)]
        with torch.no_grad():
            for _ in range(20):
                m.encode(m.frontend(x))
            for e in evs:
                e[0].record()
                st = m.frontend(x)
                e[1].record()
                m.encode(st)
        torch.cuda.synchronize()
        meds.append(float(np.median([e[0].elapsed_time(e[1]) for e in evs])))
    print(f"{Path(os.environ.get('UITK_LIB_PATH', 'libuitk.so')).name:28s} {name} x {B}: front-end {min(meds):.4f} ms (medians of {rounds} rounds: "
          + " ".join(f"{v:.4f}" for v in meds) + f"), share {m.get_option('fe_asym_pm')[0]}, outputs sha {h}", flush=True)
