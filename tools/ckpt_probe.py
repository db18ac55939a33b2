"""GPU: one checkpoint under every precision the family admits -- deviation from the C oracle on the wide held-out set, next to the
library's own estimate (uitk_self_check).    python tools/ckpt_probe.py <factory> <full|bneck> <gelu|relu> <pooling> <seed> [structured|gaussian]"""
import sys
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tests" / "golden")]
from oracle.oracle import Oracle  # noqa: E402
from uit_mobile_amd import models  # noqa: E402
from uit_mobile_amd.synth import make_structured_trained_like_state_dict, make_trained_like_state_dict  # noqa: E402
import test_gpu_parity as T  # noqa: E402

name, att, act, pooling, seed = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
style = sys.argv[6] if len(sys.argv) > 6 else "structured"
make_sd = {"gaussian": make_trained_like_state_dict, "structured": make_structured_trained_like_state_dict}[style]
golden = lambda n: dict(np.load(ROOT / "tests" / "golden" / f"{n}.npz"))
x, names = T.held_out_inputs(golden, with_names=True)
xd = torch.from_numpy(x).cuda()
sd = make_sd(name, seed, attention=att)
r = Oracle(sd, act=act, pooling=pooling).forward(x)
print(f"{name} {att} {act} pooling {pooling} seed {seed} ({style}); oracle logits {r['logits'].min():.2f} .. {r['logits'].max():.2f}")
precs = ["auto", "f16", "f16c", "f16x2", "bf16"] + (["f16a", "f16ax"] if pooling == "token" else []) + (["f16l6"] if (att, act) == ("bneck", "relu") else [])
for prec in precs:
    if pooling != "mean" and prec == "bf16":
        continue
    with warnings.catch_warnings(record=True) as ws:
        warnings.simplefilter("always")
        m = getattr(models, name)(outputdim=537, target_length=102, pooling=pooling, precision=prec)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        m = m.cuda().eval().prepare("cuda")
        with torch.no_grad():
            lg = m(xd, return_logits=True)[1].cpu().numpy().reshape(r["logits"].shape)
        d = np.abs(lg - r["logits"]).reshape(len(x), -1).max(1)
        sc = m.self_check()
        rep = m.calibration_report()
    print(f"  {prec:6s} -> runs {m.effective_precision():28s} worst max|dlogit| {d.max():.2e} (input {names[int(d.argmax())]}), median over clips {np.median(d):.2e}; "
          f"self_check {sc and '%.2e' % sc[0]}; report {None if rep is None else {k: (round(v, 6) if isinstance(v, float) else v) for k, v in rep.items() if k in ('max_nearest', 'max_calibrated', 'accepted', 'escalated', 'within_budget')}}")
    for w in ws:
        if "Parity" in type(w.message).__name__:
            print("      warning:", str(w.message)[:300])
