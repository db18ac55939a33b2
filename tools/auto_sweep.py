"""GPU box: the default mode ('auto') over trained-like checkpoints OUTSIDE the test-suite's sweep (seeds 212 ...), every family, on the
wide held-out set of tests/test_gpu_parity.py: worst logit deviation from the C oracle per family, the worst input, what the budget
check decided per checkpoint.     python tools/auto_sweep.py [nseeds=24] [all|token|mean|dm|more] [gaussian|structured|sink|sink4]
-> profiles/round5/auto_sweep_*.log      ("structured": synth.make_structured_trained_like_state_dict -- heavy-tailed, low-rank, uneven rows; "sink" / "sink4": an attention sink in
every block and head, synth.make_sink_trained_like_state_dict at gain 2 / 4)"""
import sys
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tests" / "golden")]
from oracle.oracle import Oracle  # noqa: E402
from uit_mobile_amd import models  # noqa: E402
from uit_mobile_amd.synth import make_sink_trained_like_state_dict, make_structured_trained_like_state_dict, make_trained_like_state_dict  # noqa: E402
import test_gpu_parity as T  # noqa: E402

warnings.simplefilter("ignore")
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
golden = lambda name: dict(np.load(ROOT / "tests" / "golden" / f"{name}.npz"))
x, names = T.held_out_inputs(golden, with_names=True)
xd = torch.from_numpy(x).cuda()
families = [(n, "bneck", "relu", "mean") for n in ("uit_xs", "uit_xxs", "uit_xxxs")] + \
           [("audio_transformer_h128_d4_m3", "full", "gelu", "mean"), ("audio_transformer_h128_d6_m3_relu", "full", "relu", "mean"),
            ("uit_xxs", "bneck", "relu", "token"), ("uit_xs", "bneck", "relu", "token"), ("audio_transformer_h128_d4_m3", "full", "gelu", "token"),
            ("uit_xxs", "bneck", "relu", "dm")]
only = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "all" else None           # e.g. "token": that pooling's rows only; "more": the second list
style = sys.argv[3] if len(sys.argv) > 3 else "gaussian"
make_sd = {"gaussian": make_trained_like_state_dict, "structured": make_structured_trained_like_state_dict, "sink": make_sink_trained_like_state_dict,
           "sink4": lambda *a, **k: make_sink_trained_like_state_dict(*a, gain=4.0, **k)}[style]
print(f"checkpoint family: {style}", flush=True)
if only == "more":                                          # the remaining factories / poolings, and the other clip shapes (0.5 s, 4 s)
    families = [("audio_transformer_h128_d6_m3", "full", "gelu", "mean"), ("audio_transformer_h128_d4_m3_relu", "full", "relu", "mean"),
                ("uit_xs", "bneck", "relu", "dm"), ("audio_transformer_h128_d4_m3", "full", "gelu", "dm"),
                ("uit_xs", "bneck", "relu", "mean:short"), ("uit_xs", "bneck", "relu", "mean:long"),
                ("audio_transformer_h128_d4_m3", "full", "gelu", "mean:short"), ("uit_xxs", "bneck", "relu", "token:short")]
    only = None
x_full = x
for name, att, act, pooling in families:
    pooling, _, shape = pooling.partition(":")
    x = x_full[:, 4000:12000] if shape == "short" else np.tile(x_full[:40], (1, 4)) if shape == "long" else x_full
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    names_ = names[:len(x)]
    if only and pooling != only:
        continue
    rows, esc = [], []
    for seed in range(212, 212 + nseeds):
        sd = make_sd(name, seed, attention=att)
        m = getattr(models, name)(outputdim=537, target_length=102, pooling=pooling)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        m = m.cuda().eval().prepare("cuda")
        r = Oracle(sd, act=act, pooling=pooling).forward(x)
        with torch.no_grad():
            if pooling == "dm":
                d = np.abs(m(xd).cpu().numpy() - r["probs"]).reshape(len(x), -1).max(1)
            else:
                d = np.abs(m(xd, return_logits=True)[1].cpu().numpy().reshape(r["logits"].shape) - r["logits"]).reshape(len(x), -1).max(1)
        rep = m.calibration_report()
        rows.append((float(d.max()), seed, names_[int(d.argmax())], rep["max_calibrated"]))
        if m.effective_precision() != ("f16a" if pooling == "token" else "f16c"):
            esc.append((seed, m.effective_precision() + ("" if rep.get("within_budget", True) else " (over budget)")))
    w = max(rows)
    e = np.array([r[0] for r in rows])
    what = "max|dprob|" if pooling == "dm" else "max|dlogit|"
    print(f"{name:34s} pooling {pooling:5s} {shape or '1 s':5s}: {nseeds} checkpoints x {len(x)} clips: worst {what} {w[0]:.2e} (seed {w[1]}, input {w[2]}), median {np.median(e):.2e}; "
          f"library's own held-out estimate: worst {max(r[3] for r in rows):.2e}; not in the starting precision: {esc}", flush=True)
    print("      per checkpoint (seed: worst, library's estimate): " + ", ".join(f"{r[1]}: {r[0]:.2e} / {r[3]:.2e}" for r in sorted(rows, key=lambda r: r[1])), flush=True)
