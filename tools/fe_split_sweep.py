"""GPU box: front-end time against the asymmetric pass split (fe_asym_pm) at UiT-XS x 8192, and what uitk_reserve's calibration picks --
several fresh handles, because the pick is a measurement."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from uit_mobile_amd.synth import make_state_dict
from uit_mobile_amd import models
from bench import synth_batch
dev = torch.device("cuda", 0)
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xs", 13).items()}
x = synth_batch(8192, 16000, 0, dev, False)


def fe_ms(m, n=40):
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(n)]
    for _ in range(10): m(x)
    for e in evs:
        e[0].record(); m(x, amax_sync=lambda s, e=e: e[1].record())
    torch.cuda.synchronize()
    t = [e[0].elapsed_time(e[1]) for e in evs]
    return float(np.median(t)), float(min(t))


m = models.uit_xs(outputdim=537, target_length=102).to(dev); m.load_state_dict(sd); m.eval()
for _ in range(100): m(x)                                              # clock ramp
print("calibrated on first full-size call:", m.get_option("fe_asym_pm"), "-> front-end median / min ms", fe_ms(m))
for rnd in range(2):
    for pm in (450, 500, 530, 560, 590, 620, 650, 680):
        m.set_option("fe_asym_pm", pm)
        print(f"round {rnd} fe_asym_pm {pm}: median {fe_ms(m)[0]:.4f} min {fe_ms(m)[1]:.4f}")
for k in range(6):
    m2 = models.uit_xs(outputdim=537, target_length=102).to(dev); m2.load_state_dict(sd); m2.eval()
    m2(x)
    print(f"fresh handle {k}: calibration picked", m2.get_option("fe_asym_pm"), "-> median / min", fe_ms(m2))
