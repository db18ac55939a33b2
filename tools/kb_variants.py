"""GPU box: kernel times of the SURVEY 8f-4 variants (full-width Attention, GELU) next to uit_xxxs / uit_xxs, 8192 clips."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from synth_weights import make_state_dict
from uit_mobile_amd import models
from bench import synth_batch
x = synth_batch(8192, 16000, 0, "cuda", False)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for name, att in (("uit_xxxs", "bneck"), ("audio_transformer_h128_d4_m3_relu", "full"), ("audio_transformer_h128_d4_m3", "full"),
                  ("uit_xxs", "bneck"), ("audio_transformer_h128_d6_m3_relu", "full"), ("audio_transformer_h128_d6_m3", "full")):
    sd = make_state_dict(name, 13, attention=att)
    m = getattr(models, name)(outputdim=537, target_length=102)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.cuda().eval()
    for _ in range(30): m(x)
    en = []
    for _ in range(40):
        st = m.frontend(x); ev[1].record(); p = m.encode(st); ev[2].record()
        torch.cuda.synchronize(); en.append(ev[1].elapsed_time(ev[2]))
    print(f"{name:36s} encoder {np.median(en):.4f} ms", flush=True)
