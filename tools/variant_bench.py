"""GPU: encoder time of the factory / pooling / precision variants for 4096 clips, for same-box A/Bs of library builds
(UITK_LIB_PATH selects the library).

    python tools/variant_bench.py [pooling]      # uit_xs + the full-width-attention factories x mean / token / dm x f16c + the split mode
    python tools/variant_bench.py attention      # the full-width-attention factories only (mean pooling): d4 GELU, d6 ReLU
    python tools/variant_bench.py token          # pooling 'token' only: f16c / the split-weight mode / f16a / f16ax, every factory family

Logs of the round-4 A/Bs made with it: profiles/round4/{token_pooling_two_tiles,dm_pooling_two_phase,full_attention_registers,gelu_packed}_ab.log"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from bench import synth_batch  # noqa: E402
from uit_mobile_amd.synth import make_trained_like_state_dict  # noqa: E402
from uit_mobile_amd import models  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "pooling"
nsamp = int(sys.argv[2]) if len(sys.argv) > 2 else 16000          # e.g. 8000: short clips (masked time patches)
x = synth_batch(4096, nsamp, 0, torch.device("cuda", 0), False)
if mode == "attention":
    cases = [(n, "full", "mean", p) for n in ("audio_transformer_h128_d4_m3", "audio_transformer_h128_d6_m3_relu") for p in ("f16c", "f16x2")]
elif mode == "token":
    cases = [(n, att, "token", p) for n, att in (("uit_xs", "bneck"), ("uit_xxs", "bneck"), ("audio_transformer_h128_d4_m3", "full"), ("audio_transformer_h128_d6_m3_relu", "full"))
             for p in ("f16c", "f16l6" if att == "bneck" else "f16x2", "f16a", "f16ax")]
else:
    cases = [(n, att, pooling, p) for n, att in (("uit_xs", "bneck"), ("audio_transformer_h128_d4_m3", "full")) for pooling in ("mean", "token", "dm")
             for p in ("f16c", "f16l6" if att == "bneck" else "f16x2") + (("f16a", "f16ax") if pooling == "token" else ())]
for name, att, pooling, prec in cases:
    sd = make_trained_like_state_dict(name, 200, attention=att)
    m = getattr(models, name)(outputdim=537, target_length=102, pooling=pooling, precision=prec)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.cuda().eval()
    with torch.no_grad():
        st = m.frontend(x)
        for _ in range(10):
            m.encode(st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            m.encode(st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
    print(f"{name:36s} pooling {pooling:5s} {prec:6s}: encoder {dt * 1e3:.4f} ms for 4096 clips", flush=True)
