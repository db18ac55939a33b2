"""GPU: encoder time of the full-width-attention factories (mean pooling, full clips), for same-box A/Bs of library builds
(UITK_LIB_PATH selects the library)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bench import synth_batch
from uit_mobile_amd.synth import make_trained_like_state_dict
from uit_mobile_amd import models
x = synth_batch(4096, 16000, 0, torch.device("cuda", 0), False)
for name in ("audio_transformer_h128_d4_m3", "audio_transformer_h128_d6_m3_relu"):
    sd = make_trained_like_state_dict(name, 200, attention="full")
    for prec in ("f16c", "f16x2"):
        m = getattr(models, name)(outputdim=537, target_length=102, precision=prec)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.cuda().eval()
        with torch.no_grad():
            st = m.frontend(x)
            for _ in range(10): m.encode(st)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): m.encode(st)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print(f"{name:36s} {prec:6s}: encoder {dt * 1e3:.4f} ms for 4096 clips", flush=True)
