"""Front-end occupancy probe (round 4): time mel_frontend_kernel for a library build (UITK_LIB_PATH) at
FE_WG workgroups per CU.  With the lab builds of tools/mklib.sh (-DUITK_LAB -DFE_NOCONST -DFE_OCC_LAB=2|3: the product's instruction
stream with 60 VGPRs of per-lane constants removed -- timing only, results wrong) this measures what a third wave per SIMD buys when
it costs neither LDS reads nor arithmetic.    FE_WG=3 UITK_LIB_PATH=tools/ab/libuitk_feC.so python tools/fe_occ_probe.py"""
import os, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT)]
from uit_mobile_amd.synth import make_state_dict
from uit_mobile_amd import models
from bench import synth_batch
dev = torch.device("cuda", 0)
wg = int(os.environ.get("FE_WG", "2"))
B = int(os.environ.get("FE_B", "8192"))
name = os.environ.get("FE_MODEL", "uit_xs")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict(name, 13).items()}
xs = [synth_batch(B, 16000, i * B, dev, False) for i in range(3)]
m = getattr(models, name)(outputdim=537, target_length=102, precision="f16", fe_autotune=False).to(dev); m.load_state_dict(sd); m.eval()
m.set_option("fe_wg_per_cu", wg)
res = {}
for share in ([500, 550, 590, 630] if wg == 2 else [500]):
    m.set_option("fe_asym_pm", share)
    N = 80
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(N)]
    for i in range(40): m.frontend(xs[i % 3])
    for i in range(N):
        evs[i][0].record(); m.frontend(xs[i % 3]); evs[i][1].record()
    torch.cuda.synchronize()
    t = [e[0].elapsed_time(e[1]) for e in evs]
    res[share] = float(np.median(t))
print(f"{os.path.basename(os.environ.get('UITK_LIB_PATH', 'libuitk.so'))} {name} B={B} wg_per_cu={wg}: " + "  ".join(f"share {k}: {v:.4f} ms" for k, v in res.items()))
