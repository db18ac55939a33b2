"""GPU: a few forwards of the `f32` rung (UiT-XS x 8192 clips) and of one general-engine geometry, for a rocprofv3 --kernel-trace --stats
breakdown of the plain fp32 path (which kernel the time goes to):

    rocprofv3 --kernel-trace --stats -d gpurun_out/f32prof -o f32 -- python3 tools/f32_breakdown.py"""
import sys
import warnings
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from uit_mobile_amd import models  # noqa: E402
from uit_mobile_amd.synth import make_state_dict  # noqa: E402

warnings.simplefilter("ignore")
which = sys.argv[1] if len(sys.argv) > 1 else "f32"
if which == "f32":
    m = models.uit_xs(outputdim=537, target_length=102, precision="f32")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xs", 3).items()})
    B, T = 8192, 16000
elif which == "default":
    m = models.UITBase(outputdim=527)                  # the constructor's own defaults: 768 wide, 12 blocks, 12 heads, 85.7 M parameters
    B, T = 64, 160000
else:
    m = models.UITBase(embed_dim=128, depth=12, num_heads=2, mlp_ratio=3.0, pooling="mean", attention_type="BNeckAttention", act_layer=nn.ReLU,
                       target_length=102, n_mels=128, n_fft=1024, win_size=1024, outputdim=537)
    B, T = 4096, 16000
m = m.cuda().eval()
x = torch.rand(B, T, device="cuda") - 0.5
m.prepare("cuda", (B, T))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.no_grad():
    e0.record()
    for _ in range(5):
        m(x)
    e1.record()
torch.cuda.synchronize()
print(f"{which}: {e0.elapsed_time(e1) / 5:.2f} ms per {B} clips = {B / (e0.elapsed_time(e1) / 5) :.0f} k clips/s")
