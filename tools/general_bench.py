"""GPU: throughput of the GENERAL engine (csrc/general.hip) on a few geometries outside the published factories' -- events on the current
stream, median of N forwards, inputs resident.  Not the headline (bench.py measures the MFMA path); recorded in DESIGN.md 4.3.

    python tools/general_bench.py [reps=10]"""
import sys
import warnings
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from uit_mobile_amd.models.uit import UITBase  # noqa: E402

CASES = [  # (name, UITBase kwargs, batch, samples)
    ("uit_xs geometry, 128 mels from n_fft 1024 (48 tokens)", dict(embed_dim=128, depth=12, num_heads=2, mlp_ratio=3.0, pooling="mean", attention_type="BNeckAttention",
                                                                act_layer=nn.ReLU, target_length=102, n_mels=128, n_fft=1024, win_size=1024, outputdim=537), 4096, 16000),
    ("uit_xs with UITBase's default target_length 1012, 10 s clips (248 tokens)", dict(embed_dim=128, depth=12, num_heads=2, mlp_ratio=3.0, pooling="mean", attention_type="BNeckAttention",
                                                                                        act_layer=nn.ReLU, outputdim=537), 512, 160000),
    ("uit_xs with 8 x 8 patches (96 tokens)", dict(embed_dim=128, depth=12, num_heads=2, mlp_ratio=3.0, pooling="mean", attention_type="BNeckAttention", act_layer=nn.ReLU,
                                                   target_length=102, patch_size=8, patch_stride=8, outputdim=537), 2048, 16000),
    ("embed 192, 3 heads, MLP x 4, depth 12, full attention, GELU, cls token (25 tokens)", dict(embed_dim=192, depth=12, num_heads=3, mlp_ratio=4.0, pooling="token", target_length=102,
                                                                                                  outputdim=527), 4096, 16000),
    ("UITBase() with the constructor's own defaults: 768 wide, 12 blocks, 12 heads, MLP x 4, 85.7 M parameters, 10 s clips (249 tokens)", dict(outputdim=527), 64, 160000),
    ("n_fft 400 / hop 160 (DFT path), otherwise uit_xs", dict(embed_dim=128, depth=12, num_heads=2, mlp_ratio=3.0, pooling="mean", attention_type="BNeckAttention", act_layer=nn.ReLU,
                                                              target_length=102, n_fft=400, win_size=400, outputdim=537), 4096, 16000),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    warnings.simplefilter("ignore")
    for name, kw, B, T in CASES:
        torch.manual_seed(0)
        m = UITBase(**kw).cuda().eval()
        assert m.general
        x = (torch.rand(B, T, device="cuda") - 0.5)
        m.prepare("cuda", (B, T))
        fe, tot = [], []
        for _ in range(reps):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            with torch.no_grad():
                e[0].record(); st = m.frontend(x); e[1].record(); m.encode(st); e[2].record()
            torch.cuda.synchronize()
            fe.append(e[0].elapsed_time(e[1])); tot.append(e[0].elapsed_time(e[2]))
        fe_ms, tot_ms = float(np.median(fe)), float(np.median(tot))
        print(f"{name}: B={B} T={T}: {tot_ms:.2f} ms per batch (front-end {fe_ms:.2f} ms) = {B / tot_ms * 1e3 / 1e3:.1f} k clips/s, {B * T / 16000 / tot_ms * 1e3 / 1e3:.1f} k audio-seconds/s", flush=True)
        del m, x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
