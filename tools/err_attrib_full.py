"""CPU: which fp16 ACTIVATION rounding point carries the error of the full-width-attention / GELU factories (audio_transformer_h128_*)?
fp64 restatement of models/uit.py:124-178 (Attention) + exact-erf GELU MLP with exact weights (= what split or calibrated weights
leave) and one rounding point switched on at a time.     python tools/err_attrib_full.py [name] [seed]"""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tools")]
import lo_term_study as L
from uit_mobile_amd.synth import DEPTH, make_trained_like_state_dict
F64 = torch.float64
f16 = lambda t: t.to(torch.float16).to(F64)


def forward(sd, mel_bn, depth, act, pts):
    g = lambda k: torch.as_tensor(np.asarray(sd[k]), dtype=F64)
    r = lambda name, t: f16(t) if name in pts else t
    B = mel_bn.shape[0]
    x = mel_bn[:, :, :96].reshape(B, 4, 16, 6, 16).permute(0, 1, 3, 2, 4).reshape(B, 24, 256)
    tok = r("patch", x) @ g("patch_embed.proj.weight").reshape(128, 256).T + g("patch_embed.proj.bias")
    tp, fp = g("time_pos_embed")[0, :, 0, :6], g("freq_pos_embed")[0, :, :, 0]
    tok = tok + (fp.T[:, None, :] + tp.T[None, :, :]).reshape(24, 128)
    scale = 64 ** -0.5 * 1.4426950408889634
    for b in range(depth):
        p = f"blocks.{b}."
        xn = r("ln1", L.ln(tok, 1e-6))
        Wq = g(p + "attn.qkv.weight") * g(p + "norm1.weight")[None]
        bq = g(p + "attn.qkv.bias") + g(p + "attn.qkv.weight") @ g(p + "norm1.bias")
        qkv = (xn @ Wq.T + bq).reshape(B, 24, 3, 2, 64).permute(2, 0, 3, 1, 4)
        q, k, v = r("q", qkv[0] * scale), r("k", qkv[1]), r("v", qkv[2])
        s = q @ k.transpose(-1, -2)
        pr = r("P", torch.exp2(s - s.max(-1, keepdim=True)[0]))
        o = r("O", ((pr @ v) / pr.sum(-1, keepdim=True)).transpose(1, 2).reshape(B, 24, 128))
        tok = tok + o @ g(p + "attn.proj.weight").T + g(p + "attn.proj.bias")
        xn = r("ln2", L.ln(tok, 1e-6))
        W1 = g(p + "mlp.fc1.weight") * g(p + "norm2.weight")[None]
        h = xn @ W1.T + (g(p + "mlp.fc1.bias") + g(p + "mlp.fc1.weight") @ g(p + "norm2.bias"))
        h = torch.relu(h) if act == "relu" else 0.5 * h * (1.0 + torch.erf(h / 2 ** 0.5))
        tok = tok + r("hidden", h) @ g(p + "mlp.fc2.weight").T + g(p + "mlp.fc2.bias")
    z = (L.ln(tok, 1e-6) * g("norm.weight") + g("norm.bias")).mean(1)
    z = L.ln(z, 1e-5) * g("outputlayer.0.weight") + g("outputlayer.0.bias")
    return z @ g("outputlayer.1.weight").T + g("outputlayer.1.bias")


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "audio_transformer_h128_d4_m3"
    seeds = [int(a) for a in sys.argv[2:]] or [206, 200, 202]
    act = "relu" if name.endswith("_relu") else "gelu"
    from bench import synth_batch
    g = np.load(ROOT / "tests" / "golden" / "samples_uit_xs.npz")
    x = np.concatenate([(synth_batch(24, 16000, 0, "cpu", False).numpy() * 32768).round().astype(np.int16),
                        np.stack([g[k][:16000] for k in ("water_000_pcm", "water_003_pcm", "85b877b5_nohash_0_pcm")])])
    pts_all = ["patch", "ln1", "q", "k", "v", "P", "O", "ln2", "hidden"]
    for seed in seeds:
        sd = make_trained_like_state_dict(name, seed, attention="full")
        mb = L.mel_bn_of(sd, x)
        exact = forward(sd, mb, DEPTH[name], act, ())
        line = [f"all {float((forward(sd, mb, DEPTH[name], act, pts_all) - exact).abs().max()):.2e}"]
        for pt in pts_all:
            line.append(f"{pt} {float((forward(sd, mb, DEPTH[name], act, (pt,)) - exact).abs().max()):.2e}")
        print(f"{name} seed {seed} |logit|max {float(exact.abs().max()):.1f}: " + "  ".join(line), flush=True)


if __name__ == "__main__":
    main()
