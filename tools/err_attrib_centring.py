"""CPU: which fp16 ACTIVATION rounding point carries the error, and what operand centring buys (round 5).  fp64 restatement of the
block (BNeckAttention models/uit.py:89-122 or Attention :124-178, ReLU / exact-erf GELU MLP, pooling 'mean' / 'token') with EXACT weights
(= what calibrated or split weights leave) and one fp16 rounding point switched on at a time -- plain, or "centred": the value minus
its calibration mean (means from OTHER clips) is rounded, the mean added back exactly (what pack.cpp's bias folds + the kernel's
LayerNorm-1 centring implement).

    python tools/err_attrib_centring.py audio_transformer_h128_d4_m3 full mean 206
    python tools/err_attrib_centring.py uit_xs bneck token 4

Round-5 findings (profiles/round5/err_attrib_centring.log): d4-GELU seed 206 (the round-4 gate case): all points 9.8e-4 -> all centred
2.1e-4; LayerNorm-1 alone 6.3e-4 -> 1.0e-4 (a massive-activation channel is ~10 in EVERY token: rounded as it is, its fp16 error is the
same in every token and survives the mean pool)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tools")]
import lo_term_study as L  # noqa: E402
from uit_mobile_amd.synth import DEPTH, make_trained_like_state_dict  # noqa: E402

F64 = torch.float64
f16 = lambda t: t.to(torch.float16).to(F64)


def forward(sd, mel_bn, depth, act, att, pooling, pts, mus=None, rec=None, cls=None):
    g = lambda k: torch.as_tensor(np.asarray(sd[k]), dtype=F64)

    def r(name, t, b=None):
        if name in pts:
            return f16(t)
        if name + "c" in pts:
            mu = mus[(name, b)]
            return f16(t - mu) + mu
        return t
    B = mel_bn.shape[0]
    inner = 128 if att == "full" else 32
    hd = inner // 2
    x = mel_bn[:, :, :96].reshape(B, 4, 16, 6, 16).permute(0, 1, 3, 2, 4).reshape(B, 24, 256)
    tok = r("patch", x) @ g("patch_embed.proj.weight").reshape(128, 256).T + g("patch_embed.proj.bias")
    tp, fp = g("time_pos_embed")[0, :, 0, :6], g("freq_pos_embed")[0, :, :, 0]
    tok = tok + (fp.T[:, None, :] + tp.T[None, :, :]).reshape(24, 128)
    if pooling == "token":
        c = (torch.as_tensor(cls, dtype=F64) if cls is not None else g("cls_token")).reshape(1, 1, 128) + g("token_pos_embed").reshape(1, 1, 128)
        tok = torch.cat([c.expand(B, 1, 128), tok], 1)
    N = tok.shape[1]
    scale = 64 ** -0.5 * 1.4426950408889634
    for b in range(depth):
        p = f"blocks.{b}."
        x0 = L.ln(tok, 1e-6)
        if rec is not None:
            rec[("ln1", b)] = x0.mean((0, 1), keepdim=True)
        xn = r("ln1", x0, b)
        Wq = g(p + "attn.qkv.weight") * g(p + "norm1.weight")[None]
        bq = g(p + "attn.qkv.bias") + g(p + "attn.qkv.weight") @ g(p + "norm1.bias")
        qkv = (xn @ Wq.T + bq).reshape(B, N, 3, 2, hd).permute(2, 0, 3, 1, 4)
        if rec is not None:
            rec[("k", b)] = qkv[1].mean((0, 2), keepdim=True)
            rec[("v", b)] = qkv[2].mean((0, 2), keepdim=True)
        q, k, v = r("q", qkv[0] * scale), r("k", qkv[1], b), r("v", qkv[2], b)
        s = q @ k.transpose(-1, -2)
        pr = r("P", torch.exp2(s - s.max(-1, keepdim=True)[0]))
        o0 = ((pr @ v) / pr.sum(-1, keepdim=True)).transpose(1, 2).reshape(B, N, inner)
        if rec is not None:
            rec[("O", b)] = o0.mean((0, 1), keepdim=True)
        tok = tok + r("O", o0, b) @ g(p + "attn.proj.weight").T + g(p + "attn.proj.bias")
        x2 = L.ln(tok, 1e-6)
        if rec is not None:
            rec[("ln2", b)] = x2.mean((0, 1), keepdim=True)
        xn = r("ln2", x2, b)
        W1 = g(p + "mlp.fc1.weight") * g(p + "norm2.weight")[None]
        h = xn @ W1.T + (g(p + "mlp.fc1.bias") + g(p + "mlp.fc1.weight") @ g(p + "norm2.bias"))
        h = torch.relu(h) if act == "relu" else 0.5 * h * (1.0 + torch.erf(h / 2 ** 0.5))
        if rec is not None:
            rec[("hidden", b)] = h.mean((0, 1), keepdim=True)
        tok = tok + r("hidden", h, b) @ g(p + "mlp.fc2.weight").T + g(p + "mlp.fc2.bias")
    z = L.ln(tok, 1e-6) * g("norm.weight") + g("norm.bias")
    z = z[:, 0] if pooling == "token" else z.mean(1)
    z = L.ln(z, 1e-5) * g("outputlayer.0.weight") + g("outputlayer.0.bias")
    return z @ g("outputlayer.1.weight").T + g("outputlayer.1.bias")


def main():
    name, att, pooling = sys.argv[1], sys.argv[2], sys.argv[3]
    seeds = [int(a) for a in sys.argv[4:]]
    act = "relu" if name.endswith("_relu") or name.startswith("uit") else "gelu"
    from bench import synth_batch
    g = np.load(ROOT / "tests" / "golden" / "samples_uit_xs.npz")
    x = np.concatenate([(synth_batch(24, 16000, 0, "cpu", False).numpy() * 32768).round().astype(np.int16),
                        np.stack([g[k][:16000] for k in ("water_000_pcm", "water_003_pcm", "85b877b5_nohash_0_pcm")])])
    xcal = (synth_batch(32, 16000, 5000, "cpu", False).numpy() * 32768).round().astype(np.int16)
    cls = 0.3 * torch.randn(1, 1, 128, generator=torch.Generator().manual_seed(7)).numpy() if pooling == "token" else None
    plain = ["patch", "ln1", "q", "k", "v", "P", "O", "ln2", "hidden"]
    for seed in seeds:
        sd = make_trained_like_state_dict(name, seed, attention=att)
        mb, mbc = L.mel_bn_of(sd, x), L.mel_bn_of(sd, xcal)
        rec = {}
        forward(sd, mbc, DEPTH[name], act, att, pooling, (), rec=rec, cls=cls)          # means from the calibration clips
        run = lambda pts: forward(sd, mb, DEPTH[name], act, att, pooling, pts, mus=rec, cls=cls)
        exact = run(())
        err = lambda pts: float((run(pts) - exact).abs().max())
        line = [f"ALL {err(plain):.2e}", f"ALL(product: ln1, k, v, O centred) {err(['patch', 'ln1c', 'q', 'kc', 'vc', 'P', 'Oc', 'ln2', 'hidden']):.2e}",
                f"ALL(+ ln2, hidden centred) {err(['patch', 'ln1c', 'q', 'kc', 'vc', 'P', 'Oc', 'ln2c', 'hiddenc']):.2e}"]
        for pt in plain:
            line.append(f"{pt} {err([pt]):.2e}" + (f" -> {err([pt + 'c']):.2e}" if (pt, 0) in rec else ""))
        print(f"{name} {att} pooling {pooling} seed {seed} |logit|max {float(exact.abs().max()):.1f}: " + "  ".join(line), flush=True)


if __name__ == "__main__":
    main()
