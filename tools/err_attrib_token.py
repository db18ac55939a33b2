"""CPU: pooling 'token' -- which fp16 ACTIVATION rounding point carries the error that precision 'f16a' (split operands of the cls tile)
leaves, on the heavy-tailed checkpoint family (synth.make_structured_trained_like_state_dict)?  fp64 restatement of the block
(BNeckAttention models/uit.py:89-122 or Attention :124-178, ReLU / exact-erf GELU, cls token :389-392, 399-401) with exact weights and
switchable rounding points; rows of the cls tile (tokens 0..15) stay exact where 'f16a' carries hi + lo.

    python tools/err_attrib_token.py audio_transformer_h128_d4_m3 full 223 216 226
    python tools/err_attrib_token.py uit_xxs bneck 234

Round-5 finding (profiles/round5/err_attrib_token.log): with full-width attention the fp16 rounding of the KEYS (then the values, then the
queries) of ALL tokens is what is left -- seed 223: 3.6e-3 as 'f16a' was, 5.7e-4 with k as hi + lo, 1.1e-4 with q, k, v as hi + lo -- so
'f16a' / 'f16ax' carry q, k, v of every column tile as two fp16 terms (encoder.hip: Qlo / Klo / Vlo).  With exact activations the
nearest-rounded fp16 WEIGHTS alone leave 1e-3 .. 4e-3 on these checkpoints: the top of the ladder needs split weights too ('f16ax')."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tools")]
import lo_term_study as L  # noqa: E402
from uit_mobile_amd.synth import DEPTH, make_structured_trained_like_state_dict  # noqa: E402

F64 = torch.float64
f16 = lambda t: t.to(torch.float16).to(F64)


def f16s(t, terms):
    hi = f16(t)
    return hi if terms == 1 else hi + f16(t - hi)


def forward(sd, mel_bn, depth, pts, mus=None, rec=None, inner=128, act="gelu"):
    """pts: {point: (mode, centred)}; mode 'all' = every row plain fp16, 'rest' = rows >= 16 plain fp16 (the cls tile exact), 'x2' = hi + lo
    everywhere; '_w': True = the folded weights rounded to nearest fp16"""
    g = lambda k: torch.as_tensor(np.asarray(sd[k]), dtype=F64)
    wr = (lambda w: f16(w)) if pts.get("_w") else (lambda w: w)

    def r(name, t, b, tokdim):
        if name not in pts:
            return t
        mode, cen = pts[name]
        mu = mus[(name, b)] if cen else 0.0
        if mode == "x2":
            return f16s(t - mu, 2) + mu
        y = f16(t - mu) + mu
        if mode == "rest":
            idx = [slice(None)] * t.dim()
            idx[tokdim] = slice(0, 16)
            y = y.clone()
            y[tuple(idx)] = t[tuple(idx)]
        return y
    B, hd = mel_bn.shape[0], inner // 2
    x = mel_bn[:, :, :96].reshape(B, 4, 16, 6, 16).permute(0, 1, 3, 2, 4).reshape(B, 24, 256)
    tok = f16(x) @ g("patch_embed.proj.weight").reshape(128, 256).T + g("patch_embed.proj.bias")
    tp, fp = g("time_pos_embed")[0, :, 0, :6], g("freq_pos_embed")[0, :, :, 0]
    tok = tok + (fp.T[:, None, :] + tp.T[None, :, :]).reshape(24, 128)
    c = g("cls_token").reshape(1, 1, 128) + g("token_pos_embed").reshape(1, 1, 128)
    tok = torch.cat([c.expand(B, 1, 128), tok], 1)
    N, scale = 25, 64 ** -0.5 * 1.4426950408889634
    for b in range(depth):
        p = f"blocks.{b}."
        x0 = L.ln(tok, 1e-6)
        if rec is not None:
            rec[("ln1", b)] = x0.mean((0, 1), keepdim=True)
        xn = r("ln1", x0, b, 1)
        Wq = g(p + "attn.qkv.weight") * g(p + "norm1.weight")[None]
        Wq = torch.cat([wr(Wq[:inner] * scale) / scale, wr(Wq[inner:])])
        bq = g(p + "attn.qkv.bias") + g(p + "attn.qkv.weight") @ g(p + "norm1.bias")
        qkv = (xn @ Wq.T + bq).reshape(B, N, 3, 2, hd).permute(2, 0, 3, 1, 4)
        if rec is not None:
            rec[("k", b)] = qkv[1].mean((0, 2), keepdim=True)
            rec[("v", b)] = qkv[2].mean((0, 2), keepdim=True)
        q, k, v = r("q", qkv[0] * scale, b, 2), r("k", qkv[1], b, 2), r("v", qkv[2], b, 2)
        s = q @ k.transpose(-1, -2)
        pr = r("P", torch.exp2(s - s.max(-1, keepdim=True)[0]), b, 2)
        o0 = ((pr @ v) / pr.sum(-1, keepdim=True)).transpose(1, 2).reshape(B, N, inner)
        if rec is not None:
            rec[("O", b)] = o0.mean((0, 1), keepdim=True)
        tok = tok + r("O", o0, b, 1) @ wr(g(p + "attn.proj.weight")).T + g(p + "attn.proj.bias")
        x2 = L.ln(tok, 1e-6)
        xn = r("ln2", x2, b, 1)
        W1 = wr(g(p + "mlp.fc1.weight") * g(p + "norm2.weight")[None])
        h = xn @ W1.T + (g(p + "mlp.fc1.bias") + g(p + "mlp.fc1.weight") @ g(p + "norm2.bias"))
        h = torch.relu(h) if act == "relu" else 0.5 * h * (1.0 + torch.erf(h / 2 ** 0.5))
        tok = tok + r("hidden", h, b, 1) @ wr(g(p + "mlp.fc2.weight")).T + g(p + "mlp.fc2.bias")
    z = (L.ln(tok, 1e-6) * g("norm.weight") + g("norm.bias"))[:, 0]
    z = L.ln(z, 1e-5) * g("outputlayer.0.weight") + g("outputlayer.0.bias")
    return z @ g("outputlayer.1.weight").T + g("outputlayer.1.bias")


def main():
    name, att = sys.argv[1], sys.argv[2]
    inner = 128 if att == "full" else 32
    act = "relu" if name.startswith("uit") or name.endswith("relu") else "gelu"
    from bench import synth_batch
    g = np.load(ROOT / "tests" / "golden" / "samples_uit_xs.npz")
    x = np.concatenate([(synth_batch(24, 16000, 0, "cpu", False).numpy() * 32768).round().astype(np.int16),
                        np.stack([g[k][:16000] for k in ("water_000_pcm", "water_003_pcm", "85b877b5_nohash_0_pcm")])])
    xcal = (synth_batch(32, 16000, 5000, "cpu", False).numpy() * 32768).round().astype(np.int16)
    for seed in [int(a) for a in sys.argv[3:]]:
        sd = make_structured_trained_like_state_dict(name, seed, attention=att)
        mb, mbc = L.mel_bn_of(sd, x), L.mel_bn_of(sd, xcal)
        rec = {}
        forward(sd, mbc, DEPTH[name], {}, rec=rec, inner=inner, act=act)              # the centring means, from the calibration clips
        run = lambda pts: forward(sd, mb, DEPTH[name], pts, mus=rec, inner=inner, act=act)
        exact = run({})
        err = lambda pts: float((run(pts) - exact).abs().max())
        A = lambda mode, cen=False: (mode, cen)
        cen = att == "full"
        f16a = dict(ln1=A("rest", cen), q=A("rest"), k=A("all", True), v=A("all", True), P=A("all"), O=A("rest", cen), ln2=A("rest"), hidden=A("rest"))
        out = [f"as 'f16a' was (cls tile split, exact weights) {err(f16a):.2e}"]
        d = dict(f16a)
        for nm in ("k", "v", "q"):
            d[nm] = A("x2", f16a[nm][1])
            out.append(f"+ {nm} hi+lo {err(d):.2e}")
        out.append(f"[the same on nearest-rounded fp16 weights {err({**d, '_w': True}):.2e}]")
        for nm in ("ln1", "P", "O", "ln2", "hidden"):
            out.append(f"[+ {nm} hi+lo everywhere {err({**d, nm: A('x2', f16a[nm][1])}):.2e}]")
        print(f"{name} {att} pooling token seed {seed} |logit|max {float(exact.abs().max()):.1f}: " + "; ".join(out), flush=True)


if __name__ == "__main__":
    main()
