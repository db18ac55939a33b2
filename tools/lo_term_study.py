"""CPU: price candidate "cheap low-order weight term" schemes for the encoder against plain f16 and f16x2, on an fp64 restatement
of the reference arithmetic with the kernel's rounding points (same method as tools/err_attrib.py), over the sweep seeds.

    python tools/lo_term_study.py [uit_xs] [nseeds]

Schemes for a linear y = W x (x = 16-bit activation operand):
  f16      : y = f16(W) x
  x2       : y = (f16(W) + f16(W - f16(W))) x                      (precision f16x2)
  fp6      : y = f16(W) x + e2m3_block32(W - f16(W)) . e2m3(x)      (MX block-scaled 16x16x128 f8f6f4 at the f16 cost of ONE k-step)
  fp6/fp4  : the same with e2m1 activations for fc2 (hidden)
  top<k>   : y = f16(W) x + f16(W - f16(W))[:, top-k input channels] x[top-k]   (outlier-channel split; channels chosen by calibration)
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from oracle.oracle import Oracle  # noqa: E402
from uit_mobile_amd.synth import make_trained_like_state_dict, synth_pcm_i16  # noqa: E402

F64 = torch.float64
PE_SPLIT = True     # patch-embed weights as hi + lo (the product rounds them structurally, which removes most of the same term)


def f16(x):
    return x.to(torch.float16).to(F64)


def q_e2m3(v):          # OCP fp6 e2m3: 0, subnormal step .125 below 1, three mantissa bits, max 7.5; round to nearest
    a = v.abs()
    e = torch.floor(torch.log2(a.clamp_min(1e-30))).clamp(0, 2)
    step = torch.where(a < 1.0, torch.full_like(a, 0.125), torch.exp2(e - 3))
    return torch.sign(v) * (torch.round(a / step) * step).clamp_max(7.5)


def q_e2m1(v):          # OCP fp4 e2m1: 0 .5 1 1.5 2 3 4 6
    a = v.abs()
    e = torch.floor(torch.log2(a.clamp_min(1e-30))).clamp(0, 2)
    step = torch.where(a < 1.0, torch.full_like(a, 0.5), torch.exp2(e - 1))
    return torch.sign(v) * (torch.round(a / step) * step).clamp_max(6.0)


def block_q(W, q, top, blk=32):  # per (row, blk consecutive k) power-of-two scale so that the block maximum fits `top`
    O, K = W.shape
    Wb = W.reshape(O, K // blk, blk)
    mx = Wb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    sc = torch.exp2(torch.ceil(torch.log2(mx / top)))
    return (q(Wb / sc) * sc).reshape(O, K)


def ln(x, eps):
    m = x.mean(-1, keepdim=True)
    v = ((x - m) ** 2).mean(-1, keepdim=True)
    return (x - m) / torch.sqrt(v + eps)


class Scheme:
    """lin(x, W, kind, blk) -> y for kind in qkv / proj / fc1 / fc2; x is the exact fp64 activation."""

    def __init__(self, name, calib=None, topk=0, fc2_act="fp6", act_scale=2.0, which=("qkv", "proj", "fc1", "fc2"), h_scale=1.0, wblk=32):
        self.name, self.calib, self.topk, self.fc2_act, self.act_scale, self.which, self.h_scale, self.wblk = name, calib, topk, fc2_act, act_scale, which, h_scale, wblk

    def lin(self, x, W, kind, blk):
        xh, Wh = f16(x), f16(W)
        y = xh @ Wh.T
        n = self.name
        if n == "f16" or kind not in self.which:
            return y
        Wl = W - Wh
        if n == "x2" or (kind == "proj"):                         # proj (K = 32, 3 % of the MFMAs): always the full f16 lo term
            return y + xh @ f16(Wl).T
        if n.startswith("fp6"):
            if kind == "fc2" and self.fc2_act == "fp4":
                s = self.h_scale
                xq = q_e2m1(xh / s) * s
                self.sat = max(getattr(self, "sat", 0.0), float(xh.amax()))
            else:
                s = self.act_scale if kind != "fc2" else 1.0
                xq = q_e2m3(xh / s) * s
            return y + xq @ block_q(Wl, q_e2m3, 7.5, self.wblk).T
        if n.startswith("top"):
            idx = self.calib[(kind, blk)][: self.topk]
            return y + xh[..., idx] @ f16(Wl[:, idx]).T
        raise ValueError(n)


def forward(sd, mel_bn, depth, sch=None, record=None):
    g = lambda k: torch.as_tensor(np.asarray(sd[k]), dtype=F64)
    lin = (lambda x, W, kind, blk: x @ W.T) if sch is None else sch.lin
    r16 = (lambda t: t) if sch is None else f16
    B = mel_bn.shape[0]
    x = mel_bn[:, :, :96].reshape(B, 4, 16, 6, 16).permute(0, 1, 3, 2, 4).reshape(B, 24, 256)
    W = g("patch_embed.proj.weight").reshape(128, 256)
    if sch is not None and getattr(sch, "does_pe", False):                  # tools/calib_study.py: the patch embed through the scheme too
        tok = sch.lin(x, W, "pe", -1) + g("patch_embed.proj.bias")
    else:
        tok = r16(x) @ (W if PE_SPLIT else r16(W)).T + g("patch_embed.proj.bias")
    tp, fp = g("time_pos_embed")[0, :, 0, :6], g("freq_pos_embed")[0, :, :, 0]
    tok = tok + (fp.T[:, None, :] + tp.T[None, :, :]).reshape(24, 128)
    scale = (128 // 2) ** -0.5 * 1.4426950408889634
    for b in range(depth):
        p = f"blocks.{b}."
        w1, b1 = g(p + "norm1.weight"), g(p + "norm1.bias")
        Wq, bq = g(p + "attn.qkv.weight"), g(p + "attn.qkv.bias")
        Wf, bf = Wq * w1[None], bq + Wq @ b1
        Wf = torch.cat([Wf[:32] * scale, Wf[32:]]); bf = torch.cat([bf[:32] * scale, bf[32:]])
        xn = ln(tok, 1e-6)
        if record is not None: record[("ln1", b)] = xn
        qkv = r16(lin(xn, Wf, "qkv", b) + bf).reshape(B, 24, 3, 2, 16).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        s = q @ k.transpose(-1, -2)
        pr = r16(torch.exp2(s - s.max(-1, keepdim=True)[0]))
        o = ((pr @ v) / pr.sum(-1, keepdim=True)).transpose(1, 2).reshape(B, 24, 32)
        tok = tok + lin(o, g(p + "attn.proj.weight"), "proj", b) + g(p + "attn.proj.bias")
        w2, b2 = g(p + "norm2.weight"), g(p + "norm2.bias")
        W1, c1 = g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias")
        xn = ln(tok, 1e-6)
        if record is not None: record[("ln2", b)] = xn
        h = torch.relu(lin(xn, W1 * w2[None], "fc1", b) + (c1 + W1 @ b2))
        if record is not None: record[("h", b)] = h
        tok = tok + lin(h, g(p + "mlp.fc2.weight"), "fc2", b) + g(p + "mlp.fc2.bias")
    z = (ln(tok, 1e-6) * g("norm.weight") + g("norm.bias")).mean(1)
    z = ln(z, 1e-5) * g("outputlayer.0.weight") + g("outputlayer.0.bias")
    return z @ g("outputlayer.1.weight").T + g("outputlayer.1.bias")


def mel_bn_of(sd, pcm):
    o = Oracle(sd)
    mel, _ = o.melspec_db(pcm)
    s = np.asarray(sd["init_bn.1.weight"], np.float64) / np.sqrt(np.asarray(sd["init_bn.1.running_var"], np.float64) + 1e-5)
    return torch.from_numpy((mel.astype(np.float64) - np.asarray(sd["init_bn.1.running_mean"], np.float64)[None, :, None]) * s[None, :, None]
                            + np.asarray(sd["init_bn.1.bias"], np.float64)[None, :, None])


def calibrate(sd, depth, global_perm=True):
    """channel ranking from a calibration batch (noise / tone / near-silence clips, NOT the evaluation clips)."""
    pcm = synth_pcm_i16(8, 16000, seed=77)
    rec = {}
    forward(sd, mel_bn_of(sd, pcm), depth, None, rec)
    cal = {}
    e_ln = sum((rec[("ln1", b)] ** 2).mean((0, 1)) + (rec[("ln2", b)] ** 2).mean((0, 1)) for b in range(depth))
    order = torch.argsort(e_ln, descending=True)
    for b in range(depth):
        cal[("qkv", b)] = order if global_perm else torch.argsort((rec[("ln1", b)] ** 2).mean((0, 1)), descending=True)
        cal[("fc1", b)] = order if global_perm else torch.argsort((rec[("ln2", b)] ** 2).mean((0, 1)), descending=True)
        cal[("fc2", b)] = torch.argsort((rec[("h", b)] ** 2).mean((0, 1)), descending=True)
    frac = float(e_ln[order[:32]].sum() / e_ln.sum())
    return cal, frac


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "uit_xs"
    nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    depth = {"uit_xs": 12, "uit_xxs": 6, "uit_xxxs": 4}[name]
    from bench import synth_batch
    x = synth_batch(24, 16000, 0, "cpu", False).numpy()
    res = {}
    for seed in range(200, 200 + nseeds):
        sd = make_trained_like_state_dict(name, seed)
        mb = mel_bn_of(sd, x)
        exact = forward(sd, mb, depth)
        cal, frac = calibrate(sd, depth)
        kw = dict(fc2_act="fp4", h_scale=2.0, wblk=128)
        schemes = [Scheme("f16"), Scheme("x2"), Scheme("fp6 all linears (fp6/fp4 s2 row)", **kw),
                   Scheme("fp6 mlp+proj = f16l6", which=("fc1", "fc2", "proj"), **kw),
                   Scheme("fp6 fc2+proj", which=("fc2", "proj"), **kw), Scheme("fp6 fc1+proj", which=("fc1", "proj"), **kw),
                   Scheme("fp6 proj only", which=("proj",), **kw), Scheme("fp6 qkv+fc2+proj", which=("qkv", "fc2", "proj"), **kw)]
        line = []
        for s in schemes:
            e = float((forward(sd, mb, depth, s) - exact).abs().max())
            res.setdefault(s.name if not s.name.startswith("top") or "mlp" in s.name else s.name, []).append(e)
            line.append(f"{s.name} {e:.2e}" + (f" (h max {s.sat:.1f})" if hasattr(s, "sat") and s.name.endswith("s1") else ""))
        print(f"seed {seed} |logit|max {float(exact.abs().max()):.1f} top-32 LN energy {frac:.2f}: " + "  ".join(line), flush=True)
    print()
    for k, v in res.items():
        v = np.array(v)
        print(f"{name} {k:16s} worst {v.max():.2e}  median {np.median(v):.2e}")


if __name__ == "__main__":
    main()
