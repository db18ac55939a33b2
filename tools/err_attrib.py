"""CPU (build container or GPU box host): attribute the 16-bit-operand error of the encoder design to its components by
emulating the kernel's ROUNDING POINTS on top of an fp64 restatement of the reference arithmetic.

    python tools/err_attrib.py [uit_xs|uit_xxxs] [trained|init|scale3] [f16|bf16]

Rounding points (csrc/encoder.hip): patch-embed operands (normalised mel, weights); per block: LN output (affine folded
into the next linear by pack.cpp) and the folded qkv / fc1 weights, Q (log2-domain scale folded), K, V, softmax probabilities,
attention output and proj weights, ReLU(hidden) and fc2 weights; head (pooled LN output and weights).  `split` = hi+lo
operands (three MFMAs) = treated as exact here.  Everything else is fp64, so the numbers isolate operand rounding.
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from oracle.oracle import Oracle  # noqa: E402
from uit_mobile_amd.synth import make_state_dict, make_trained_like_state_dict, synth_pcm_i16  # noqa: E402


def rnd(x, mode):
    if mode is None:
        return x
    t = torch.as_tensor(x, dtype=torch.float64)
    if mode == "f16":
        return t.to(torch.float16).to(torch.float64)
    if mode == "bf16":
        return t.to(torch.bfloat16).to(torch.float64)
    if mode == "fp8":
        return t.to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64)
    raise ValueError(mode)


def ln(x, eps):
    m = x.mean(-1, keepdim=True)
    v = ((x - m) ** 2).mean(-1, keepdim=True)
    return (x - m) / torch.sqrt(v + eps)


def forward(sd, mel_bn, depth, pe=None, blk=None, head=None, attn=None, parts=("ln", "w", "qkv", "p", "o", "h")):
    """mel_bn: [B,64,>=96] init_bn output (fp64).  pe/blk/head/attn: rounding mode or None (= exact / split)."""
    g = lambda k: torch.as_tensor(np.asarray(sd[k]), dtype=torch.float64)
    attn = blk if attn is None else attn
    B = mel_bn.shape[0]
    x = mel_bn[:, :, :96].reshape(B, 4, 16, 6, 16).permute(0, 1, 3, 2, 4).reshape(B, 24, 256)      # token f*6+t, k = fi*16+ti
    W = g("patch_embed.proj.weight").reshape(128, 256)
    tok = rnd(x, pe) @ rnd(W, pe).T + g("patch_embed.proj.bias")
    tp, fp = g("time_pos_embed")[0, :, 0, :6], g("freq_pos_embed")[0, :, :, 0]
    pos = (fp.T[:, None, :] + tp.T[None, :, :]).reshape(24, 128)
    tok = tok + pos
    scale = (128 // 2) ** -0.5 * 1.4426950408889634
    for b in range(depth):
        p = f"blocks.{b}."
        w1, b1 = g(p + "norm1.weight"), g(p + "norm1.bias")
        Wq, bq = g(p + "attn.qkv.weight"), g(p + "attn.qkv.bias")
        Wf, bf = Wq * w1[None], bq + Wq @ b1
        Wf = torch.cat([Wf[:32] * scale, Wf[32:]]); bf = torch.cat([bf[:32] * scale, bf[32:]])
        xh = rnd(ln(tok, 1e-6), blk if "ln" in parts else None)
        qkv = xh @ rnd(Wf, blk if ("w" in parts or "w_qkv" in parts) else None).T + bf
        qkv = rnd(qkv, attn if "qkv" in parts else None).reshape(B, 24, 3, 2, 16).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        s = q @ k.transpose(-1, -2)
        pr = torch.exp2(s - s.max(-1, keepdim=True)[0])
        pr = rnd(pr, attn if "p" in parts else None)
        o = (pr @ v) / pr.sum(-1, keepdim=True)
        o = o.transpose(1, 2).reshape(B, 24, 32)
        tok = tok + rnd(o, blk if "o" in parts else None) @ rnd(g(p + "attn.proj.weight"), blk if ("w" in parts or "w_proj" in parts) else None).T + g(p + "attn.proj.bias")
        w2, b2 = g(p + "norm2.weight"), g(p + "norm2.bias")
        W1, c1 = g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias")
        h = rnd(ln(tok, 1e-6), blk if "ln" in parts else None) @ rnd(W1 * w2[None], blk if ("w" in parts or "w_fc1" in parts) else None).T + (c1 + W1 @ b2)
        h = rnd(torch.relu(h), blk if "h" in parts else None)
        tok = tok + h @ rnd(g(p + "mlp.fc2.weight"), blk if ("w" in parts or "w_fc2" in parts) else None).T + g(p + "mlp.fc2.bias")
    z = (ln(tok, 1e-6) * g("norm.weight") + g("norm.bias")).mean(1)
    z = ln(z, 1e-5) * g("outputlayer.0.weight") + g("outputlayer.0.bias")
    return rnd(z, head) @ rnd(g("outputlayer.1.weight"), head).T + g("outputlayer.1.bias"), tok


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "uit_xs"
    kind = sys.argv[2] if len(sys.argv) > 2 else "trained"
    mode = sys.argv[3] if len(sys.argv) > 3 else "f16"
    depth = {"uit_xs": 12, "uit_xxs": 6, "uit_xxxs": 4}[name]
    sd = make_trained_like_state_dict(name, 1) if kind == "trained" else make_state_dict(name, 13, 3.0 if kind == "scale3" else 1.0)
    pcm = synth_pcm_i16(8, 16000, seed=0)
    pcm[5] = 0
    o = Oracle(sd)
    mel, _ = o.melspec_db(pcm)
    s = np.asarray(sd["init_bn.1.weight"], np.float64) / np.sqrt(np.asarray(sd["init_bn.1.running_var"], np.float64) + 1e-5)
    mel_bn = torch.from_numpy((mel.astype(np.float64) - np.asarray(sd["init_bn.1.running_mean"], np.float64)[None, :, None]) * s[None, :, None]
                              + np.asarray(sd["init_bn.1.bias"], np.float64)[None, :, None])
    exact, tok0 = forward(sd, mel_bn, depth)
    ref = o.forward(pcm)["logits"][:, 0]
    print(f"{name} {kind} {mode}: logit range {exact.min():.2f} .. {exact.max():.2f}; fp64 restatement vs C oracle {np.abs(exact.numpy() - ref).max():.1e}")

    def err(**kw):
        l, tok = forward(sd, mel_bn, depth, **kw)
        return float((l - exact).abs().max()), float(((tok - tok0).abs().max(-1)[0] / tok0.abs().max(-1)[0]).max())

    rows = [("everything rounded (plain 16-bit operands)", dict(pe=mode, blk=mode, head=mode)),
            ("head split (round-1 kernel)", dict(pe=mode, blk=mode)),
            ("patch embed only", dict(pe=mode)),
            ("blocks only (= head + patch embed split)", dict(blk=mode)),
            ("head only", dict(head=mode)),
            ("blocks: LN outputs + folded weights of qkv/fc1/proj/fc2 only", dict(blk=mode, parts=("ln", "w"))),
            ("blocks: weights only", dict(blk=mode, parts=("w",))),
            ("blocks: qkv weights only", dict(blk=mode, parts=("w_qkv",))),
            ("blocks: proj weights only", dict(blk=mode, parts=("w_proj",))),
            ("blocks: fc1 weights only", dict(blk=mode, parts=("w_fc1",))),
            ("blocks: fc2 weights only", dict(blk=mode, parts=("w_fc2",))),
            ("blocks: everything but fc1/fc2 weights", dict(blk=mode, parts=("ln", "w_qkv", "w_proj", "qkv", "p", "o", "h"))),
            ("blocks: LN outputs only", dict(blk=mode, parts=("ln",))),
            ("blocks: q,k,v only", dict(blk=mode, parts=("qkv",))),
            ("blocks: softmax probabilities only", dict(blk=mode, parts=("p",))),
            ("blocks: attention output only", dict(blk=mode, parts=("o",))),
            ("blocks: relu(hidden) only", dict(blk=mode, parts=("h",)))]
    if mode != "f16":
        rows.append((f"blocks {mode} with f16 attention internals (q,k,v,P)", dict(blk=mode, attn="f16")))
    for label, kw in rows:
        e, t = err(**kw)
        print(f"  {label:66s} max|dlogit| {e:.2e}   max rel token err {t:.1e}")


if __name__ == "__main__":
    main()
