"""numpy fp64 restatement of the calibration statistics of uitk_calibrate_weights (test helper, CPU only).

What the library measures (csrc/calibrate.hip, layout of include/uitk.h: uitk_calib_stats_doubles): the uncentred second moments
E[x x^T] of the operand of every single-weight linear of the path -- the clamped init_bn(log-mel) patches (patch embed), the
LayerNorm-1 output (qkv), the attention output (proj), the LayerNorm-2 output (fc1), the activation output (fc2), and behind them the
per-block MEAN of the LayerNorm-1 output (the operand centring of the attention, pack.cpp) -- restated here
from the reference's arithmetic (models/uit.py:89-122, 181-248, 379-395) on the oracle's mel.  Used (1) on CPU to drive
uitk_pack_weights_calibrated, (2) on the GPU box as the reference for uitk_get_calib_stats.
"""
import numpy as np
from scipy.special import erf

from oracle.oracle import Oracle


def _ln(x, eps):
    m = x.mean(-1, keepdims=True)
    v = ((x - m) ** 2).mean(-1, keepdims=True)
    return (x - m) / np.sqrt(v + eps)


def _f16(x):
    return x.astype(np.float16).astype(np.float64)


def patches(sd, pcm):
    """-> [B, 24, 256] init_bn(dB mel, batch-global floor applied), k = mel-in-patch * 16 + frame-in-patch, token = f * 6 + t"""
    mel, _ = Oracle(sd).melspec_db(pcm)
    g = lambda k: np.asarray(sd[k], np.float64)
    s = g("init_bn.1.weight") / np.sqrt(g("init_bn.1.running_var") + 1e-5)
    x = (mel.astype(np.float64) - g("init_bn.1.running_mean")[None, :, None]) * s[None, :, None] + g("init_bn.1.bias")[None, :, None]
    B = x.shape[0]
    return x[:, :, :96].reshape(B, 4, 16, 6, 16).transpose(0, 1, 3, 2, 4).reshape(B, 24, 256)


def calib_stats(sd, pcm, depth, inner=32, act="relu", cls=False, centre=None):
    """-> (stats float64 [uitk_calib_stats_doubles], logits-free): the layout of include/uitk.h"""
    g = lambda k: np.asarray(sd[k], np.float64)
    x = patches(sd, pcm)
    B = x.shape[0]
    out, means = [], []
    if centre is None:                                                 # uitk_dev.h: operand_centring() -- full attention, GELU or pooling 'token'
        centre = inner != 32 or act != "relu" or cls

    def second_moment(a):
        a = _f16(a.reshape(-1, a.shape[-1]))
        return (a.T @ a / a.shape[0]).ravel()
    out.append(second_moment(x))
    tok = x @ g("patch_embed.proj.weight").reshape(128, 256).T + g("patch_embed.proj.bias")
    tp, fp = g("time_pos_embed")[0, :, 0, :6], g("freq_pos_embed")[0, :, :, 0]
    tok = tok + (fp.T[:, None, :] + tp.T[None, :, :]).reshape(24, 128)
    if cls:                                                            # pooling 'token' (uit.py:389-391)
        c = g("cls_token").reshape(1, 1, 128) + g("token_pos_embed").reshape(1, 1, 128)
        tok = np.concatenate([np.broadcast_to(c, (B, 1, 128)), tok], 1)
    N, hd = tok.shape[1], inner // 2
    scale = (128 // 2) ** -0.5
    for b in range(depth):
        p = f"blocks.{b}."
        xn = _ln(tok, 1e-6)
        means.append(xn.reshape(-1, 128).mean(0))                      # round 5: E[LayerNorm-1 output] per block (operand centring)
        # builds that centre this operand at run time multiply round16(x - mu): that operand's moment
        out.append(second_moment(xn - (np.float32(means[-1]).astype(np.float64) if centre else 0.0)))
        qkv = (xn * g(p + "norm1.weight") + g(p + "norm1.bias")) @ g(p + "attn.qkv.weight").T + g(p + "attn.qkv.bias")
        qkv = qkv.reshape(B, N, 3, 2, hd).transpose(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        s = q @ k.transpose(0, 1, 3, 2) * scale
        pr = np.exp(s - s.max(-1, keepdims=True))
        o = ((pr @ v) / pr.sum(-1, keepdims=True)).transpose(0, 2, 1, 3).reshape(B, N, inner)
        out.append(second_moment(o))
        tok = tok + o @ g(p + "attn.proj.weight").T + g(p + "attn.proj.bias")
        xn = _ln(tok, 1e-6)
        out.append(second_moment(xn))
        h = (xn * g(p + "norm2.weight") + g(p + "norm2.bias")) @ g(p + "mlp.fc1.weight").T + g(p + "mlp.fc1.bias")
        h = np.maximum(h, 0.0) if act == "relu" else 0.5 * h * (1.0 + erf(h / np.sqrt(2.0)))
        out.append(second_moment(h))
        tok = tok + h @ g(p + "mlp.fc2.weight").T + g(p + "mlp.fc2.bias")
    return np.concatenate(out + means)
