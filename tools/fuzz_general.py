"""GPU: random UITBase geometries through the library's GENERAL engine (csrc/general.hip) against the C oracle (pinned for these kwargs by
tests/golden/general_*.npz).  Every case draws embed_dim / heads / mlp_ratio / depth / attention / activation / pooling / patch size /
n_mels / n_fft (powers of two and not) / win / hop / center / target_length / eval_avg / f_min / f_max, a batch and a clip length.

    python tools/fuzz_general.py [cases=40] [seed=0]      -> one line per case, a summary, exit code 1 on a failure

Gate: 1e-3 on logits and probabilities (the task's tolerance); expected ~1e-5 (fp32 against fp32), a few 1e-4 on full-scale pure tones."""
import sys
import time
import warnings
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tests" / "golden")]
from oracle.oracle import Oracle  # noqa: E402
from uit_mobile_amd.models.uit import UITBase  # noqa: E402
from uit_mobile_amd.synth import make_general_state_dict, synth_pcm_i16  # noqa: E402


def draw(rng):
    P = int(rng.choice([4, 8, 16, 16]))
    att = str(rng.choice(["bneck", "full"]))
    dim = int(rng.choice([64, 128, 192] if att == "bneck" else [32, 48, 64, 96, 128, 192]))
    inner = dim if att == "full" else dim // 4
    heads = int(rng.choice([h for h in (1, 2, 3, 4, 6, 8) if dim % h == 0 and inner % h == 0]))
    n_fft = int(rng.choice([256, 400, 512, 512, 640, 1024]))
    win = int(rng.choice([n_fft, n_fft, n_fft - 112]))
    n_mels = int(rng.choice([m for m in (24, 32, 40, 64, 80, 128) if m >= P and m <= n_fft // 4]))
    tl = int(rng.choice([32, 64, 102, 102, 160]))
    kw = dict(outputdim=int(rng.integers(3, 60)), patch_size=P, patch_stride=P, embed_dim=dim, depth=int(rng.integers(1, 4)), num_heads=heads,
              mlp_ratio=float(rng.choice([1.0, 2.0, 3.0, 4.0])), pooling=str(rng.choice(["mean", "token", "dm"])), target_length=tl,
              n_mels=n_mels, n_fft=n_fft, win_size=win, hop_size=int(rng.choice([100, 128, 160, 200])), center=bool(rng.integers(2)),
              eval_avg=str(rng.choice(["mean", "max"])), f_min=int(rng.choice([0, 0, 60])), f_max=int(rng.choice([8000, 8000, 7000])))
    act = str(rng.choice(["relu", "gelu"]))
    return kw, att, act


def run(cases=40, seed=0, budget_s=None, verbose=True):
    rng = np.random.default_rng(seed)
    warnings.simplefilter("ignore")
    t0, rows, bad = time.time(), [], []
    for i in range(cases):
        if budget_s is not None and time.time() - t0 > budget_s:
            break
        kw, att, act = draw(rng)
        hop, tl, P = kw["hop_size"], kw["target_length"], kw["patch_size"]
        kind = str(rng.choice(["short", "full", "crops"]))
        nF = {"short": int(rng.integers(P, max(P + 1, tl))), "full": tl, "crops": int(rng.integers(tl + 1, 3 * tl))}[kind]
        T = (nF - 1) * hop + (0 if kw["center"] else kw["n_fft"]) + int(rng.integers(0, hop))
        T = max(T, kw["n_fft"] // 2 + 2, kw["n_fft"] if not kw["center"] else 0)
        B = int(rng.choice([1, 2, 3, 5, 9]))
        i16 = bool(rng.integers(2))
        sd = make_general_state_dict(embed_dim=kw["embed_dim"], depth=kw["depth"], mlp_ratio=kw["mlp_ratio"], attention=att, n_mels=kw["n_mels"],
                                     n_fft=kw["n_fft"], win_size=kw["win_size"], f_min=float(kw["f_min"]), f_max=float(kw["f_max"]), patch_size=P,
                                     target_length=tl, outputdim=kw["outputdim"], seed=int(rng.integers(1 << 16)))
        pcm = synth_pcm_i16(B, T, seed=int(rng.integers(1 << 20)))
        what = f"{i:3d} D{kw['embed_dim']} h{kw['num_heads']} x{kw['mlp_ratio']:.0f} d{kw['depth']} {att:5s} {act} {kw['pooling']:5s} P{P} mels{kw['n_mels']} fft{kw['n_fft']}/{kw['win_size']}/{hop}" \
               f" {'c' if kw['center'] else 'n'} tl{tl} {kw['eval_avg']} {kind:5s} T={T} B={B} {'i16' if i16 else 'f32'}"
        try:
            m = UITBase(**kw, attention_type={"bneck": "BNeckAttention", "full": "Attention"}[att], act_layer={"relu": nn.ReLU, "gelu": nn.GELU}[act])
            m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
            m = m.cuda().eval()
            x = torch.from_numpy(pcm if i16 else pcm.astype(np.float32) / 32768.0).cuda()
            with torch.no_grad():
                probs, logits = (m(x), None) if kw["pooling"] == "dm" else m(x, return_logits=True)
            probs = probs.cpu().numpy()
            ref = Oracle(sd, target_length=tl, eval_avg=kw["eval_avg"], hop=hop, heads=kw["num_heads"], act=act, pooling=kw["pooling"], center=kw["center"]).forward(pcm)
            d = float(np.abs(probs - ref["probs"]).max())
            if logits is not None:
                d = max(d, float(np.abs(logits.cpu().numpy() - ref["logits"]).max()))
            ok = m.general == (not _published(kw)) and bool(np.isfinite(probs).all()) and d < 1e-3
            row = f"{what}: {'general' if m.general else 'mfma   '} max|d| {d:.2e} {'ok' if ok else 'FAIL'}"
        except Exception as e:
            ok, row = False, f"{what}: {type(e).__name__}: {e}"[:300]
        rows.append(row)
        if verbose:
            print(row, flush=True)
        if not ok:
            bad.append(row)
    print(f"{len(rows)} cases in {time.time() - t0:.0f} s, {len(bad)} failures")
    return rows, bad


def _published(kw):
    return (kw["embed_dim"] == 128 and kw["num_heads"] == 2 and kw["mlp_ratio"] == 3.0 and kw["n_mels"] == 64 and kw["n_fft"] == 512 and kw["patch_size"] == 16
            and kw["target_length"] // 16 == 6 and kw["win_size"] <= 512 and kw["hop_size"] <= 512)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    _, bad = run(n, s)
    sys.exit(1 if bad else 0)
