"""CPU: price PACK-TIME ways of buying the parity margin for single fp16 weights -- nothing here costs a
cycle at run time:

  rtn      : round to nearest (+ the product's data-free structured rounding of conv / fc2 is NOT modelled here: PE is split)
  bc       : rtn + bias correction: b += (W - f16(W)) mu, mu = calibration mean of the linear's input operand (per channel)
  gptq     : Hessian-aware rounding of every weight row against H = E[x x^T] of a calibration batch (greedy column-by-column
             rounding with error feedback through the Cholesky factor of H^-1, i.e. GPTQ on the fp16 grid), uncentred H
  gptq+bc  : the same against the centred covariance, + bias correction for the mean

on the fp64 restatement of tools/lo_term_study.py (the kernel's rounding points), calibrated on ONE input set and scored on
DISJOINT ones (bench noise batch, synth tones / silence, the reference's sample clips when /root/reference is present).

    python tools/calib_study.py [uit_xs] [nseeds] [ncal]
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden"), str(ROOT / "tools")]
import lo_term_study as L  # noqa: E402
from uit_mobile_amd.synth import make_trained_like_state_dict, synth_pcm_i16  # noqa: E402

F64 = torch.float64
f16 = L.f16


def gptq_f16(W, H, damp=1e-2):
    """rows of W [O, K] rounded to fp16 one column at a time; the rounding error of column j is pushed onto the columns not yet
    rounded so that (W - Q) x stays small for x ~ H.  Columns are visited in order of decreasing H_jj ("act order")."""
    K = W.shape[1]
    H = H.clone()
    d = torch.diagonal(H)
    dead = d <= 0
    H[dead, dead] = 1.0
    perm = torch.argsort(torch.diagonal(H), descending=True)
    Wp, Hp = W[:, perm].clone(), H[perm][:, perm]
    Hp = Hp + damp * torch.mean(torch.diagonal(Hp)) * torch.eye(K, dtype=F64)
    Hinv = torch.cholesky_inverse(torch.linalg.cholesky(Hp))
    U = torch.linalg.cholesky(Hinv, upper=True)                # Hinv = U^T U, rows of U drive the feedback
    Q = torch.zeros_like(Wp)
    for j in range(K):
        q = f16(Wp[:, j])
        Q[:, j] = q
        e = (Wp[:, j] - q) / U[j, j]
        Wp[:, j + 1:] -= e[:, None] * U[j, j + 1:][None, :]
    out = torch.zeros_like(Q)
    out[:, perm] = Q
    return out


class Calibrated:
    """lin() of lo_term_study.forward with per-(kind, block) pre-rounded weights and bias corrections."""

    def __init__(self, mode, stats, damp=1e-2, does_pe=True):
        self.mode, self.stats, self.damp, self.cache, self.does_pe = mode, stats, damp, {}, does_pe

    def lin(self, x, W, kind, blk):
        key = (kind, blk)
        if key not in self.cache:
            mu, S2 = self.stats[key]                            # mean [K], uncentred second moment [K, K] of the f16 operand
            if self.mode == "rtn":
                Q, db = f16(W), None
            elif self.mode == "bc":
                Q = f16(W); db = (W - Q) @ mu
            elif self.mode == "gptq":
                Q, db = gptq_f16(W, S2, self.damp), None
            elif self.mode == "gptq+bc":
                Q = gptq_f16(W, S2 - torch.outer(mu, mu), self.damp); db = (W - Q) @ mu
            elif self.mode == "gptqU+bc":                       # uncentred H (mean direction weighted in) AND the bias correction
                Q = gptq_f16(W, S2, self.damp); db = (W - Q) @ mu
            else:
                raise ValueError(self.mode)
            self.cache[key] = (Q, db)
        Q, db = self.cache[key]
        y = f16(x) @ Q.T
        return y if db is None else y + db


def record_stats(sd, mb, depth):
    """operand statistics of every linear on the EXACT network (errors are 1e-3 relative: exact vs quantised prefix is immaterial)"""
    rec = {}

    class Rec:
        does_pe = True

        def lin(self, x, W, kind, blk):
            xh = f16(x).reshape(-1, x.shape[-1])
            rec[(kind, blk)] = (xh.mean(0), xh.T @ xh / xh.shape[0])
            return x @ W.T
    L.forward.__globals__["f16"] = lambda t: t                  # exact forward with the recording hook
    try:
        L.forward(sd, mb, depth, Rec())
    finally:
        L.forward.__globals__["f16"] = f16
    return rec


def sample_clips():
    d = Path("/root/reference/samples")
    if not d.exists():
        return None
    sys.path.insert(0, str(ROOT))
    from uit_mobile_amd.wav import read_wav_pcm16
    out = []
    for p in sorted(d.glob("*.wav")):
        x, _ = read_wav_pcm16(str(p))
        x = x[:16000]
        out.append(np.pad(x, (0, 16000 - len(x))))
    return np.stack(out).astype(np.int16)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "uit_xs"
    nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    ncal = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    depth = {"uit_xs": 12, "uit_xxs": 6, "uit_xxxs": 4}[name]
    from bench import synth_batch
    samples = sample_clips()
    # calibration set: synthetic mix (noise / tones / silence) with seeds the evaluation never uses (+ half of the sample clips)
    calset = sys.argv[4] if len(sys.argv) > 4 else "synth+samples"      # synth | synth+samples | samples | noise
    damp = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-2
    cal_pcm = []
    if "synth" in calset:
        cal_pcm.append(synth_pcm_i16(ncal, 16000, seed=4242))
    if calset == "noise":
        cal_pcm.append(np.clip(np.round((np.random.default_rng(3).random((ncal, 16000)) - 0.5) * 32768), -32768, 32767).astype(np.int16))
    if "samples" in calset and samples is not None:
        cal_pcm.append(samples[0::2])
    cal_pcm = np.concatenate(cal_pcm)
    print(f"calibration set: {calset}, {len(cal_pcm)} clips, damp {damp}")
    evals = {"bench": synth_batch(24, 16000, 0, "cpu", False).numpy(), "synth": synth_pcm_i16(16, 16000, seed=9)}
    if samples is not None:
        evals["samples(held out)"] = samples[1::2] if "samples" in calset else samples
    rng = np.random.default_rng(5)
    t = np.arange(16000) / 16000.0
    odd = np.stack([0.5 * np.sin(2 * np.pi * 3000 * t), 0.3 * np.sign(np.sin(2 * np.pi * 200 * t)), np.zeros(16000), 1e-3 * rng.standard_normal(16000),
                    np.cumsum(rng.standard_normal(16000)) / 200.0, 0.8 * np.sin(2 * np.pi * (100 + 3000 * t) * t)])
    evals["odd (tone/square/silence/chirp)"] = np.clip(np.round(odd * 32768), -32768, 32767).astype(np.int16)
    modes = ["rtn", "bc", "gptq", "gptq+bc", "gptqU+bc"]
    res = {}
    for seed in range(200, 200 + nseeds):
        sd = make_trained_like_state_dict(name, seed)
        stats = record_stats(sd, L.mel_bn_of(sd, cal_pcm), depth)
        sch = {m: Calibrated(m, stats, damp) for m in modes}
        sch["x2"] = L.Scheme("x2")
        sch["f16l6"] = L.Scheme("fp6 mlp+proj = f16l6", which=("fc1", "fc2", "proj"), fc2_act="fp4", h_scale=2.0, wblk=128)
        line = []
        for en, pcm in evals.items():
            mb = L.mel_bn_of(sd, pcm)
            exact = L.forward(sd, mb, depth)
            for m, s in sch.items():
                e = float((L.forward(sd, mb, depth, s) - exact).abs().max())
                res.setdefault((m, en), []).append(e)
                if en == "bench":
                    line.append(f"{m} {e:.2e}")
        print(f"seed {seed}: " + "  ".join(line), flush=True)
    print()
    for en in evals:
        for m in list(modes) + ["x2", "f16l6"]:
            v = np.array(res[(m, en)])
            print(f"{name} {en:32s} {m:9s} worst {v.max():.2e}  median {np.median(v):.2e}")
        print()


if __name__ == "__main__":
    main()
