"""GPU: calibrated rounding end to end -- (1) the statistics uitk_calibrate_weights measures vs the numpy restatement
(tests/calib_ref.py), (2) max|dlogit| against the C oracle for precision f16 / f16c (library's own signals) / f16c (caller's clips)
/ f16l6 / f16x2 over the trained-like sweep checkpoints on HELD-OUT inputs (bench batch, synth mix, the reference's sample clips
from the committed fixture), (3) what building a calibrated handle costs.

    python tools/calib_check.py [uit_xs] [nseeds]
"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden"), str(ROOT / "tests")]
from bench import synth_batch  # noqa: E402
from calib_ref import calib_stats  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from uit_mobile_amd.synth import DEPTH, make_trained_like_state_dict, synth_pcm_i16  # noqa: E402
from uit_mobile_amd import models  # noqa: E402


def build(name, sd, prec, calib=None):
    m = getattr(models, name)(outputdim=537, target_length=102, precision=prec)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    if calib is not None:
        m.calibrate(calib)
    return m


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "uit_xs"
    nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    depth = DEPTH[name]
    full = name.startswith("audio_transformer")
    att, act, inner = ("full" if full else "bneck"), ("gelu" if full and not name.endswith("_relu") else "relu"), (128 if full else 32)
    cal = synth_pcm_i16(32, 16000, seed=4242)
    cal_d = torch.from_numpy(cal).cuda()
    # (1) statistics
    sd = make_trained_like_state_dict(name, 200, attention=att)
    m = build(name, sd, "f16c", cal_d)
    t0 = time.perf_counter()
    got = m.calib_stats()
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    want = calib_stats(sd, cal, depth, inner=inner, act=act)
    off, worst = 0, 0.0
    sizes = [("pe", 256)] + [(f"b{b}.{k}", n) for b in range(depth) for k, n in (("ln1", 128), ("att", inner), ("ln2", 128), ("hid", 384))]
    for nm, n in sizes:
        g, w = got[off:off + n * n], want[off:off + n * n]
        rel = np.linalg.norm(g - w) / np.linalg.norm(w)
        worst = max(worst, rel)
        off += n * n
    print(f"statistics vs numpy restatement: worst relative Frobenius error over {len(sizes)} matrices {worst:.2e}; "
          f"calibrated handle built in {t_build * 1e3:.0f} ms (caller's 32 clips)")
    m2 = build(name, sd, "f16c")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m2.calib_stats()
    torch.cuda.synchronize()
    print(f"calibrated handle built in {(time.perf_counter() - t0) * 1e3:.0f} ms (library's own signals)")
    m3 = build(name, sd, "f16")
    t0 = time.perf_counter()
    m3._handle(torch.device("cuda", 0))
    print(f"plain handle built in {(time.perf_counter() - t0) * 1e3:.0f} ms")
    # (2) errors on held-out inputs
    evals = {"bench": synth_batch(24, 16000, 0, "cpu", False).numpy(), "synth9": synth_pcm_i16(16, 16000, seed=9)}
    g = np.load(ROOT / "tests" / "golden" / "samples_uit_xs.npz")
    evals["samples"] = np.stack([g[k][:16000] for k in ("water_000_pcm", "water_003_pcm", "85b877b5_nohash_0_pcm")])
    rng = np.random.default_rng(5)
    t = np.arange(16000) / 16000.0
    odd = np.stack([0.5 * np.sin(2 * np.pi * 3000 * t), 0.3 * np.sign(np.sin(2 * np.pi * 200 * t)), np.zeros(16000), 1e-3 * rng.standard_normal(16000),
                    np.cumsum(rng.standard_normal(16000)) / 200.0, 0.8 * np.sin(2 * np.pi * (100 + 3000 * t) * t)])
    evals["odd"] = np.clip(np.round(odd * 32768), -32768, 32767).astype(np.int16)
    modes = [("f16", None), ("f16c", None), ("f16c+x", cal_d)] + ([] if full else [("f16l6", None)]) + [("f16x2", None)]
    res = {}
    for seed in range(200, 200 + nseeds):
        sd = make_trained_like_state_dict(name, seed, attention=att)
        o = Oracle(sd, act=act)
        refs = {k: o.forward(v) for k, v in evals.items()}
        line = []
        for mode, c in modes:
            m = build(name, sd, mode.split("+")[0], c)
            for k, v in evals.items():
                with torch.no_grad():
                    p, l = m(torch.from_numpy(np.ascontiguousarray(v)).cuda(), return_logits=True)
                dl = float(np.abs(l.cpu().numpy().reshape(refs[k]["logits"].shape) - refs[k]["logits"]).max())
                dp = float(np.abs(p.cpu().numpy() - refs[k]["probs"]).max())
                res.setdefault((mode, k), []).append((dl, dp))
            line.append(f"{mode} {res[(mode, 'bench')][-1][0]:.2e}")
            del m
        print(f"seed {seed}: " + "  ".join(line), flush=True)
    print()
    for k in evals:
        for mode, _ in modes:
            v = np.array(res[(mode, k)])
            print(f"{name} {k:8s} {mode:7s} max|dlogit| worst {v[:, 0].max():.2e} median {np.median(v[:, 0]):.2e}   max|dprob| worst {v[:, 1].max():.2e}")
        print()


if __name__ == "__main__":
    main()
