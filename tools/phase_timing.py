"""GPU diagnostic (needs a -DUITK_LAB library: tools/mklib.sh lab -DUITK_LAB; UITK_LIB_PATH=tools/ab/libuitk_lab.so): where do an
encoder block's cycles go?  (instrumented kernel build, s_memtime stamps)"""
import ctypes, os, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from uit_mobile_amd.synth import make_state_dict
from uit_mobile_amd import models, _lib
from bench import synth_batch
dev = torch.device("cuda", 0)
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xs", 13).items()}
x = synth_batch(8192, 16000, 0, dev, False)
names = ["LN1", "wait QK", "QK", "wait VP", "attn+proj", "LN2", "wait MLP(12)", "MLP(12)"]
names2 = ["load wait", "prologue barrier", "patch embed", "epilogue total", "stream kick", "load issue", "clamp",
          "final LN+pool", "head LN+LDS", "epi barrier", "z reads", "head tiles+stores"]
for waves, B in (("8", 8192),):
    m = models.uit_xs(outputdim=537, target_length=102, precision="f16").to(dev); m.load_state_dict(sd); m.eval()
    m.set_option("enc_waves", int(waves))
    xb = x[:B]
    m(xb); torch.cuda.synchronize()
    nw = B // 2
    buf = torch.zeros((nw, 20), dtype=torch.int64, device=dev)
    h = m._handle(dev)
    fbuf = torch.zeros((4096, 8), dtype=torch.int64, device=dev)
    _lib.check(_lib.load().uitk_set_timing_buffers(h, ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(fbuf.data_ptr())), "timing")
    m(xb); torch.cuda.synchronize()
    _lib.load().uitk_set_timing_buffers(h, None, None)
    raw = buf.cpu().numpy().astype(np.float64)
    med = np.median(raw[:, :8] / 12.0, 0)
    fixed = np.median(raw[:, 8:], 0)
    print(f"waves/WG {waves} B {B}: per-block cycles (median over {nw} waves), block total {med.sum():.0f}; "
          f"12 blocks {12 * med.sum():.0f}; fixed {fixed.sum():.0f}")
    print("   " + "  ".join(f"{n}={v:.0f}" for n, v in zip(names, med)))
    print("   " + "  ".join(f"{n}={v:.0f}" for n, v in zip(names2, fixed)))
    half = (nw // 2)
    for tag, sl in (("first-round WGs", slice(0, half)), ("second-round WGs", slice(half, nw))):
        fx = np.median(raw[sl, 8:], 0)
        print(f"   {tag}: " + "  ".join(f"{n}={v:.0f}" for n, v in zip(names2, fx)))
    life = raw.sum(1).reshape(-1, 8).max(1)              # encoder workgroup lifetime = its slowest wave (stamped phases only)
    half_wg = len(life) // 2
    print(f"   encoder workgroup lifetimes (cycles): median {np.median(life):.0f}  p10 {np.percentile(life, 10):.0f}  p90 {np.percentile(life, 90):.0f}  "
          f"max {life.max():.0f}  -> slowest / median = {life.max() / np.median(life):.3f};  first-round median {np.median(life[:half_wg]):.0f}, "
          f"second-round {np.median(life[half_wg:]):.0f};  by blockIdx % 8: " + " ".join(f"{np.median(life[i::8]):.0f}" for i in range(8)))
    fr = fbuf.cpu().numpy().astype(np.float64)
    fr = fr[fr.sum(1) > 0]
    npass = B * 7 / 512.0
    fn = ["pass setup+load issue", "load wait+window+FFT1+tw+Ewrite", "Eread+FFT2+Zwrite", "Zread+split+P", "barrier", "mel MFMA", "tail", "log+store"]
    print(f"   frontend per-pass cycles (median over {len(fr)} waves, {npass:.1f} passes/WG): "
          + "  ".join(f"{n}={v:.0f}" for n, v in zip(fn, np.median(fr, 0) / npass)) + f"  total/pass {np.median(fr.sum(1)) / npass:.0f}")
    tot = fr.sum(1)                                      # per-wave lifetime in the pass loops (static pass assignment: the kernel
    pw = tot.reshape(-1, 4).max(1)                       # ends with its slowest workgroup)
    print(f"   frontend workgroup lifetimes (cycles): median {np.median(pw):.0f}  p90 {np.percentile(pw, 90):.0f}  max {pw.max():.0f}  "
          f"min {pw.min():.0f}  -> slowest / median = {pw.max() / np.median(pw):.3f}")
    wv = tot.reshape(-1, 4)
    print("   frontend lifetime by wave index (median): " + "  ".join(f"w{i}={np.median(wv[:, i]):.0f}" for i in range(4))
          + "   by blockIdx % 8 (XCD, median of workgroup max): " + "  ".join(f"{np.median(pw[i::8]):.0f}" for i in range(8)))
    order = np.argsort(pw)
    print("   slowest workgroups (blockIdx): " + " ".join(str(int(i)) for i in order[-12:]) + "   fastest: " + " ".join(str(int(i)) for i in order[:12]))

