"""GPU box: precision='fp8' (config #5) -- (a) does the kernel compute the intended quantised network (GPU vs tests/fp8_emu.py),
(b) what the quantisation costs against the exact reference arithmetic (oracle), (c) kernel time vs f16.   python tools/fp8_report.py"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tests" / "golden")]
import fp8_emu
from oracle.oracle import Oracle
from uit_mobile_amd.synth import make_state_dict, make_trained_like_state_dict, synth_pcm_i16
from uit_mobile_amd import models
from bench import synth_batch

pcm = synth_pcm_i16(8, 16000, seed=0); pcm[5] = 0
for name, kind in (("uit_xxxs", "init"), ("uit_xs", "init"), ("uit_xxxs", "trained"), ("uit_xs", "trained")):
    sd = make_trained_like_state_dict(name, 1) if kind == "trained" else make_state_dict(name, 13)
    depth = {"uit_xs": 12, "uit_xxs": 6, "uit_xxxs": 4}[name]
    o = Oracle(sd)
    r = o.forward(pcm)
    emu = fp8_emu.forward(sd, fp8_emu.mel_bn_from_oracle(sd, r["mel_db"]), depth).numpy()
    out = {}
    for prec in ("f16", "fp8"):
        m = getattr(models, name)(outputdim=537, target_length=102, precision=prec)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.cuda().eval()
        p, l = m(torch.from_numpy(pcm).cuda(), return_logits=True)
        out[prec] = (p.cpu().numpy(), l.cpu().numpy()[:, 0])
    ref_l, ref_p = r["logits"][:, 0], r["probs"]
    print(f"{name} {kind}: |logit|max {np.abs(ref_l).max():.1f} | f16 vs oracle dlogit {np.abs(out['f16'][1]-ref_l).max():.1e} | "
          f"fp8 GPU vs emu dlogit {np.abs(out['fp8'][1]-emu).max():.2e} | fp8 GPU vs oracle dlogit {np.abs(out['fp8'][1]-ref_l).max():.2e} "
          f"dprob {np.abs(out['fp8'][0]-ref_p).max():.2e} | emu vs oracle dlogit {np.abs(emu-ref_l).max():.2e} | finite {bool(np.isfinite(out['fp8'][1]).all())}", flush=True)

x = synth_batch(8192, 16000, 0, "cuda", False)
sd = make_state_dict("uit_xs", 13)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for prec in ("f16", "fp8", "f16", "fp8"):
    m = models.uit_xs(outputdim=537, target_length=102, precision=prec)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.cuda().eval()
    for _ in range(30): m(x)
    fe, en = [], []
    for _ in range(40):
        ev[0].record(); st = m.frontend(x); ev[1].record(); p = m.encode(st); ev[2].record()
        torch.cuda.synchronize(); fe.append(ev[0].elapsed_time(ev[1])); en.append(ev[1].elapsed_time(ev[2]))
    print(f"{prec}: front-end {np.median(fe):.4f} ms encoder {np.median(en):.4f} ms", flush=True)
