"""GPU: what is left between precision 'f32' (the library's plain fp32 forward) and the C oracle?  Splits it into the ENCODER error of the
rung -- its logits against an fp64 evaluation (tools/err_attrib_centring.py) of the SAME mel spectrogram it read (uitk_mel_db) -- and the
FRONT-END difference -- that fp64 evaluation from the product's mel against the same from the oracle's mel: two fp32 FFTs differ on their
round-off floor (a few hundredths of a dB in bins far below a frame's peak), and a checkpoint amplifies that by its own conditioning.

    python tools/fe_split.py        -> profiles/round5/fe_split.log

Measured (uit_xxs seed 216, the 67-clip wide set): without an attention sink the rung is 7.5e-7 from fp64 (the oracle: 1.2e-5) and the
front-end difference is 2.3e-4 (a bin-edge tone pair); with the sink of synth.make_sink_trained_like_state_dict (gain 4) the rung is 2.0e-6
from fp64 and the front-end difference is 6.2e-4 -- every bit of the rung's deviation from the oracle.  On such a checkpoint a 1e-3 gate
between two fp32 implementations measures their FFTs, not their encoders.  Second part: both front-ends against an fp64 log-mel of the
same clips (numpy): rms 0.0022 dB each, worst 0.12 dB (product) / 0.10 dB (oracle) -- neither is the noisy one."""
import sys, warnings
sys.path[:0] = ["/root/repo", "/root/repo/tools", "/root/repo/tests", "/root/repo/tests/golden"]
import numpy as np, torch
warnings.simplefilter("ignore")
import err_attrib_centring as E
from oracle.oracle import Oracle
from uit_mobile_amd import models
from uit_mobile_amd.synth import DEPTH, hann_periodic, make_sink_trained_like_state_dict as mk, make_trained_like_state_dict as mk0, melscale_fbanks
import test_gpu_parity as T
golden = lambda n: dict(np.load(f"/root/repo/tests/golden/{n}.npz"))
x, names = T.held_out_inputs(golden, with_names=True)
xd = torch.from_numpy(x).cuda()
def bn(sd, mel):
    s = np.asarray(sd["init_bn.1.weight"], np.float64) / np.sqrt(np.asarray(sd["init_bn.1.running_var"], np.float64) + 1e-5)
    return torch.from_numpy((mel.astype(np.float64) - np.asarray(sd["init_bn.1.running_mean"], np.float64)[None, :, None]) * s[None, :, None] + np.asarray(sd["init_bn.1.bias"], np.float64)[None, :, None])
for tag, sd in (("no sink", mk0("uit_xxs", 216)), ("sink4", mk("uit_xxs", 216, gain=4.0))):
    o = Oracle(sd)
    r = o.forward(x)
    m = models.uit_xxs(outputdim=537, target_length=102, precision="f32")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.cuda().eval()
    with torch.no_grad():
        lg = m(xd, return_logits=True)[1].cpu().numpy().reshape(len(x), -1)
        mel_gpu = m.melspec_db(xd).cpu().numpy()
    mel_or = r["mel_db"]
    t_gpu = E.forward(sd, bn(sd, mel_gpu), DEPTH["uit_xxs"], "relu", "bneck", "mean", ()).numpy()
    t_or = E.forward(sd, bn(sd, mel_or), DEPTH["uit_xxs"], "relu", "bneck", "mean", ()).numpy()
    lo = r["logits"].reshape(len(x), -1)
    d_enc = np.abs(lg - t_gpu).max(1); d_fe = np.abs(t_gpu - t_or).max(1); d_tot = np.abs(lg - lo).max(1); d_or = np.abs(lo - t_or).max(1)
    dm = np.abs(mel_gpu[:, :, :96] - mel_or[:, :, :96]).reshape(len(x), -1).max(1)
    i = int(d_tot.argmax())
    print(f"{tag}: f32 rung vs fp64-from-ITS-OWN-mel (encoder error of the rung): worst {d_enc.max():.2e}; oracle vs fp64-from-its-mel {d_or.max():.2e}; "
          f"fp64 from the product's mel vs fp64 from the oracle's mel (FRONT-END difference, amplified by the checkpoint): worst {d_fe.max():.2e} ({names[int(d_fe.argmax())]}); "
          f"f32 rung vs oracle: worst {d_tot.max():.2e} ({names[i]}; its mel differs by up to {dm[i]:.3f} dB from the oracle's in frames 0..95)", flush=True)

# ---- both front-ends against an fp64 log-mel (the last checkpoint's mel: the front-end does not depend on the encoder weights)
mel_or = r["mel_db"]
# exact (fp64) log-mel: reflect pad 256, periodic Hann, 512-point rFFT, power, HTK filterbank, 10 log10(max(., 1e-10)), batch-global floor
xf = x.astype(np.float64) / 32768.0
pad = np.pad(xf, ((0, 0), (256, 256)), mode="reflect")
nF = 1 + x.shape[1] // 160
idx = 160 * np.arange(nF)[:, None] + np.arange(512)[None, :]
fr = pad[:, idx] * hann_periodic(512).astype(np.float64)
P = np.abs(np.fft.rfft(fr, axis=-1)) ** 2
fb = melscale_fbanks(257, 0.0, 8000.0, 64, 16000).astype(np.float64)
db = 10.0 * np.log10(np.maximum(P @ fb, 1e-10)).transpose(0, 2, 1)
db = np.maximum(db, db.max() - 120.0)
for tag, mel in (("product (GPU fp32 FFT)", mel_gpu), ("oracle (C fp32 FFT)", mel_or)):
    d = np.abs(mel.astype(np.float64) - db)[:, :, :96].reshape(len(x), -1)
    w = d.max(1)
    order = np.argsort(-w)[:4]
    print(f"{tag} vs fp64 log-mel: worst {w.max():.4f} dB, median over clips {np.median(w):.5f} dB, rms {np.sqrt((d ** 2).mean()):.6f} dB; worst clips: " + ", ".join(f"{names[i]} {w[i]:.4f}" for i in order), flush=True)
dd = np.abs(mel_gpu.astype(np.float64) - mel_or)[:, :, :96].reshape(len(x), -1).max(1)
print("product vs oracle: worst %.4f dB (%s)" % (dd.max(), names[int(dd.argmax())]))
