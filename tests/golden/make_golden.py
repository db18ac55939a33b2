"""Generate the golden fixtures in tests/golden/*.npz by running the REFERENCE's own models/uit.py on torch CPU.

Run only in the build container (needs /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference module is imported unmodified from where it lies (see _ref_shims.py for the three stand-in
packages); the weights are the deterministic synthetic checkpoints of uit_mobile_amd/synth.py loaded with
`load_state_dict(strict=True)` (this pins the state_dict key set and shapes too).  Everything stored is data:
inputs (int16 PCM; float input = pcm/32768, exact) and the reference's outputs at the stage boundaries of
SURVEY.md section 8a.  Stage tensors are captured with forward hooks on the reference's own sub-modules.
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path[:0] = [str(HERE), str(HERE.parents[1])]           # _ref_shims (here) and the package (uit_mobile_amd.synth: the seeded checkpoints)
# imported BEFORE the stand-in torchaudio module is installed (transformers probes for the real package)
from transformers.audio_utils import mel_filter_bank, spectrogram  # noqa: E402
import _ref_shims  # noqa: E402
from uit_mobile_amd.synth import make_kwargs_variant_state_dict, make_state_dict, make_trained_like_state_dict, synth_pcm_i16  # noqa: E402

REF = Path("/root/reference")
uit = _ref_shims.import_reference_uit(str(REF))
torch.set_num_threads(8)
torch.manual_seed(0)


def build_ref(name, seed, scale, outputdim=537, eval_avg="mean"):
    m = getattr(uit, name)(outputdim=outputdim, target_length=102, eval_avg=eval_avg)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict(name, seed, scale, outputdim).items()}
    missing = set(m.state_dict().keys()) ^ set(sd.keys())
    assert not missing, missing
    m.load_state_dict(sd, strict=True)
    return m.eval()


def build_ref_trained(name, seed, outputdim=537, eval_avg="mean"):
    """The reference's own UITBase carrying the trained-like checkpoint of synth_weights.make_trained_like_state_dict."""
    m = getattr(uit, name)(outputdim=outputdim, target_length=102, eval_avg=eval_avg)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_trained_like_state_dict(name, seed, outputdim).items()}
    assert not (set(m.state_dict().keys()) ^ set(sd.keys()))
    m.load_state_dict(sd, strict=True)
    return m.eval()


def run_ref(m, pcm_i16, stages_for=0):
    """-> dict of reference outputs for wave = pcm/32768 (dataset.py:44-46 convention)."""
    wave = torch.from_numpy(pcm_i16.astype(np.float32) / 32768.0)
    cap = {}
    hooks = [m.front_end.register_forward_hook(lambda mod, i, o: cap.__setitem__("mel_db", o.detach().clone()))]
    toks = []
    if stages_for:
        # tokens entering blocks[0] == patch-embed + pos (+flatten); then the output of every block
        hooks.append(m.blocks[0].register_forward_pre_hook(lambda mod, i: toks.append(i[0].detach().clone())))
        for blk in m.blocks:
            hooks.append(blk.register_forward_hook(lambda mod, i, o: toks.append(o.detach().clone())))
    pre = []
    hooks.append(m.outputlayer.register_forward_hook(lambda mod, i, o: pre.append(o.detach().clone())))
    with torch.no_grad():
        probs = m(wave)
    for h in hooks:
        h.remove()
    out = dict(probs=probs.numpy(), mel_db=cap["mel_db"].numpy(),
               logits=torch.stack(pre, 1).numpy())           # [B, ncrops, C]
    if stages_for:
        depth = len(m.blocks)
        ncrops = len(pre)
        st = torch.stack(toks, 0).reshape(ncrops, depth + 1, *toks[0].shape)   # [ncrops, depth+1, B, N, D]
        out["stages"] = st.permute(2, 0, 1, 3, 4)[:stages_for].contiguous().numpy()
    return out


def read_wav_i16(path):
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    assert sr == 16000 and data.dtype == np.int16 and data.ndim == 1
    return data


def crosscheck_frontend():
    """Independent check of the torchaudio restatement against transformers.audio_utils (numpy)."""
    pcm = read_wav_i16(REF / "samples/water_000.wav").astype(np.float32) / 32768.0
    fe = torch.nn.Sequential(_ref_shims.MelSpectrogram(f_min=0, sample_rate=16000, win_length=512, center=True,
                                                       n_fft=512, f_max=8000, hop_length=160, n_mels=64),
                             _ref_shims.AmplitudeToDB(top_db=120))
    ours = fe(torch.from_numpy(pcm)[None])[0].numpy()
    fb = mel_filter_bank(257, 64, 0.0, 8000.0, 16000, norm=None, mel_scale="htk")
    win = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(512) / 512))
    theirs = spectrogram(pcm.astype(np.float64), win, 512, 160, fft_length=512, power=2.0, center=True,
                         pad_mode="reflect", mel_filters=fb, log_mel="dB", reference=1.0, min_value=1e-10,
                         db_range=120.0)
    d_fb = np.abs(fe[0].mel_scale.fb.numpy() - fb).max()
    d_db = np.abs(ours - theirs).max()
    print(f"[crosscheck] transformers.audio_utils: fb max diff {d_fb:.2e}, dB max diff {d_db:.2e} {ours.shape}")
    assert d_fb < 1e-4 and d_db < 5e-3
    sd = make_state_dict("uit_xxxs", 0)
    assert np.abs(sd["front_end.0.mel_scale.fb"] - fe[0].mel_scale.fb.numpy()).max() < 2e-5
    assert np.abs(sd["front_end.0.spectrogram.window"] - fe[0].spectrogram.window.numpy()).max() < 1e-6
    return float(d_fb), float(d_db)


def main_round2():
    """Round-2 fixtures: trained-checkpoint magnitudes (|logit| >= 10), the 10-crop 160 000-sample clip, a short clip on
    trained-like weights."""
    pcm = synth_pcm_i16(8, 16000, seed=0)
    pcm[5] = 0
    for name, seed in (("uit_xs", 1), ("uit_xxxs", 2)):
        m = build_ref_trained(name, seed)
        r = run_ref(m, pcm, stages_for=2)
        assert np.abs(r["logits"]).max() >= 10.0, np.abs(r["logits"]).max()
        np.savez_compressed(HERE / f"trained_{name}.npz", pcm=pcm, seed=seed, probs=r["probs"], logits=r["logits"],
                            stages=r["stages"])
        print("trained-like", name, "logit range", r["logits"].min(), r["logits"].max(), "probs max", r["probs"].max())
    # short clips (51 frames -> 3 time patches: masked padding tokens in the kernel) on trained-like weights: sharp attention
    pcm_short = synth_pcm_i16(3, 8000, seed=4)
    r = run_ref(build_ref_trained("uit_xxxs", 2), pcm_short)
    np.savez_compressed(HERE / "trained_short_uit_xxxs.npz", pcm=pcm_short, seed=2, probs=r["probs"], logits=r["logits"])
    # the reference's own sample clips (real audio: water, a GSC keyword) through the trained-like XS checkpoint
    m = build_ref_trained("uit_xs", 1)
    out = {}
    for wav in ("water_000.wav", "water_003.wav", "85b877b5_nohash_0.wav"):
        r = run_ref(m, read_wav_i16(REF / "samples" / wav)[None])
        key = wav.split(".")[0]
        out[f"{key}_probs"], out[f"{key}_logits"] = r["probs"][0], r["logits"][0]
        print("trained-like uit_xs on", wav, "logit range", r["logits"].min(), r["logits"].max())
    np.savez_compressed(HERE / "trained_samples_uit_xs.npz", seed=1, **out)      # PCM: samples_uit_xs.npz
    # AudioSet-style 10 s clip: 160 000 samples -> 1001 frames -> 10 crops, the last one re-cropped (models/uit.py:468-488)
    pcm10 = synth_pcm_i16(2, 160000, seed=6)[1:2]            # clip index 1 of the generator = a loud tone; keep one noise clip too
    pcm10 = np.concatenate([pcm10, synth_pcm_i16(1, 160000, seed=7)])
    for avg in ("mean", "max"):
        m = build_ref("uit_xxxs", 11, 1.0, eval_avg=avg)
        r = run_ref(m, pcm10)
        assert r["logits"].shape[1] == 10
        if avg == "mean":
            np.savez_compressed(HERE / "long10_uit_xxxs.npz", pcm=pcm10, seed=11, probs_mean=r["probs"], logits=r["logits"])
            out10 = dict(np.load(HERE / "long10_uit_xxxs.npz"))
        else:
            out10["probs_max"] = r["probs"]
            np.savez_compressed(HERE / "long10_uit_xxxs.npz", **out10)
    print("long10", out10["logits"].shape)


def main_variants():
    """the other factories of models/uit.py (:514-579) -- full-width Attention (2 heads x 64), GELU or ReLU MLP,
    mean pooling -- at init scale and (ReLU, GELU d4) at trained-like magnitudes; per-block tokens for two clips."""
    pcm = synth_pcm_i16(8, 16000, seed=0)
    pcm[5] = 0
    for name, seed, trained in (("audio_transformer_h128_d4_m3_relu", 21, False), ("audio_transformer_h128_d4_m3", 22, False),
                                ("audio_transformer_h128_d6_m3_relu", 23, True), ("audio_transformer_h128_d6_m3", 24, True)):
        m = getattr(uit, name)(outputdim=537, target_length=102)
        mk = make_trained_like_state_dict if trained else make_state_dict
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in mk(name, seed, attention="full").items()}
        assert not (set(m.state_dict().keys()) ^ set(sd.keys()))
        m.load_state_dict(sd, strict=True)
        r = run_ref(m.eval(), pcm, stages_for=2)
        np.savez_compressed(HERE / f"variant_{name}.npz", pcm=pcm, seed=seed, trained=trained, probs=r["probs"], logits=r["logits"],
                            stages=r["stages"])
        print("variant", name, "trained-like" if trained else "init", "logit range", r["logits"].min(), r["logits"].max())


def main_dm():
    """pooling='dm' (models/uit.py:405-412) on the uit_* family: init-scale 3XS incl. a short clip batch, trained-like XS."""
    pcm = synth_pcm_i16(8, 16000, seed=0)
    pcm[5] = 0
    out = {}
    for tag, name, seed, trained in (("xxxs", "uit_xxxs", 31, False), ("xs_trained", "uit_xs", 3, True)):
        m = getattr(uit, name)(outputdim=537, target_length=102, pooling="dm")
        mk = make_trained_like_state_dict if trained else make_state_dict
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in mk(name, seed).items()}, strict=True)
        m.eval()
        with torch.no_grad():
            out[f"{tag}_probs"] = m(torch.from_numpy(pcm.astype(np.float32) / 32768.0)).numpy()
            if not trained:
                short = synth_pcm_i16(3, 8000, seed=4)
                out["xxxs_short_pcm"], out["xxxs_short_probs"] = short, m(torch.from_numpy(short.astype(np.float32) / 32768.0)).numpy()
                two = synth_pcm_i16(2, 16384, seed=5)                  # two crops
                out["xxxs_crops_pcm"], out["xxxs_crops_probs"] = two, m(torch.from_numpy(two.astype(np.float32) / 32768.0)).numpy()
        out[f"{tag}_seed"] = seed
        print("dm", name, "probs range", out[f"{tag}_probs"].min(), out[f"{tag}_probs"].max())
    np.savez_compressed(HERE / "pooling_dm.npz", pcm=pcm, **out)


def main_token():
    """pooling='token' (models/uit.py:389-392, 399-401; the UITBase default): a cls token in front of the 24 patch tokens."""
    pcm = synth_pcm_i16(8, 16000, seed=0)
    pcm[5] = 0
    out = {}
    for tag, name, seed, trained, att in (("xxxs", "uit_xxxs", 51, False, "bneck"), ("xs_trained", "uit_xs", 4, True, "bneck"),
                                          ("h128", "audio_transformer_h128_d4_m3", 52, False, "full")):
        m = getattr(uit, name)(outputdim=537, target_length=102, pooling="token")
        mk = make_trained_like_state_dict if trained else make_state_dict
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in mk(name, seed, attention=att).items()}
        sd["cls_token"] = sd["cls_token"] * (0.0 if False else 1.0) + 0.3 * torch.randn(1, 1, 128, generator=torch.Generator().manual_seed(seed))
        m.load_state_dict(sd, strict=True)
        r = run_ref(m.eval(), pcm)
        out[f"{tag}_probs"], out[f"{tag}_logits"], out[f"{tag}_seed"] = r["probs"], r["logits"], seed
        out[f"{tag}_cls_token"] = sd["cls_token"].numpy()
        if tag == "xxxs":
            short = synth_pcm_i16(3, 8000, seed=4)
            r = run_ref(m, short)
            out["xxxs_short_pcm"], out["xxxs_short_probs"], out["xxxs_short_logits"] = short, r["probs"], r["logits"]
        print("token", name, "logit range", out[f"{tag}_logits"].min(), out[f"{tag}_logits"].max())
    np.savez_compressed(HERE / "pooling_token.npz", pcm=pcm, **out)


def main_kwargs():
    """Round-3 fixtures: the UITBase kwargs the wrapper used to refuse (models/uit.py:253-293): qkv_bias=False + LayerScale
    (init_values) + init_bn=False (the reference's GlobalNormer) on one model; center=False (no reflect padding, 1 + (T - n_fft) / hop
    frames) on 1 s clips (97 frames, one crop) and on 17 000-sample clips (104 frames, two crops)."""
    pcm = synth_pcm_i16(6, 16000, seed=21)
    pcm[4] = 0
    m = uit.uit_xxxs(outputdim=537, target_length=102, qkv_bias=False, init_values=0.1, init_bn=False)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_kwargs_variant_state_dict("uit_xxxs", 33).items()}
    assert not (set(m.state_dict().keys()) ^ set(sd.keys())), set(m.state_dict().keys()) ^ set(sd.keys())
    m.load_state_dict(sd, strict=True)
    r = run_ref(m.eval(), pcm, stages_for=2)
    np.savez_compressed(HERE / "kwargs_uit_xxxs.npz", pcm=pcm, seed=33, probs=r["probs"], logits=r["logits"], stages=r["stages"])
    print("kwargs variant logit range", r["logits"].min(), r["logits"].max())
    out = {}
    for tag, T in (("1s", 16000), ("2crop", 17000)):
        p = synth_pcm_i16(4, T, seed=22)
        p[3] = 0
        m = uit.uit_xxxs(outputdim=537, target_length=102, center=False)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xxxs", 34, 2.0).items()}, strict=True)
        r = run_ref(m.eval(), p)
        out[f"{tag}_pcm"], out[f"{tag}_probs"], out[f"{tag}_logits"], out[f"{tag}_mel_db"] = p, r["probs"], r["logits"], r["mel_db"]
        print("center=False", tag, "frames", r["mel_db"].shape[-1], "crops", r["logits"].shape[1])
    np.savez_compressed(HERE / "nocenter_uit_xxxs.npz", seed=34, **out)
    # round 4: win_size < n_fft (models/uit.py:290, 301-303: MelSpectrogram(win_length=400, n_fft=512) -> torch.stft pads the window)
    from uit_mobile_amd.synth import hann_periodic
    out = {}
    for tag, T, center in (("1s", 16000, True), ("2crop", 16384, True), ("nocenter", 16000, False)):
        p = synth_pcm_i16(4, T, seed=23)
        p[2] = 0
        m = uit.uit_xxxs(outputdim=537, target_length=102, win_size=400, center=center)
        sd = dict(make_state_dict("uit_xxxs", 35, 2.0))
        sd["front_end.0.spectrogram.window"] = hann_periodic(400)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        r = run_ref(m.eval(), p)
        out[f"{tag}_pcm"], out[f"{tag}_probs"], out[f"{tag}_logits"], out[f"{tag}_mel_db"] = p, r["probs"], r["logits"], r["mel_db"]
        print("win_size=400", tag, "frames", r["mel_db"].shape[-1], "crops", r["logits"].shape[1])
    np.savez_compressed(HERE / "win400_uit_xxxs.npz", seed=35, **out)


def main_heldout():
    """Round-5 fixture: the WIDE held-out set under the imported reference.  All eleven recordings of the
    reference's samples/ (native length: one or two crops, each its own forward call as inference.py does) and one member of every
    adversarial family of uit_mobile_amd.synth (impulses, clicks, DC, clipping, LSB dither, Nyquist alternation, quiet speech-like
    stacks ... as ONE batch: batch-global top_db floor), through three trained-like checkpoints: uit_xs seed 207 (the sweep's worst
    checkpoint for nearest-rounded fp16 weights), uit_xxxs seed 203 and the full-attention GELU factory d4 seed 206 (the checkpoint
    that sat at the gate in round 4)."""
    from uit_mobile_amd.synth import adversarial_pcm_i16
    wavs = sorted(p.name for p in (REF / "samples").glob("*.wav"))
    assert len(wavs) == 11, wavs
    adv, names = adversarial_pcm_i16(0, 1)
    out = {"adv_pcm": adv, "adv_names": np.array(len(names)), "wav_count": np.array(len(wavs))}
    for i, wv in enumerate(wavs):
        out[f"wav{i:02d}_pcm"] = read_wav_i16(REF / "samples" / wv)
    for tag, name, seed, att in (("xs207", "uit_xs", 207, "bneck"), ("xxxs203", "uit_xxxs", 203, "bneck"),
                                 ("d4gelu206", "audio_transformer_h128_d4_m3", 206, "full")):
        m = getattr(uit, name)(outputdim=537, target_length=102)
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_trained_like_state_dict(name, seed, attention=att).items()}
        assert not (set(m.state_dict().keys()) ^ set(sd.keys()))
        m.load_state_dict(sd, strict=True)
        m.eval()
        for i in range(len(wavs)):
            r = run_ref(m, out[f"wav{i:02d}_pcm"][None])
            out[f"{tag}_wav{i:02d}_probs"], out[f"{tag}_wav{i:02d}_logits"] = r["probs"][0], r["logits"][0]
        r = run_ref(m, adv)
        out[f"{tag}_adv_probs"], out[f"{tag}_adv_logits"] = r["probs"], r["logits"]
        print("heldout", tag, "wav logit range", min(out[f"{tag}_wav{i:02d}_logits"].min() for i in range(11)),
              max(out[f"{tag}_wav{i:02d}_logits"].max() for i in range(11)), "adv", r["logits"].min(), r["logits"].max())
    np.savez_compressed(HERE / "heldout_wide.npz", **out)
    print("wavs:", wavs, "adversarial:", names)


def main_general():
    """Round-6 fixtures: UITBase geometries OUTSIDE the published factories' (general_cases.py) through the reference's own UITBase --
    what pins the oracle, and through it the library's general engine (csrc/general.hip), for n_mels / n_fft / win / hop / center /
    patch_size / target_length / embed_dim / num_heads / mlp_ratio / f_min / f_max away from their defaults."""
    from general_cases import GENERAL_CASES, state_dict_args
    from uit_mobile_amd.synth import make_general_state_dict
    for tag, case in GENERAL_CASES.items():
        kw = dict(case["kwargs"])
        kw["attention_type"] = {"bneck": "BNeckAttention", "full": "Attention"}[case["attention"]]
        kw["act_layer"] = {"relu": torch.nn.ReLU, "gelu": torch.nn.GELU}[case["act"]]
        m = uit.UITBase(**kw)
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_general_state_dict(**state_dict_args(case)).items()}
        diff = set(m.state_dict().keys()) ^ set(sd.keys())
        assert not diff, (tag, diff)
        m.load_state_dict(sd, strict=True)
        m.eval()
        out = {}
        for i, (B, T) in enumerate(case["clips"]):
            pcm = synth_pcm_i16(B, T, seed=300 + 10 * case["seed"] + i)
            if B > 2:
                pcm[B - 1] = 0                                   # a silent clip: the batch-global floor
            r = run_ref(m, pcm)
            out[f"c{i}_pcm"], out[f"c{i}_probs"] = pcm, r["probs"]
            if tag != "default_uitbase":                         # (its front-end is the default one, pinned elsewhere: keep the fixture small)
                out[f"c{i}_mel_db"] = r["mel_db"]
            if kw["pooling"] != "dm":
                out[f"c{i}_logits"] = r["logits"]
            print("general", tag, "clips", (B, T), "frames", r["mel_db"].shape[-1], "crops", r["logits"].shape[1],
                  "logit range", float(r["logits"].min()), float(r["logits"].max()), "prob range", float(r["probs"].min()), float(r["probs"].max()))
        np.savez_compressed(HERE / f"general_{tag}.npz", **out)


def main():
    if "--general" in sys.argv:
        return main_general()
    if "--heldout" in sys.argv:
        return main_heldout()
    if "--kwargs" in sys.argv:
        return main_kwargs()
    if "--token" in sys.argv:
        return main_token()
    if "--dm" in sys.argv:
        return main_dm()
    if "--round2" in sys.argv:
        return main_round2()
    if "--variants" in sys.argv:
        return main_variants()
    d_fb, d_db = crosscheck_frontend()
    meta = dict(crosscheck_fb=d_fb, crosscheck_db=d_db)

    # ---- case 1: synthetic 1 s batch (silent + loud + quiet clips -> batch-global top_db floor)
    pcm = synth_pcm_i16(8, 16000, seed=0)
    pcm[5] = 0                                               # a fully silent clip
    for name, seed in (("uit_xxxs", 11), ("uit_xxs", 12), ("uit_xs", 13)):
        m = build_ref(name, seed, 1.0)
        r = run_ref(m, pcm, stages_for=2)
        np.savez_compressed(HERE / f"synth1s_{name}.npz", pcm=pcm, seed=seed, scale=1.0, **r)
        print(name, "probs", r["probs"].shape, "logit range", r["logits"].min(), r["logits"].max(),
              "mel", r["mel_db"].min(), r["mel_db"].max())
    # larger-magnitude weights (x3) for the precision modes
    m = build_ref("uit_xs", 14, 3.0)
    r = run_ref(m, pcm[:4])
    np.savez_compressed(HERE / "synth1s_uit_xs_scale3.npz", pcm=pcm[:4], seed=14, scale=3.0,
                        probs=r["probs"], logits=r["logits"])

    # ---- case 2: the reference's sample clips (config #1: water_000 = 16384 samples -> 2 crops)
    for name, seed in (("uit_xs", 13), ("uit_xxxs", 11)):
        m = build_ref(name, seed, 1.0)
        out = {}
        for wav in ("water_000.wav", "water_003.wav", "85b877b5_nohash_0.wav"):
            p = read_wav_i16(REF / "samples" / wav)
            r = run_ref(m, p[None])
            key = wav.split(".")[0]
            out[f"{key}_pcm"] = p
            out[f"{key}_probs"] = r["probs"][0]
            out[f"{key}_logits"] = r["logits"][0]
            print(name, wav, len(p), "crops", r["logits"].shape[1], "top3", np.argsort(-r["probs"][0])[:3])
        # evaluate.py:56-64 pad rule on the 16000-sample GSC clip: diff = int((101 - T/160)*160) = 160
        p = read_wav_i16(REF / "samples/85b877b5_nohash_0.wav")
        diff = int((102 - 1 - len(p) / 160) * 160)
        pp = np.concatenate([p, np.zeros(diff, np.int16)])
        r = run_ref(m, pp[None])
        out["gsc_padded_pcm"], out["gsc_padded_probs"], out["gsc_pad_diff"] = pp, r["probs"][0], diff
        np.savez_compressed(HERE / f"samples_{name}.npz", seed=seed, **out)

    # ---- case 3: long clip -> multi-crop with tail re-crop (313 frames -> 102,102,102,tail), mean and max
    pcm_long = synth_pcm_i16(2, 50000, seed=3)
    for avg in ("mean", "max"):
        m = build_ref("uit_xxxs", 11, 1.0, eval_avg=avg)
        r = run_ref(m, pcm_long)
        np.savez_compressed(HERE / f"long_{avg}_uit_xxxs.npz", pcm=pcm_long, seed=11, probs=r["probs"],
                            logits=r["logits"], mel_db=r["mel_db"])
        print("long", avg, r["logits"].shape)

    # ---- case 4: short clip (8000 samples -> 51 frames -> 3 time patches, 12 tokens), and 527 classes
    pcm_short = synth_pcm_i16(3, 8000, seed=4)
    m = build_ref("uit_xxxs", 11, 1.0)
    r = run_ref(m, pcm_short, stages_for=1)
    np.savez_compressed(HERE / "short_uit_xxxs.npz", pcm=pcm_short, seed=11, **r)
    m = build_ref("uit_xxs", 15, 1.0, outputdim=527)
    r = run_ref(m, pcm[:3])
    np.savez_compressed(HERE / "synth1s_uit_xxs_527.npz", pcm=pcm[:3], seed=15, probs=r["probs"], logits=r["logits"])

    # ---- pos-embed resize (models/uit.py:416-450): checkpoint with a longer time_pos_embed -> truncated
    sd = make_state_dict("uit_xxxs", 11, 1.0, target_length=160)
    m = getattr(uit, "uit_xxxs")(outputdim=537, target_length=102)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    r = run_ref(m.eval(), pcm[:2])
    np.savez_compressed(HERE / "posresize_uit_xxxs.npz", pcm=pcm[:2], seed=11, probs=r["probs"])
    np.savez(HERE / "meta.npz", **meta)
    main_round2()
    main_variants()
    main_dm()
    main_token()
    main_kwargs()
    main_heldout()
    print("done")


if __name__ == "__main__":
    main()
