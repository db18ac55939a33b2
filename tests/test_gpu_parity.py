"""GPU parity tests proper (-m gpu): the HIP path, called through the C ABI (libuitk.so via the Python wrapper),
against (a) the committed golden fixtures produced by the reference's own models/uit.py on torch CPU and (b) the
CPU oracle on seeded inputs.

Tolerance (BASELINE.json north_star): outputs within 1e-3 abs of the reference PyTorch CPU path -- asserted on both
the probabilities the reference returns and the pre-sigmoid logits, for the default precision of the factories = the product and
bench mode (round 5: 'auto' = 'f16c' -- fp16 MFMA operands, fp32 accumulate, single fp16 weights with calibrated rounding; the classifier
head runs on split hi+lo operands -- held to the library's absolute budget on its held-out clips, else rebuilt on split weights): `build()` / `build_trained()` take the factory's default unless a test names another precision as its
subject.  The bf16-operand and fp8 modes are measured, non-parity modes: their tests assert measured bounds and say so.
"""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import assert_mel_close, centred
from oracle.oracle import Oracle
from uit_mobile_amd.synth import make_state_dict, make_trained_like_state_dict, synth_pcm_i16
from uit_mobile_amd import models

pytestmark = pytest.mark.gpu
TOL = 1e-3
ROOT = Path(__file__).resolve().parents[1]


def build(name, seed, scale=1.0, prec=None, outputdim=537, **kw):      # prec=None: the factory's own default = the product / bench mode
    sd = make_state_dict(name, int(seed), float(scale), outputdim=outputdim,
                         **({"target_length": kw.pop("ckpt_target_length")} if "ckpt_target_length" in kw else {}))
    m = getattr(models, name)(outputdim=outputdim, target_length=102, precision=prec, **kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.to("cuda").eval(), sd


def build_trained(name, seed, prec=None, **kw):
    """Trained-checkpoint magnitudes (synth_weights.make_trained_like_state_dict): reference logits beyond -10."""
    sd = make_trained_like_state_dict(name, int(seed))
    m = getattr(models, name)(outputdim=537, target_length=102, precision=prec, **kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.to("cuda").eval(), sd


def run(m, pcm):
    with torch.no_grad():
        p, l = m(torch.from_numpy(np.ascontiguousarray(pcm)).cuda(), return_logits=True)
    torch.cuda.synchronize()
    return p.cpu().numpy(), l.cpu().numpy()


def test_native_library_is_loaded():
    from uit_mobile_amd import _lib
    assert _lib.load().uitk_version().startswith(b"uitk")
    loaded = open("/proc/self/maps").read()
    assert "libuitk.so" in loaded


@pytest.mark.parametrize("name", ["uit_xxxs", "uit_xxs", "uit_xs"])
def test_golden_synth_1s(golden, name):
    """Configs #2/#3 shapes at golden size: 8 clips incl. silence + full-scale tone (batch-global top_db floor)."""
    g = golden(f"synth1s_{name}")
    m, _ = build(name, g["seed"])
    mel = m.melspec_db(torch.from_numpy(g["pcm"]).cuda()).cpu().numpy()
    assert_mel_close(mel, g["mel_db"], scale=2.0)
    assert mel.min() == pytest.approx(mel.max() - 120.0, abs=1e-4)
    p, l = run(m, g["pcm"])
    assert np.abs(l - g["logits"]).max() < TOL
    assert np.abs(p - g["probs"]).max() < TOL
    # float input (what inference.py / the collate hand over) must give the same result as int16 PCM
    p32, _ = run(m, g["pcm"].astype(np.float32) / 32768.0)
    assert np.array_equal(p32, p)


@pytest.mark.parametrize("name", ["uit_xxxs", "uit_xs"])
def test_golden_stage_tokens(golden, name):
    """Residual stream after the patch embed and after EVERY block (4 for 3XS, 12 for XS) against the reference's own stream
    (hooks on its blocks, make_golden.py), centred (conftest.centred: the kernel keeps its stream centred)."""
    g = golden(f"synth1s_{name}")
    m, _ = build(name, g["seed"])
    _, toks = m.debug_tokens(torch.from_numpy(g["pcm"][:2]).cuda())
    toks = toks.cpu().numpy()[:, 0]
    assert toks.shape == g["stages"][:, 0].shape
    # fp16 operand rounding of O(1..3) activations: ~2e-3 abs per stage, not growing with depth (fp32 residual)
    d = np.abs(centred(toks) - centred(g["stages"][:, 0])).max(axis=(0, 2, 3))
    assert d.max() < 8e-3, d


@pytest.mark.parametrize("name", ["uit_xs", "uit_xxxs"])
def test_golden_trained_like_checkpoint(golden, name):
    """Parity at trained-checkpoint magnitudes: the reference's logits span -10.6..+3.8 (XS) / -14.2..+2.8 (3XS), LayerNorm
    gains 0.3..2.5, a residual stream with a common per-token offset and massive-activation channels.  1e-3 on logits AND
    probabilities against the fixtures the imported reference produced; per-block centred tokens at the fp16 operand level."""
    g = golden(f"trained_{name}")
    assert np.abs(g["logits"]).max() >= 10.0
    m, _ = build_trained(name, g["seed"])
    p, l = run(m, g["pcm"])
    dl, dp = np.abs(l - g["logits"]).max(), np.abs(p - g["probs"]).max()
    print(f"trained-like {name}: max|dlogit| {dl:.2e} max|dprob| {dp:.2e}")
    assert dl < TOL and dp < TOL
    _, toks = m.debug_tokens(torch.from_numpy(g["pcm"][:2]).cuda())
    toks = toks.cpu().numpy()[:, 0]
    assert np.abs(toks.mean(-1)).max() < 5e-3                     # the kernel's stream is centred (pack.cpp)
    # fp16 operand rounding relative to streams whose bulk is O(3) and whose massive channels reach 70: 2e-2 abs
    assert np.abs(centred(toks) - centred(g["stages"][:, 0])).max() < 2e-2


@pytest.mark.parametrize("name", ["uit_xs", "uit_xxxs"])
def test_f16x2_split_weights_cut_the_error_at_trained_magnitudes(golden, name):
    """precision='f16x2' (every weight as hi + lo fp16 terms, two MFMAs per fragment): the weight-rounding term that dominates the
    fp16 error at trained-checkpoint magnitudes (6.6e-4 of 7.1e-4 on the trained-like XS) is gone; what remains is
    the fp16 rounding of the activations.  Asserted: 4e-4 on logits (f16: 5.9e-4 / 4.1e-4 measured) and at most 0.6x f16's own
    error on the same box, 1e-3 everywhere else the f16 mode is tested at golden size (init scale, short clips, two crops)."""
    g = golden(f"trained_{name}")
    m2, _ = build_trained(name, g["seed"], prec="f16x2")
    m1, _ = build_trained(name, g["seed"], prec="f16")
    p2, l2 = run(m2, g["pcm"])
    p1, l1 = run(m1, g["pcm"])
    d2, d1 = np.abs(l2 - g["logits"]).max(), np.abs(l1 - g["logits"]).max()
    print(f"trained-like {name}: max|dlogit| f16x2 {d2:.2e} vs f16 {d1:.2e}; max|dprob| {np.abs(p2 - g['probs']).max():.2e}")
    assert d2 < 4e-4 and d2 < 0.65 * d1
    assert np.abs(p2 - g["probs"]).max() < 2.5e-4
    _, toks = m2.debug_tokens(torch.from_numpy(g["pcm"][:2]).cuda())
    assert np.abs(centred(toks.cpu().numpy()[:, 0]) - centred(g["stages"][:, 0])).max() < 2e-2


SWEEP_SEEDS = list(range(200, 212))            # the 12 trained-like checkpoints per model of tools/seed_sweep.py (round 2: f16 worst 1.13e-3)


def held_out_inputs(golden, with_names=False):
    """Inputs no calibration has seen -- round 5: the WIDE set.  The bench batch (noise / tones / silent tails),
    ALL ELEVEN recordings of the reference's samples/ (their first second; the two shorter ones end in digital silence), two members
    of every adversarial family of uit_mobile_amd.synth (impulse trains, single clicks, DC offsets, hard clipping, +-1 LSB dither,
    Nyquist alternation, speech-like AM/FM stacks at -40 dBFS and louder, clipped noise, square waves, silence + burst, exponential
    sweeps, bin-edge tone pairs, re-struck resonances) and round 4's four odd signals (square wave, silence, brown noise, chirp)."""
    from bench import synth_batch
    from uit_mobile_amd.synth import adversarial_pcm_i16
    g = golden("heldout_wide")
    t = np.arange(16000) / 16000.0
    rng = np.random.default_rng(5)
    odd = np.stack([0.3 * np.sign(np.sin(2 * np.pi * 200 * t)), np.zeros(16000), np.cumsum(rng.standard_normal(16000)) / 200.0,
                    0.8 * np.sin(2 * np.pi * (100 + 3000 * t) * t)])
    wavs = np.zeros((int(g["wav_count"]), 16000), np.int16)
    for i in range(len(wavs)):
        w = g[f"wav{i:02d}_pcm"][:16000]
        wavs[i, :len(w)] = w
    adv, adv_names = adversarial_pcm_i16(1, 2)
    x = np.concatenate([(synth_batch(24, 16000, 0, "cpu", False).numpy() * 32768.0).round().astype(np.int16), wavs, adv,
                        np.clip(np.round(odd * 32768), -32768, 32767).astype(np.int16)])
    names = [f"bench#{i}" for i in range(24)] + [f"wav{i:02d}" for i in range(len(wavs))] + adv_names + ["square200", "silence", "brown", "chirp"]
    assert len(names) == len(x)
    return (x, names) if with_names else x


@pytest.mark.parametrize("name", ["uit_xs", "uit_xxs", "uit_xxxs"])
@pytest.mark.parametrize("prec", ["auto", "f16l6"])
def test_parity_mode_all_sweep_seeds(golden, name, prec):
    """THE parity statement.  Precision 'auto' -- the factories' default: the plain fp16 kernel on weights rounded against the operand
    statistics of the library's own calibration signals (uitk_calibrate_weights), held to the library's absolute budget on its held-out
    clips and rebuilt on split weights when over it -- holds 1e-3 on logits AND probabilities on ALL twelve trained-like checkpoints of
    every model (|logit| up to 14.4) on the WIDE held-out set (round 5: 67 clips -- bench batch, all eleven recordings of the
    reference, two members of fourteen adversarial families, four odd signals).  The worst input per model is printed; how many
    checkpoints the budget check escalated is printed too (expected: none).  Round 3's parity mode 'f16l6' (low-order weight term on
    the MX-fp6 MFMA, 1.26x the encoder time) stays built and is held to its own bound."""
    assert getattr(models, name)(outputdim=537, target_length=102).precision == "auto"       # what the factories, inference.py and bench.py run
    x, names = held_out_inputs(golden, with_names=True)
    xd = torch.from_numpy(x).cuda()
    worst_l, worst_p, over, worst_at, escalated = 0.0, 0.0, [], None, []
    for seed in SWEEP_SEEDS:
        m, sd = build_trained(name, seed, prec=prec)
        assert m.precision == prec
        with torch.no_grad():
            p, l = m(xd, return_logits=True)
        r = Oracle(sd).forward(x)
        d = np.abs(l.cpu().numpy().reshape(r["logits"].shape) - r["logits"]).reshape(len(x), -1).max(1)
        dl = float(d.max())
        dp = float(np.abs(p.cpu().numpy() - r["probs"]).max())
        if dl > worst_l:
            worst_at = (seed, names[int(d.argmax())])
        worst_l, worst_p = max(worst_l, dl), max(worst_p, dp)
        if dl >= TOL or dp >= TOL:
            over.append((seed, dl, dp))
        if prec == "auto":
            eff = m.effective_precision()
            if eff != "f16c":
                escalated.append((seed, eff, m.calibration_report().get("reason")))
            else:                                           # a batch this small runs on 4-wave encoder workgroups (uitk_encode picks the
                m.set_option("enc_waves", 8)                # shape by batch size): the 8-wave instantiation bench.py times gives the same bits
                with torch.no_grad():
                    p8, l8 = m(xd, return_logits=True)
                assert torch.equal(p8, p) and torch.equal(l8, l)
        del m
    print(f"{prec} {name}: worst max|dlogit| {worst_l:.2e} (checkpoint {worst_at[0]}, input {worst_at[1]}), worst max|dprob| {worst_p:.2e} over "
          f"{len(SWEEP_SEEDS)} checkpoints, {len(x)} held-out clips; escalated: {escalated}")
    assert not over, over
    assert worst_l < (7e-4 if prec == "auto" else 8e-4) and worst_p < 2e-4


@pytest.mark.parametrize("name,act", [("audio_transformer_h128_d6_m3_relu", "relu"), ("audio_transformer_h128_d6_m3", "gelu"),
                                      ("audio_transformer_h128_d4_m3", "gelu")])
def test_full_attention_factories_parity_mode_all_sweep_seeds(golden, name, act):
    """The same statement for the full-width-attention / GELU factories in the default mode (round 4: 'f16c' for every family; round 3
    needed split weights, f16x2 at 1.63x the encoder time, for these): all twelve trained-like checkpoints inside 1e-3 on logits and
    probabilities (|logit| up to 14.9) on held-out inputs.  Round 3, f16x2: worst 5.4e-4 / 4.9e-4 / 6.5e-4; single fp16 weights rounded
    to nearest: 1.2e-3 / 1.2e-3 / 1.6e-3 (profiles/round3/seed_sweep_*.log)."""
    x = held_out_inputs(golden)
    xd = torch.from_numpy(x).cuda()
    worst_l, worst_p = 0.0, 0.0
    for seed in SWEEP_SEEDS:
        sd = make_trained_like_state_dict(name, seed, attention="full")
        m = getattr(models, name)(outputdim=537, target_length=102)
        assert m.precision == "auto"
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        with torch.no_grad():
            p, l = m.cuda().eval()(xd, return_logits=True)
        r = Oracle(sd, act=act).forward(x)
        worst_l = max(worst_l, float(np.abs(l.cpu().numpy().reshape(r["logits"].shape) - r["logits"]).max()))
        worst_p = max(worst_p, float(np.abs(p.cpu().numpy() - r["probs"]).max()))
        del m
    print(f"f16c {name}: worst max|dlogit| {worst_l:.2e}, worst max|dprob| {worst_p:.2e} over {len(SWEEP_SEEDS)} checkpoints")
    # round 4: one d4-GELU checkpoint (seed 206) sat at 9.7e-4 -- the fp16 ACTIVATION floor of this family (split weights: 8.7e-4).  Round 5:
    # operand centring (the calibration mean of LayerNorm-1's output leaves the operand before it is rounded; k - E[k], v - E[v] by
    # bias algebra) took that floor away: measured 6.1e-4 / 6.6e-4 / 5.7e-4 -- asserted at 8e-4
    assert worst_l < 8e-4 and worst_p < 2.5e-4


@pytest.mark.parametrize("name", ["uit_xs", "uit_xxxs"])
def test_parity_mode_fixtures_and_shapes(golden, name):
    """precision='f16l6' against the fixtures of the imported reference: trained-like checkpoint (tighter than f16's own bound),
    per-block centred tokens, and -- for the other shapes of the path -- init scale, a short (masked) clip and the two-crop sample."""
    g = golden(f"trained_{name}")
    m, _ = build_trained(name, g["seed"], prec="f16l6")
    m1, _ = build_trained(name, g["seed"], prec="f16")
    p, l = run(m, g["pcm"])
    p1, l1 = run(m1, g["pcm"])
    d, d1 = np.abs(l - g["logits"]).max(), np.abs(l1 - g["logits"]).max()
    print(f"trained-like {name}: max|dlogit| f16l6 {d:.2e} vs f16 {d1:.2e}; max|dprob| {np.abs(p - g['probs']).max():.2e}")
    assert d < 5e-4 and d < 0.85 * d1 and np.abs(p - g["probs"]).max() < 2.5e-4
    _, toks = m.debug_tokens(torch.from_numpy(g["pcm"][:2]).cuda())
    toks = toks.cpu().numpy()[:, 0]
    assert np.abs(toks.mean(-1)).max() < 5e-3
    assert np.abs(centred(toks) - centred(g["stages"][:, 0])).max() < 2e-2
    if name == "uit_xs":
        gi = golden("synth1s_uit_xs")
        mi, _ = build("uit_xs", gi["seed"], prec="f16l6")
        pi, li = run(mi, gi["pcm"])
        assert np.abs(li - gi["logits"]).max() < 2.5e-4 and np.abs(pi - gi["probs"]).max() < 1e-4
        gs, pcm = golden("trained_samples_uit_xs"), golden("samples_uit_xs")
        ms, _ = build_trained("uit_xs", gs["seed"], prec="f16l6")
        ps, ls = run(ms, pcm["water_000_pcm"][None])
        assert ls.shape[1] == 2 and np.abs(ls[0] - gs["water_000_logits"]).max() < 6e-4 and np.abs(ps[0] - gs["water_000_probs"]).max() < 2.5e-4
    else:
        gs = golden("trained_short_uit_xxxs")
        ms, _ = build_trained("uit_xxxs", gs["seed"], prec="f16l6")
        ps, ls = run(ms, gs["pcm"])
        assert np.abs(ls - gs["logits"]).max() < 6e-4 and np.abs(ps - gs["probs"]).max() < 2.5e-4


def test_parity_mode_ragged_and_full_size():
    """precision='f16l6' on ragged batch sizes (odd last wave, partial workgroup) against the oracle, and at the bench size:
    permutation of the clips permutes the rows bit for bit (even offsets keep the row parity), a sampled subset matches the oracle."""
    m, sd = build("uit_xxxs", 5, prec="f16l6")
    o = Oracle(sd)
    for B in (1, 3, 17, 33):
        pcm = synth_pcm_i16(B, 16000, seed=40 + B)
        p, l = run(m, pcm)
        r = o.forward(pcm)
        assert np.abs(p - r["probs"]).max() < TOL and np.abs(l - r["logits"]).max() < TOL
    pcm = synth_pcm_i16(4096, 16000, seed=41)
    p, _ = run(m, pcm)
    perm = np.random.default_rng(3).permutation(2048) * 2
    idx = np.stack([perm, perm + 1], 1).reshape(-1)                 # clip pairs move together: row parity is kept
    p2, _ = run(m, pcm[idx])
    assert np.array_equal(p2, p[idx])
    sub = np.random.default_rng(4).choice(4096, 8, replace=False)
    assert np.abs(p[sub] - o.forward(pcm[sub])["probs"]).max() < TOL


def test_precision_margin_recommends_by_checkpoint(golden):
    """UITBase.precision_margin: the f16 / f16l6 logit difference (= the weight-rounding term single fp16 weights carry) says whether
    a checkpoint is safe on round-to-nearest fp16 weights: init-scale weights -> 'f16'; the worst trained-like seed of the sweep (f16
    error 1.1e-3 at |logit| 12) -> 'f16l6'.  The factories default to 'f16c' (calibrated rounding), which needs no such decision."""
    x = torch.from_numpy(synth_pcm_i16(24, 16000, seed=3)).cuda()
    assert models.uit_xs(outputdim=537, target_length=102).precision == "auto"
    assert models.audio_transformer_h128_d4_m3(outputdim=537, target_length=102).precision == "auto"
    assert models.audio_transformer_h128_d4_m3(outputdim=537, target_length=102, pooling="dm").precision == "auto"
    m, _ = build("uit_xs", 13, prec="f16")
    r = m.precision_margin(x)
    assert r["recommended"] == "f16" and r["max_abs_logit_diff"] < 3e-4 and m.precision == "f16"
    m, _ = build_trained("uit_xs", 205, prec="f16")
    r = m.precision_margin(x)
    print("precision_margin, trained-like seed 205:", r)
    assert r["recommended"] == "f16l6" and r["max_abs_logit"] > 10.0 and r["max_abs_logit_diff"] > 3e-4
    m, _ = build("uit_xs", 13, prec="fp8")
    with pytest.raises(ValueError):
        m.precision_margin(x)


def test_f16x2_golden_sizes(golden):
    """The split-weight mode through the other shapes of the path: init-scale XS (silence + tone batch), a short trained-like clip
    (masked tokens), and the reference's two-crop sample clip."""
    g = golden("synth1s_uit_xs")
    m, _ = build("uit_xs", g["seed"], prec="f16x2")
    p, l = run(m, g["pcm"])
    assert np.abs(l - g["logits"]).max() < 2e-4 and np.abs(p - g["probs"]).max() < 1e-4
    g = golden("trained_short_uit_xxxs")
    m, _ = build_trained("uit_xxxs", g["seed"], prec="f16x2")
    p, l = run(m, g["pcm"])
    assert np.abs(l - g["logits"]).max() < 5e-4 and np.abs(p - g["probs"]).max() < 2.5e-4
    g, pcm = golden("trained_samples_uit_xs"), golden("samples_uit_xs")
    m, _ = build_trained("uit_xs", g["seed"], prec="f16x2")
    p, l = run(m, pcm["water_000_pcm"][None])
    assert l.shape[1] == 2 and np.abs(l[0] - g["water_000_logits"]).max() < 5e-4 and np.abs(p[0] - g["water_000_probs"]).max() < 2.5e-4


def test_golden_trained_like_checkpoint_on_the_reference_sample_clips(golden):
    """Real audio (the reference's water / keyword sample clips, two crops each for the water clips) through the trained-like
    XS checkpoint: reference logits down to -11."""
    g, pcm = golden("trained_samples_uit_xs"), golden("samples_uit_xs")
    m, _ = build_trained("uit_xs", g["seed"])
    for key in ("water_000", "water_003", "85b877b5_nohash_0"):
        p, l = run(m, pcm[f"{key}_pcm"][None])
        assert l.shape[1] == g[f"{key}_logits"].shape[0]
        assert np.abs(l[0] - g[f"{key}_logits"]).max() < TOL
        assert np.abs(p[0] - g[f"{key}_probs"]).max() < TOL


def test_golden_trained_like_short_clip(golden):
    """Short clips leave masked padding tokens in the wave; with the sharp attention of trained-like weights their scores must
    not set the softmax's running max (per-key masking of the max in the SHORT instantiation of the encoder)."""
    g = golden("trained_short_uit_xxxs")
    m, _ = build_trained("uit_xxxs", g["seed"])
    p, l = run(m, g["pcm"])
    assert np.isfinite(l).all()
    assert np.abs(l - g["logits"]).max() < TOL and np.abs(p - g["probs"]).max() < TOL


@pytest.mark.parametrize("name,act", [("audio_transformer_h128_d4_m3_relu", "relu"), ("audio_transformer_h128_d4_m3", "gelu"),
                                      ("audio_transformer_h128_d6_m3_relu", "relu"), ("audio_transformer_h128_d6_m3", "gelu")])
def test_golden_other_factories(golden, name, act):
    """the non-bottleneck factories of models/uit.py (:514-579) on the ATT = 1 instantiation of the encoder:
    full-width Attention (2 heads x 64: three ring stages per head, scores on four chained K = 16 MFMAs, 20 stages per block),
    exact-erf GELU or ReLU.  d4 at init scale, d6 at trained-like magnitudes (|logit| 12); 1e-3 on logits and probabilities
    against the imported reference, per-block tokens at the fp16 operand level, ragged batch and a short clip vs the oracle."""
    g = golden(f"variant_{name}")
    trained = bool(g["trained"])
    sd = (make_trained_like_state_dict if trained else make_state_dict)(name, int(g["seed"]), attention="full")
    m = getattr(models, name)(outputdim=537, target_length=102)
    assert m.precision == "auto"                                      # the default of every family (round 3: split weights, f16x2, for ATT = 1 / GELU)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    p, l = run(m, g["pcm"])
    dl, dp = np.abs(l - g["logits"]).max(), np.abs(p - g["probs"]).max()
    print(f"variant {name}: |logit|max {np.abs(g['logits']).max():.1f} max|dlogit| {dl:.2e} max|dprob| {dp:.2e}")
    # the 1e-3 gate on logits and probabilities at init scale AND at trained-like magnitudes (|logit|max 12.3-12.7).  Round 2 ran these
    # factories on single fp16 weights and held the trained-like logits to a measured 1.5e-3; printed below for the record.
    assert dl < TOL and dp < TOL
    _, toks = m.debug_tokens(torch.from_numpy(g["pcm"][:2]).cuda())
    assert np.abs(centred(toks.cpu().numpy()[:, 0]) - centred(g["stages"][:, 0])).max() < (2e-2 if trained else 8e-3)
    o = Oracle(sd, act=act)
    for pcm in (synth_pcm_i16(5, 16000, seed=3), synth_pcm_i16(3, 8000, seed=4), synth_pcm_i16(2, 16384, seed=5)):
        ref = o.forward(pcm)
        p, l = run(m, pcm)
        assert np.abs(l - ref["logits"]).max() < TOL and np.abs(p - ref["probs"]).max() < TOL
    m16 = getattr(models, name)(outputdim=537, target_length=102, precision="f16")
    m16.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    p16, l16 = run(m16.to("cuda").eval(), g["pcm"])
    print(f"   single fp16 weights on the same fixture: max|dlogit| {np.abs(l16 - g['logits']).max():.2e}")
    assert np.abs(p16 - g["probs"]).max() < TOL and (trained or np.abs(l16 - g["logits"]).max() < TOL)


def test_golden_pooling_dm(golden):
    """pooling='dm' (models/uit.py:405-412) -- per time patch the mean over the four frequency patches through
    outputlayer and the sigmoid, then the mean over the time patches -- on the POOL = 1 instantiation: init-scale 3XS (8 clips,
    short clips with 3 time patches, two crops) and trained-like XS against the imported reference, 1e-3 on the probabilities
    (this pooling has no pre-sigmoid output: asking for logits is an error)."""
    g = golden("pooling_dm")
    sd = make_state_dict("uit_xxxs", int(g["xxxs_seed"]))
    m = models.uit_xxxs(outputdim=537, target_length=102, pooling="dm")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    for pcm, want in ((g["pcm"], g["xxxs_probs"]), (g["xxxs_short_pcm"], g["xxxs_short_probs"]), (g["xxxs_crops_pcm"], g["xxxs_crops_probs"])):
        with torch.no_grad():
            p = m(torch.from_numpy(pcm).cuda()).cpu().numpy()
        assert np.abs(p - want).max() < TOL
    with pytest.raises(ValueError):
        m(torch.from_numpy(g["pcm"]).cuda(), return_logits=True)
    o = Oracle(sd, pooling="dm")
    for B in (1, 3, 17):
        pcm = synth_pcm_i16(B, 16000, seed=60 + B)
        with torch.no_grad():
            p = m(torch.from_numpy(pcm).cuda()).cpu().numpy()
        assert np.abs(p - o.forward(pcm)["probs"]).max() < TOL
    sd = make_trained_like_state_dict("uit_xs", int(g["xs_trained_seed"]))
    for prec in ("f16c", "f16", "f16l6"):
        m = models.uit_xs(outputdim=537, target_length=102, pooling="dm", precision=prec)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        with torch.no_grad():
            p = m.to("cuda").eval()(torch.from_numpy(g["pcm"]).cuda()).cpu().numpy()
        print(f"pooling dm, trained-like uit_xs ({prec}): max|dprob| {np.abs(p - g['xs_trained_probs']).max():.2e}")
        assert np.abs(p - g["xs_trained_probs"]).max() < TOL
    # and on the full-width-attention GELU factory, against the oracle
    sd = make_state_dict("audio_transformer_h128_d4_m3", 41, attention="full")
    m = models.audio_transformer_h128_d4_m3(outputdim=537, target_length=102, pooling="dm")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    pcm = synth_pcm_i16(5, 16000, seed=8)
    with torch.no_grad():
        p = m.to("cuda").eval()(torch.from_numpy(pcm).cuda()).cpu().numpy()
    assert np.abs(p - Oracle(sd, act="gelu", pooling="dm").forward(pcm)["probs"]).max() < TOL


def test_golden_pooling_token(golden):
    """pooling='token' (models/uit.py:389-392, 399-401; the UITBase default) on the POOL = 2 instantiation -- one clip
    per wave, column 0 = cls_token + token_pos_embed, columns 1..24 the patch tokens, the rest dead -- against the imported
    reference: init-scale 3XS (incl. short clips), trained-like XS (|logit| 12), the full-width-attention GELU factory; ragged
    batches and two crops against the oracle."""
    g = golden("pooling_token")
    for tag, name, trained, att, act in (("xxxs", "uit_xxxs", False, "bneck", "relu"), ("xs_trained", "uit_xs", True, "bneck", "relu"),
                                         ("h128", "audio_transformer_h128_d4_m3", False, "full", "gelu")):
        sd = dict((make_trained_like_state_dict if trained else make_state_dict)(name, int(g[f"{tag}_seed"]), attention=att))
        sd["cls_token"] = g[f"{tag}_cls_token"]
        # every family in its parity mode (round 2 held plain f16 to its measured 1.5e-3 on the trained-like fixture)
        prec = "auto"                                             # the factories' default
        m = getattr(models, name)(outputdim=537, target_length=102, pooling="token")
        assert m.precision == prec
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        m = m.to("cuda").eval()
        p, l = run(m, g["pcm"])
        dl, dp = np.abs(l - g[f"{tag}_logits"]).max(), np.abs(p - g[f"{tag}_probs"]).max()
        print(f"pooling token {name} ({prec} -> {m.effective_precision()}): |logit|max {np.abs(g[f'{tag}_logits']).max():.1f} max|dlogit| {dl:.2e} max|dprob| {dp:.2e}; "
              f"report {m.calibration_report()}")
        assert dl < TOL and dp < TOL
        if trained:
            # round 5: with pooling 'token' the head reads ONE row -- what is over the library's budget there is fp16 ACTIVATION rounding
            # (ln2 / hidden operands of the cls row), which split weights do not touch: 'auto' escalates to 'f16a' (those two operands of
            # the cls tile as hi + lo terms): 7.8e-4 (round 4, f16c) -> asserted at 7e-4
            assert m.effective_precision() in ("f16c", "f16a")
            if m.effective_precision() == "f16a":
                assert dl < 7e-4
        if tag == "xxxs":
            p, l = run(m, g["xxxs_short_pcm"])
            assert np.abs(l - g["xxxs_short_logits"]).max() < TOL
            o = Oracle(sd, act=act, pooling="token")
            for pcm in (synth_pcm_i16(1, 16000, seed=1), synth_pcm_i16(9, 16000, seed=2), synth_pcm_i16(3, 16384, seed=3)):
                ref = o.forward(pcm)
                p, l = run(m, pcm)
                assert np.abs(l - ref["logits"]).max() < TOL and np.abs(p - ref["probs"]).max() < TOL


@pytest.mark.parametrize("pooling", ["token", "dm"])
def test_full_attention_factory_pooling_at_trained_magnitudes(pooling):
    """Round 3: split fp16 weights (f16x2) with pooling 'token' / 'dm' on the full-width-attention GELU factory at trained-checkpoint
    magnitudes, against the C oracle (ragged batch, a short clip): the 1e-3 gate -- single fp16 weights printed beside it."""
    name = "audio_transformer_h128_d6_m3"
    sd = dict(make_trained_like_state_dict(name, 205, attention="full"))
    o = Oracle(sd, act="gelu", pooling=pooling)
    res = {}
    for prec in ("f16c", "f16x2", "f16") + (("f16a", "auto") if pooling == "token" else ("auto",)):
        m = getattr(models, name)(outputdim=537, target_length=102, pooling=pooling, precision=prec)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        m = m.to("cuda").eval()
        worst = 0.0
        for pcm in (synth_pcm_i16(9, 16000, seed=71), synth_pcm_i16(3, 8000, seed=72)):
            ref = o.forward(pcm)
            with torch.no_grad():
                out = m(torch.from_numpy(pcm).cuda(), return_logits=True) if pooling == "token" else (m(torch.from_numpy(pcm).cuda()), None)
            worst = max(worst, float(np.abs(out[0].cpu().numpy() - ref["probs"]).max()))
            if pooling == "token":
                worst = max(worst, float(np.abs(out[1].cpu().numpy().reshape(ref["logits"].shape) - ref["logits"]).max()))
        res[prec] = worst
    print(f"full attention + GELU, pooling {pooling}, trained-like: " + ", ".join(f"{k} {v:.2e}" for k, v in res.items()) + f"; auto ran {m.effective_precision()}")
    assert res["auto"] < 7e-4
    assert res["f16x2"] < 8e-4 and res["f16c"] < 8e-4         # round 4: f16c 8.2e-4 (token); round 5 with operand centring: 6.2e-4


@pytest.mark.parametrize("avg", ["mean", "max"])
def test_golden_ten_crop_clip(golden, avg):
    """AudioSet-style 10 s clips: 160 000 samples -> 1001 frames -> 10 crops, tail re-cropped (models/uit.py:468-488;
    evaluate.py:68-111), mean and max over the crops' probabilities."""
    g = golden("long10_uit_xxxs")
    m, _ = build("uit_xxxs", g["seed"], eval_avg=avg)
    assert m.num_crops(g["pcm"].shape[1]) == 10
    p, l = run(m, g["pcm"])
    assert l.shape == g["logits"].shape
    assert np.abs(l - g["logits"]).max() < TOL
    assert np.abs(p - g[f"probs_{avg}"]).max() < TOL


@pytest.mark.parametrize("offset", [30.0, 1000.0])
def test_common_residual_offset_cannot_break_the_layernorm_statistics(offset):
    """A checkpoint whose positional embeddings shift EVERY channel of every token by `offset` (per-token mean >> per-token
    deviation).  The reference's two-pass LayerNorm does not care; the kernel's one-pass variance E[x^2] - mean^2 would lose
    all its digits at offset 1000 -- if pack.cpp did not keep the residual stream centred (LayerNorm is shift-invariant)."""
    sd = dict(make_trained_like_state_dict("uit_xxxs", 5))
    sd["time_pos_embed"] = (np.asarray(sd["time_pos_embed"]) + offset).astype(np.float32)
    m = models.uit_xxxs(outputdim=537, target_length=102)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    pcm = synth_pcm_i16(6, 16000, seed=3)
    ref = Oracle(sd).forward(pcm)
    p, l = run(m, pcm)
    # the ORACLE (fp32, like the reference) is the noisier side at offset 1000: its own stream carries 1000 +- 3 in fp32
    tol = TOL if offset < 100 else 5e-3
    assert np.abs(l - ref["logits"]).max() < tol and np.abs(p - ref["probs"]).max() < TOL


@pytest.mark.parametrize("name", ["uit_xs", "uit_xxxs"])
def test_golden_reference_sample_clips(golden, name):
    """BASELINE config #1: water_000.wav (16384 samples -> 2 crops), plus the GSC clip and the evaluate.py pad rule."""
    g = golden(f"samples_{name}")
    m, _ = build(name, g["seed"])
    for key in ("water_000", "water_003", "85b877b5_nohash_0"):
        p, l = run(m, g[f"{key}_pcm"][None])
        assert l.shape[1] == g[f"{key}_logits"].shape[0]
        assert np.abs(l[0] - g[f"{key}_logits"]).max() < TOL
        assert np.abs(p[0] - g[f"{key}_probs"]).max() < TOL
    p, _ = run(m, g["gsc_padded_pcm"][None])
    assert np.abs(p[0] - g["gsc_padded_probs"]).max() < TOL


@pytest.mark.parametrize("avg", ["mean", "max"])
def test_golden_long_clip_multicrop(golden, avg):
    g = golden(f"long_{avg}_uit_xxxs")
    m, _ = build("uit_xxxs", g["seed"], eval_avg=avg)
    assert m.num_crops(g["pcm"].shape[1]) == 4
    p, l = run(m, g["pcm"])
    assert np.abs(l - g["logits"]).max() < TOL
    assert np.abs(p - g["probs"]).max() < TOL


def test_golden_other_kwargs_and_center_false(golden):
    """Round 3 (VERDICT r2 missing #4): the UITBase kwargs the wrapper used to refuse, against fixtures of the imported reference:
    qkv_bias=False + LayerScale (init_values) + init_bn=False on one model (exact pack-time folds), and center=False in the front-end
    (97 frames for a 1 s clip; 104 frames -> two crops for 17 000 samples), in the parity mode and in plain f16."""
    from uit_mobile_amd.synth import make_kwargs_variant_state_dict
    g = golden("kwargs_uit_xxxs")
    sd = make_kwargs_variant_state_dict("uit_xxxs", int(g["seed"]))
    for prec in ("f16l6", "f16"):
        m = models.uit_xxxs(outputdim=537, target_length=102, qkv_bias=False, init_values=0.1, init_bn=False, precision=prec)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        p, l = run(m.to("cuda").eval(), g["pcm"])
        print(f"kwargs variant {prec}: max|dlogit| {np.abs(l - g['logits']).max():.2e}")
        assert np.abs(l - g["logits"]).max() < TOL and np.abs(p - g["probs"]).max() < TOL
    g = golden("nocenter_uit_xxxs")
    sd = make_state_dict("uit_xxxs", int(g["seed"]), 2.0)
    m = models.uit_xxxs(outputdim=537, target_length=102, center=False)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    for tag, crops in (("1s", 1), ("2crop", 2)):
        p, l = run(m, g[f"{tag}_pcm"])
        assert l.shape[1] == crops
        assert np.abs(l - g[f"{tag}_logits"]).max() < TOL and np.abs(p - g[f"{tag}_probs"]).max() < TOL
        with torch.no_grad():
            mel = m.melspec_db(torch.from_numpy(g[f"{tag}_pcm"]).cuda()).cpu().numpy()
        assert_mel_close(mel, g[f"{tag}_mel_db"])
    with pytest.raises(ValueError):                          # shorter than one window
        m(torch.zeros(1, 400).cuda())


def test_golden_win_size_shorter_than_n_fft(golden):
    """Round 4: win_size = 400 with n_fft = 512 (the window zero-padded to n_fft at pack time, as torch.stft does): mel, logits and
    probabilities against the fixtures of the imported reference, centre and no-centre framing, one and two crops; a window longer
    than n_fft is refused."""
    from uit_mobile_amd.synth import hann_periodic
    g = golden("win400_uit_xxxs")
    sd = dict(make_state_dict("uit_xxxs", int(g["seed"]), 2.0))
    sd["front_end.0.spectrogram.window"] = hann_periodic(400)
    for tag, crops, center in (("1s", 1, True), ("2crop", 2, True), ("nocenter", 1, False)):
        m = models.uit_xxxs(outputdim=537, target_length=102, win_size=400, center=center)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        m = m.to("cuda").eval()
        p, l = run(m, g[f"{tag}_pcm"])
        assert l.shape[1] == crops
        assert np.abs(l - g[f"{tag}_logits"]).max() < TOL and np.abs(p - g[f"{tag}_probs"]).max() < TOL
        with torch.no_grad():
            mel = m.melspec_db(torch.from_numpy(g[f"{tag}_pcm"]).cuda()).cpu().numpy()
        assert_mel_close(mel, g[f"{tag}_mel_db"])
    with pytest.raises(ValueError):
        models.uit_xxxs(outputdim=537, target_length=102, win_size=640).to("cuda").eval()(torch.zeros(1, 16000).cuda())


def test_golden_short_clip(golden):
    g = golden("short_uit_xxxs")
    m, _ = build("uit_xxxs", g["seed"])
    p, l = run(m, g["pcm"])
    assert np.abs(l - g["logits"]).max() < TOL
    assert np.abs(p - g["probs"]).max() < TOL


def test_golden_527_classes_and_pos_resize(golden):
    g = golden("synth1s_uit_xxs_527")
    m, _ = build("uit_xxs", g["seed"], outputdim=527)
    p, _ = run(m, g["pcm"])
    assert p.shape == (3, 527) and np.abs(p - g["probs"]).max() < TOL
    g = golden("posresize_uit_xxxs")                      # checkpoint with 10 time-pos entries -> truncated to 6
    m, _ = build("uit_xxxs", g["seed"], ckpt_target_length=160)
    p, _ = run(m, g["pcm"])
    assert np.abs(p - g["probs"]).max() < TOL


def test_scaled_weights_precision_modes(golden):
    """x3 weights: logits span +-2.5.  fp16 operands stay within 1e-3 on logits AND probabilities (measured 5.5e-4 / 1.3e-4).
    ... and outside it as soon as the weights grow: bf16 blocks carry 8x the operand rounding (measured 4.9e-3 on logits,
    1.2e-3 on probabilities here; 5e-3 on a trained-like checkpoint by emulation).  These bounds are the MEASURED error of a
    non-parity mode, asserted so that it cannot silently get worse -- not a parity claim."""
    g = golden("synth1s_uit_xs_scale3")
    m, _ = build("uit_xs", g["seed"], g["scale"])
    p, l = run(m, g["pcm"])
    assert np.abs(p - g["probs"]).max() < TOL
    assert np.abs(l - g["logits"]).max() < TOL
    m, _ = build("uit_xs", g["seed"], g["scale"], prec="bf16")
    p, l = run(m, g["pcm"])
    assert np.abs(p - g["probs"]).max() < 2e-3
    assert np.abs(l - g["logits"]).max() < 8e-3


@pytest.mark.parametrize("fixture,trained,tol_emu,tol_logit,tol_prob", [("synth1s_uit_xs", False, 4e-3, 2.5e-2, 6e-3),
                                                                       ("trained_uit_xs", True, 2.5e-2, 2e-1, 4e-2)])
def test_fp8_mode_is_the_intended_quantised_network_and_its_error_is_on_record(golden, fixture, trained, tol_emu, tol_logit, tol_prob):
    """BASELINE config #5 (precision='fp8': e4m3 weights + activations with MX block scales on the 16x16x128 MFMA for
    qkv / fc1 / fc2).  Two separate statements:
      (a) the kernel computes the INTENDED quantised network: GPU vs the fp64 restatement of that network (tests/fp8_emu.py)
          -- measured 1.3e-3 (init weights) / 8.2e-3 (trained-like), i.e. rounding flips of single e4m3 activations;
      (b) what the quantisation costs against the reference: measured max |dlogit| 7.7e-3 / 6.6e-2, max |dprob| 1.9e-3 / 1.2e-2.
    (b) is 8x - 70x OUTSIDE the 1e-3 parity gate: fp8 is a measured throughput mode, never a parity mode."""
    import fp8_emu
    g = golden(fixture)
    m, sd = (build_trained("uit_xs", g["seed"], prec="fp8") if trained else build("uit_xs", g["seed"], prec="fp8"))
    p, l = run(m, g["pcm"])
    assert np.isfinite(l).all()
    mel, _ = Oracle(sd).melspec_db(g["pcm"])
    emu = fp8_emu.forward(sd, fp8_emu.mel_bn_from_oracle(sd, mel), 12).numpy()
    d_emu, d_l, d_p = np.abs(l[:, 0] - emu).max(), np.abs(l - g["logits"]).max(), np.abs(p - g["probs"]).max()
    print(f"fp8 {fixture}: GPU vs emulation {d_emu:.2e}; vs reference max|dlogit| {d_l:.2e} max|dprob| {d_p:.2e}")
    assert d_emu < tol_emu
    assert d_l < tol_logit and d_p < tol_prob
    assert d_l > TOL                                   # on record: outside the parity gate


def test_fp8_weights_rounded_on_their_grid_by_gptq(golden):
    """Round 5, BASELINE config #5 "fp8 weights": (a) precision 'fp8c' = the fp8 kernel (e4m3 weights AND activations
    on the 16x16x128 MFMA) with the qkv / fc1 / fc2 weights GPTQ-rounded to the MX e4m3 grid against the calibration statistics --
    against 'fp8' (nearest on the same grid) on the trained-like fixture of the imported reference; (b) precision 'fp8w' = what the
    config names literally: fp8-grid WEIGHTS (GPTQ), expanded to fp16 at pack time, the f16 kernel on fp16 activations.  Both are
    measured modes far outside the 1e-3 gate; the bounds below are their measured errors, asserted so they cannot silently get worse."""
    g = golden("trained_uit_xs")
    res = {}
    for prec in ("fp8", "fp8c", "fp8w", "f16"):
        m, sd = build_trained("uit_xs", g["seed"], prec=prec)
        p, l = run(m, g["pcm"])
        assert np.isfinite(l).all()
        res[prec] = (float(np.abs(l - g["logits"]).max()), float(np.abs(p - g["probs"]).max()), float(np.sqrt(np.mean((l - g["logits"]) ** 2))))
        if prec in ("fp8c", "fp8w"):
            rep = m.calibration_report()
            print(f"{prec}: library's held-out check {rep}")
            assert rep is not None and m.effective_precision().startswith("fp8")
    print("trained-like XS vs the imported reference, (max|dlogit|, max|dprob|, rms dlogit): " + ", ".join(f"{k} {v[0]:.2e}/{v[1]:.2e}/{v[2]:.2e}" for k, v in res.items()))
    assert res["fp8c"][2] < res["fp8"][2] and res["fp8c"][0] < 2e-1                # GPTQ on the grid beats nearest on the grid (rms over all logits)
    assert res["fp8w"][0] < 1e-1 and res["fp8w"][0] > TOL                           # weights-only: activations stay fp16
    # the weights-only mode runs on any family / pooling (it is the f16 kernel)
    sd = make_trained_like_state_dict("audio_transformer_h128_d4_m3", 206, attention="full")
    m = models.audio_transformer_h128_d4_m3(outputdim=537, target_length=102, precision="fp8w", pooling="token")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    pcm = synth_pcm_i16(5, 16000, seed=3)
    p, l = run(m.cuda().eval(), pcm)
    r = Oracle(sd, act="gelu", pooling="token").forward(pcm)
    print(f"fp8w, full attention + GELU, pooling token: max|dlogit| {np.abs(l - r['logits']).max():.2e}")
    assert np.abs(l - r["logits"]).max() < 2e-1


def test_bf16_mode_init_scale(golden):
    """bf16 operands (a measured, non-parity mode: DESIGN.md 2.1).  With the split-weight patch embed and the split head the
    plain-bf16 blocks are what is left: inside 1e-3 at init-scale weights (measured 7.5e-4 / 1.9e-4) ..."""
    g = golden("synth1s_uit_xs")
    m, _ = build("uit_xs", g["seed"], prec="bf16")
    p, l = run(m, g["pcm"])
    assert np.abs(p - g["probs"]).max() < TOL
    assert np.abs(l - g["logits"]).max() < TOL


@pytest.mark.parametrize("B", [1, 2, 3, 15, 16, 17, 33])
def test_ragged_batch_sizes_vs_oracle(B):
    """Odd batches leave a wave with one clip and a workgroup partly empty."""
    m, sd = build("uit_xxxs", 21)
    pcm = synth_pcm_i16(B, 16000, seed=B)
    ref = Oracle(sd).forward(pcm)
    p, l = run(m, pcm)
    assert np.abs(l - ref["logits"]).max() < TOL
    assert np.abs(p - ref["probs"]).max() < TOL


@pytest.mark.parametrize("name,B,seed", [("uit_xxxs", 4096, 22), ("uit_xs", 8192, 23)])
def test_full_size_properties(name, B, seed):
    """BASELINE config sizes -- #2: UiT-3XS x 4096, #3: UiT-XS x 8192 -- through size-independent properties instead of an
    oracle run: (1) batch-permutation equivariance, (2) outputs are probabilities, (3) a sampled subset that contains the
    batch's loudest clip (hence the batch's top_db floor) matches the oracle."""
    m, sd = build(name, seed)
    pcm = synth_pcm_i16(B, 16000, seed=5)
    # clip 0 = full-scale 1 kHz square wave: the loudest clip by a margin, so any subset containing it has the
    # batch's amax and therefore the batch's top_db floor
    pcm[0] = np.where((np.arange(16000) // 8) % 2 == 0, 32767, -32768).astype(np.int16)
    p, l = run(m, pcm)
    assert np.isfinite(p).all() and (p > 0).all() and (p < 1).all()
    perm = np.random.default_rng(0).permutation(B)
    pp, _ = run(m, pcm[perm])
    # a clip's slot inside its wave changes the fp32 summation order of the token mean-pool (~1e-7); the pooled
    # vector is then rounded to fp16 for the head GEMM, where a flipped rounding shows as ~2e-5: not bitwise
    # (bitwise when the slot PARITY is kept: test_two_ranks_on_one_gpu_...)
    assert np.abs(pp - p[perm]).max() < 1e-4
    idx = np.arange(0, B, 257)                            # includes clip 0 and tone / silence / quiet classes
    sub = pcm[idx]
    assert idx[0] == 0 and (idx % 8 == 2).any()
    ref = Oracle(sd).forward(sub)
    assert np.abs(p[idx] - ref["probs"]).max() < TOL
    assert np.abs(l[idx] - ref["logits"]).max() < TOL


@pytest.mark.parametrize("precision", ["f16c", "f16l6", "f16", "f16x2"])
def test_run_to_run_bit_identical(precision):
    """The same batch twelve times: every output bit identical.  Round 3 found a build that was not (an inline-asm maximum read MFMA
    accumulators before they were written -- DESIGN.md 2.2; only roundings moved, so every parity test passed): this is its guard.
    UiT-3XS x 4096 (config #2's size: two waves per SIMD contend for issue slots, which is what made the hazard timing-dependent)."""
    sd = make_state_dict("uit_xxxs", 5)
    m = models.uit_xxxs(outputdim=537, target_length=102, precision=precision)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.cuda().eval()
    x = torch.from_numpy(synth_pcm_i16(4096, 16000, seed=41)).cuda()
    with torch.no_grad():
        first = m(x).clone()
        differing = sum(int(not torch.equal(first, m(x))) for _ in range(12))
    assert differing == 0


def test_pass_distribution_does_not_change_results():
    """The front-end's work distribution (asymmetric static split of the persistent grid) decides WHO transforms a
    frame, never how: the calibrated default, a symmetric split and an extreme share give bit-identical probabilities and mel; so
    does the generic top_db clamp path of the encoder against its plain-max path.  Options go through uitk_set_option (the library
    reads no environment)."""
    pcm = synth_pcm_i16(8192, 16000, seed=9)
    pcm[3] = 0
    x = torch.from_numpy(pcm).cuda()
    outs = []
    for opts in ({}, {"fe_asym_pm": 500}, {"fe_asym_pm": 700}, {"fe_asym_pm": 300}, {"enc_generic_clamp": 1}, {"fe_tail_skip": 0},
                 {"enc_waves": 4}, {"enc_waves": 8}):
        m, _ = build("uit_xxxs", 31)
        for k, v in opts.items():
            m.set_option(k, v)
        with torch.no_grad():
            outs.append((m(x).cpu().numpy(), m.melspec_db(x[:64]).cpu().numpy()))
    for p, mel in outs[1:]:
        assert np.array_equal(p, outs[0][0]) and np.array_equal(mel, outs[0][1])
    # the encoder's workgroup shape follows the batch size (4-wave workgroups up to 8 clip rows per CU): small batches give the bits
    # of either explicit shape, short clips (masked time patches) included
    for xs in (x[:100], x[:2048], x[:2049], x[:7, :9000], x[:4096 + 101], x[:6144]):      # (the last two: full 8-wave rounds + a 4-wave tail launch)
        trio = []
        for waves in (0, 4, 8):
            m, _ = build("uit_xxxs", 31, fe_autotune=False)         # (no pass-split measurement for these shapes: a later test counts them)
            m.set_option("enc_waves", waves)
            with torch.no_grad():
                trio.append(m(xs).cpu().numpy())
        assert np.array_equal(trio[0], trio[1]) and np.array_equal(trio[0], trio[2])
    for prec in ("bf16", "fp8"):                        # the other precisions that have both shapes
        trio = []
        for waves in (0, 4, 8):
            m, _ = build("uit_xxxs", 31, prec=prec, fe_autotune=False)
            m.set_option("enc_waves", waves)
            with torch.no_grad():
                trio.append(m(x[:100]).cpu().numpy())
        assert np.array_equal(trio[0], trio[1]) and np.array_equal(trio[0], trio[2]), prec
    m, _ = build("uit_xxxs", 31)
    with pytest.raises(ValueError):                     # unknown option: UITK_ERR_UNSUPPORTED when the handle is created
        m.set_option("no_such_option", 1)
        m(x[:2])


def test_leftover_frames_still_decide_the_floor_when_they_are_the_loudest():
    """Round 4: the front-end skips the FFT of a clip's leftover frames (1 s clips: frames 96..100, which never reach the encoder) when a
    Parseval bound shows that they cannot raise the batch amax.  The bound must never skip a frame that DOES: here the batch's loudest
    content sits in the last 40 ms of one otherwise quiet clip, so the top_db floor -- visible in every silent region of the batch -- is
    set by a leftover frame; mel, probabilities and logits against the oracle, and bit-identical with the skip switched off."""
    B, T = 600, 16000                                       # enough passes for the persistent grid; clip 7 carries the burst
    pcm = (synth_pcm_i16(B, T, seed=77).astype(np.float32) * 0.02).astype(np.int16)      # a quiet batch (-34 dB re full scale)
    t = np.arange(T - 15500)
    pcm[7, 15500:] = np.round(30000.0 * np.sin(2 * np.pi * 1234.0 * t / 16000.0)).astype(np.int16)    # frames 97..100 only
    pcm[8] = 0                                              # digital silence: shows the floor
    m, sd = build("uit_xxxs", 12)
    x = torch.from_numpy(pcm).cuda()
    with torch.no_grad():
        p1, l1 = m(x, return_logits=True)
        mel1 = m.melspec_db(x[:16])
    o = Oracle(sd)
    ref_mel, amax = o.melspec_db(pcm)
    assert ref_mel[7, :, 97:].max() == pytest.approx(amax, abs=1e-3) and ref_mel[:, :, :96].max() < amax - 10.0    # the burst IS the amax
    sub = np.r_[0:16]
    # (the oracle's floor is that of the batch it is given: hand it the whole batch for the mel, a subset with clip 7 for the head)
    assert_mel_close(mel1.cpu().numpy(), Oracle(sd).melspec_db(pcm[:16])[0])
    assert mel1.cpu().numpy()[8].max() == pytest.approx(amax - 120.0, abs=2e-3)           # silence sits on the floor the burst sets
    ref = o.forward(pcm[sub])
    assert np.abs(p1.cpu().numpy()[sub] - ref["probs"]).max() < TOL and np.abs(l1.cpu().numpy()[sub, 0] - ref["logits"][:, 0]).max() < TOL
    m.set_option("fe_tail_skip", 0)
    with torch.no_grad():
        p0, l0 = m(x, return_logits=True)
    assert torch.equal(p0, p1) and torch.equal(l0, l1)


def test_errors_map_to_exceptions():
    m, _ = build("uit_xxxs", 1)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 200, device="cuda"))             # shorter than the reflect pad
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 16000))                          # CPU tensor: no CPU path
    m.eval_avg = "median"
    with pytest.raises(ValueError):
        m(torch.zeros(1, 16000, device="cuda"))           # models/uit.py:488
    m.train()
    m.eval_avg = "mean"
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 16000, device="cuda"))


def test_options_round_trip_and_frontend_split_is_measured_once_per_process():
    """uitk_set_option / uitk_get_option: what is set is what is read; an unknown name is an error.  The front-end's pass split
     is measured by the first full-size call of a batch shape on this device (uitk_config.fe_autotune, on the caller's
    buffer and stream: no allocation, no device synchronisation) and is then shared by EVERY handle of the process: a second model
    reserves in well under a millisecond, runs its first call without measuring anything and reports the same share.  Like every
    option it changes no result bit."""
    import time
    m, _ = build("uit_xxxs", 3, prec="f16l6")
    x = torch.from_numpy(synth_pcm_i16(4096, 16000, seed=9)).cuda()
    with torch.no_grad():
        base = m(x).clone()
    torch.cuda.synchronize()
    share = m.get_option("fe_asym_pm")[0]
    assert share in (500, 550, 590, 630, 670)
    m2, _ = build("uit_xxxs", 3, prec="f16l6")
    m2._handle(torch.device("cuda", torch.cuda.current_device()))       # (packing + upload: not what is timed)
    m2.reserve(4096, 16000)                                             # first reserve of this handle: allocates the workspace
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m2.reserve(4096, 16000)
    dt_reserve = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.no_grad():
        out2 = m2(x)
    dt_enqueue = time.perf_counter() - t0                               # enqueue only: nothing is measured, nothing synchronises
    torch.cuda.synchronize()
    print(f"second handle: reserve {dt_reserve * 1e6:.0f} us, first forward enqueued in {dt_enqueue * 1e6:.0f} us, share {m2.get_option('fe_asym_pm')[0]}")
    assert dt_reserve < 1e-3 and dt_enqueue < 5e-3
    assert m2.get_option("fe_asym_pm")[0] == share and torch.equal(out2, base)
    # explicit tuning entry (uitk_tune_frontend) and fe_autotune=False: the explicit call measures, implicit calls never do
    m3, _ = build("uit_xxxs", 3, prec="f16l6", fe_autotune=False)
    x3 = torch.from_numpy(synth_pcm_i16(2048, 16000, seed=9)).cuda()     # another passes-per-workgroup bucket: not measured yet
    with torch.no_grad():
        o3 = m3(x3).clone()
    assert m3.get_option("fe_asym_pm")[0] == -1                         # the built-in split: nothing was measured for this shape
    m3.tune_frontend(x3)
    assert m3.get_option("fe_asym_pm")[0] in (500, 550, 590, 630, 670)
    with torch.no_grad():
        assert torch.equal(o3, m3(x3))
    m.set_option("fe_asym_pm", 520)
    assert m.get_option("fe_asym_pm")[0] == 520 and m.get_option("enc_waves")[0] in (0, 8)
    with torch.no_grad():
        assert torch.equal(base, m(x))
    with pytest.raises(ValueError):                       # UITK_ERR_UNSUPPORTED / UITK_ERR_INVALID_ARG -> ValueError (_lib.check)
        m.get_option("no_such_option")
    with pytest.raises(ValueError):
        m.set_option("fe_asym_pm", 999)
    with pytest.raises(ValueError):                       # validated without a live handle too
        build("uit_xxxs", 3)[0].set_option("no_such_option", 1)


def test_calibration_statistics_match_the_restatement():
    """uitk_calibrate_weights measures, on the device (csrc/calibrate.hip), the second moments of the operand of every single-weight
    linear; the numpy restatement of the reference's arithmetic (tests/calib_ref.py) on the same clips must give the same matrices --
    BNeckAttention + ReLU, and full-width attention + GELU with the cls token of pooling 'token'."""
    from calib_ref import calib_stats
    cal = synth_pcm_i16(16, 16000, seed=4242)
    for name, att, act, pooling, inner in (("uit_xxxs", "bneck", "relu", "mean", 32), ("audio_transformer_h128_d4_m3", "full", "gelu", "token", 128)):
        sd = make_trained_like_state_dict(name, 203, attention=att)
        m = getattr(models, name)(outputdim=537, target_length=102, pooling=pooling)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        m = m.cuda().eval().calibrate(torch.from_numpy(cal).cuda())
        got = m.calib_stats()
        want = calib_stats(sd, cal, 4, inner=inner, act=act, cls=pooling == "token")
        assert got.shape == want.shape
        off, worst = 0, 0.0
        for n in [256] + [128, inner, 128, 384] * 4:
            g_, w_ = got[off:off + n * n], want[off:off + n * n]
            worst = max(worst, float(np.linalg.norm(g_ - w_) / np.linalg.norm(w_)))
            off += n * n
        assert got.size - off == 4 * 128                                # round 5: the LayerNorm-1 means behind the second moments
        worst = max(worst, float(np.abs(got[off:] - want[off:]).max() / np.abs(want[off:]).max()))
        print(f"{name}: worst relative Frobenius error of the 17 statistics matrices {worst:.2e}")
        assert worst < 2e-3


def test_calibrated_rounding_with_the_callers_clips_and_weight_edits(golden):
    """model.calibrate(x): the caller's clips instead of the library's signals -- both stay inside the gate on held-out inputs and both
    beat round-to-nearest fp16 weights on a checkpoint where those are over the gate; the calibrated image follows the weights: an
    in-place edit (no_grad mul_, which no hook sees) re-packs AND re-calibrates on the next call (stale images)."""
    x = held_out_inputs(golden)
    xd = torch.from_numpy(x).cuda()
    sd = make_trained_like_state_dict("uit_xs", 207)                    # the sweep's worst checkpoint for plain fp16 (1.2e-3)
    ref = Oracle(sd).forward(x)
    err = {}
    for tag, prec, cal in (("nearest", "f16", None), ("own signals", "f16c", None), ("caller's clips", "f16c", synth_pcm_i16(48, 16000, seed=777))):
        m = models.uit_xs(outputdim=537, target_length=102, precision=prec)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        m = m.cuda().eval()
        if cal is not None:
            m.calibrate(torch.from_numpy(cal).cuda())
        with torch.no_grad():
            p, l = m(xd, return_logits=True)
        err[tag] = float(np.abs(l.cpu().numpy().reshape(ref["logits"].shape) - ref["logits"]).max())
        assert np.abs(p.cpu().numpy() - ref["probs"]).max() < 2.5e-4
    print("max|dlogit| on held-out inputs, trained-like XS seed 207:", {k: f"{v:.2e}" for k, v in err.items()})
    assert err["own signals"] < 6e-4 and err["caller's clips"] < 6e-4 and err["nearest"] > 1.5 * max(err["own signals"], err["caller's clips"])
    with pytest.raises(ValueError):
        build("uit_xxxs", 1, prec="f16l6")[0].calibrate(None)           # calibration belongs to 'f16c'
    # weight edit: the last model halves its classifier bias in place; the next call must see it (and still be calibrated)
    with torch.no_grad():
        m.outputlayer[1].bias.mul_(0.5)
    sd2 = dict(sd)
    sd2["outputlayer.1.bias"] = np.asarray(sd["outputlayer.1.bias"]) * 0.5
    ref2 = Oracle(sd2).forward(x[:6])
    with torch.no_grad():
        p2, l2 = m(xd[:6], return_logits=True)
    assert np.abs(l2.cpu().numpy().reshape(ref2["logits"].shape) - ref2["logits"]).max() < 6e-4
    with pytest.raises(ValueError):
        m.calibrate(torch.from_numpy(synth_pcm_i16(4, 8000, seed=1)).cuda())(xd[:2])      # clips shorter than target_length frames
    # calibration is deterministic: two handles built from the same weights hold the same statistics and give the same bits (what
    # makes the ranks of a sharded run agree without exchanging anything); long calibration clips count as their crops
    pair = []
    for _ in range(2):
        mm = models.uit_xxxs(outputdim=537, target_length=102)
        mm.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_trained_like_state_dict("uit_xxxs", 203).items()}, strict=True)
        mm = mm.cuda().eval()
        with torch.no_grad():
            pair.append((mm(xd).clone(), mm.calib_stats()))
    assert torch.equal(pair[0][0], pair[1][0]) and np.array_equal(pair[0][1], pair[1][1])
    mm.calibrate(torch.from_numpy(synth_pcm_i16(6, 64000, seed=31)).cuda())               # 4 s clips: 4 crops each = 24 calibration rows
    with torch.no_grad():
        p4 = mm(xd)
    ref3 = Oracle(make_trained_like_state_dict("uit_xxxs", 203)).forward(x)
    assert np.abs(p4.cpu().numpy() - ref3["probs"]).max() < 2.5e-4 and not np.array_equal(mm.calib_stats(), pair[1][1])


def test_calibration_acceptance_check(golden):
    """uitk_calibrate_weights installs its image only after a held-out comparison with the nearest-rounded image (round 4): both run
    the product kernels on sixteen clips of the library's second signal family; the reference there is the fp32 forward of
    calibrate.hip finished in double on the host.  (a) The report's numbers are those an oracle run would give: same order as the
    C oracle's errors on the test's own held-out inputs.  (b) With calib_accept_pct = 1 the calibrated image is refused: the handle
    answers bit for bit like precision 'f16' and says so.  (c) A calibration on digital silence -- statistics of rank ~24 -- either
    passes the check or is refused, and stays inside the gate either way.  (d) Every pooling and the full-attention / GELU family
    go through the same check (pooling 'dm' is compared on probabilities)."""
    x = held_out_inputs(golden)
    xd = torch.from_numpy(x).cuda()
    sd = make_trained_like_state_dict("uit_xs", 207)
    tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    ref = Oracle(sd).forward(x)

    def make(prec, **opts):
        m = models.uit_xs(outputdim=537, target_length=102, precision=prec)
        m.load_state_dict(tsd, strict=True)
        m = m.cuda().eval()
        for k, v in opts.items():
            m.set_option(k, v)
        return m

    def err_of(m):
        with torch.no_grad():
            p, l = m(xd, return_logits=True)
        return float(np.abs(l.cpu().numpy().reshape(ref["logits"].shape) - ref["logits"]).max()), p.clone()

    # (a)
    m = make("f16c")
    e_cal, p_cal = err_of(m)
    rep = m.calibration_report()
    e_near, p_near = err_of(make("f16"))
    print(f"acceptance check, XS seed 207: {rep}; C oracle on the test's held-out inputs: nearest {e_near:.2e}, calibrated {e_cal:.2e}")
    assert rep is not None and rep["accepted"] and m.get_option("weights_calibrated")[0] == 1
    assert rep["max_calibrated"] < 0.7 * rep["max_nearest"] and rep["rms_calibrated"] < rep["rms_nearest"]
    assert 0.4 < rep["max_nearest"] / e_near < 2.5 and 0.4 < rep["max_calibrated"] / e_cal < 2.5
    assert make("f16").calibration_report() is None
    # (b)
    m = make("f16c", calib_accept_pct=1)
    e_ref, p_ref = err_of(m)
    rep = m.calibration_report()
    assert rep is not None and not rep["accepted"] and m.get_option("weights_calibrated")[0] == 0
    assert torch.equal(p_ref, p_near) and e_ref == e_near
    m = make("f16c", calib_accept_pct=0)                                # no check: installed unconditionally, nothing to report
    e0, p0 = err_of(m)
    assert m.calibration_report() is None and m.get_option("weights_calibrated")[0] == 1 and torch.equal(p0, p_cal)
    # (c)
    m = make("f16c").calibrate(torch.zeros((32, 16000), dtype=torch.int16, device="cuda"))
    e_sil, p_sil = err_of(m)
    rep = m.calibration_report()
    print(f"calibrated on digital silence: {rep}; C oracle, held-out inputs: {e_sil:.2e}")
    assert rep is not None and (m.get_option("weights_calibrated")[0] == 1) == rep["accepted"]
    if rep["accepted"]:
        assert rep["max_calibrated"] <= 1.05 * rep["max_nearest"]
    else:
        assert torch.equal(p_sil, p_near)
    assert e_sil < 1.3 * e_near
    # (d)
    for name, att, pooling in (("uit_xxxs", "bneck", "dm"), ("uit_xxxs", "bneck", "token"), ("audio_transformer_h128_d4_m3", "full", "mean")):
        mm = getattr(models, name)(outputdim=537, target_length=102, pooling=pooling)
        mm.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_trained_like_state_dict(name, 203, attention=att).items()}, strict=True)
        mm = mm.cuda().eval()
        with torch.no_grad():
            mm(xd[:4])
        rep = mm.calibration_report()
        print(f"{name} pooling {pooling}: {rep}")
        assert rep is not None and 0 < rep["max_calibrated"] < 1e-3 and 0 < rep["max_nearest"] < 3e-3
        if not rep["escalated"]:
            assert (mm.get_option("weights_calibrated")[0] == 1) == rep["accepted"]


HELDOUT_MODELS = (("xs207", "uit_xs", 207, "bneck", "relu"), ("xxxs203", "uit_xxxs", 203, "bneck", "relu"),
                  ("d4gelu206", "audio_transformer_h128_d4_m3", 206, "full", "gelu"))


@pytest.mark.parametrize("tag,name,seed,att,act", HELDOUT_MODELS)
def test_wide_held_out_set_against_the_imported_reference(golden, tag, name, seed, att, act):
    """Round 5: the default mode against what the IMPORTED reference answered (tests/golden/heldout_wide.npz) on
    all eleven recordings of its samples/ -- native length, one forward call each, one or two crops -- and on one member of every
    adversarial family as one batch, for three trained-like checkpoints: the sweep's worst XS checkpoint for nearest rounding (207),
    a 3XS one, and the full-attention GELU checkpoint that sat at the gate in round 4 (d4, seed 206)."""
    g = golden("heldout_wide")
    sd = make_trained_like_state_dict(name, seed, attention=att)
    m = getattr(models, name)(outputdim=537, target_length=102)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval().prepare("cuda")
    worst_wav, worst_p = (0.0, None), 0.0
    for i in range(int(g["wav_count"])):
        p, l = run(m, g[f"wav{i:02d}_pcm"][None])
        assert l.shape[1:] == g[f"{tag}_wav{i:02d}_logits"].shape
        d = float(np.abs(l[0] - g[f"{tag}_wav{i:02d}_logits"]).max())
        worst_wav = max(worst_wav, (d, f"wav{i:02d}"))
        worst_p = max(worst_p, float(np.abs(p[0] - g[f"{tag}_wav{i:02d}_probs"]).max()))
    p, l = run(m, g["adv_pcm"])
    from uit_mobile_amd.synth import adversarial_pcm_i16
    names = adversarial_pcm_i16(0, 1)[1]
    d = np.abs(l - g[f"{tag}_adv_logits"]).reshape(len(names), -1).max(1)
    worst_p = max(worst_p, float(np.abs(p - g[f"{tag}_adv_probs"]).max()))
    print(f"{tag} ({m.effective_precision()}): vs the imported reference -- recordings worst {worst_wav[0]:.2e} ({worst_wav[1]}), adversarial worst "
          f"{d.max():.2e} ({names[int(d.argmax())]}), probabilities {worst_p:.2e}; report {m.calibration_report()}")
    assert worst_wav[0] < 8e-4 and d.max() < 8e-4 and worst_p < 2.5e-4


@pytest.mark.parametrize("tag,name,seed,att,act", HELDOUT_MODELS)
def test_wide_held_out_set_on_short_and_long_clips(golden, tag, name, seed, att, act):
    """The other clip shapes of the path on the wide held-out signals at trained magnitudes (default mode vs the C oracle): SHORT clips
    (0.5 s: 51 frames -> 3 time patches, masked padding tokens -- the instantiations whose masks became accumulator start values in
    round 5; the calibration means were measured on full clips) and LONG clips (4 s: 4 crops with the re-cropped tail, mean over crops)."""
    from uit_mobile_amd.synth import adversarial_pcm_i16
    g = golden("heldout_wide")
    sd = make_trained_like_state_dict(name, seed, attention=att)
    m = getattr(models, name)(outputdim=537, target_length=102)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval().prepare("cuda")
    o = Oracle(sd, act=act)
    adv, names = adversarial_pcm_i16(3, 1)
    wavs = np.stack([np.pad(g[f"wav{i:02d}_pcm"], (0, max(0, 16384 - len(g[f"wav{i:02d}_pcm"]))))[:16384] for i in range(int(g["wav_count"]))])
    res = {}
    for label, x in (("short 0.5 s", np.concatenate([adv[:, 4000:12000], wavs[:, 2000:10000]])),
                     ("long 4 s", np.concatenate([np.tile(adv, (1, 4))[:, :64000], np.tile(wavs, (1, 4))[:, :64000]]))):
        p, l = run(m, x)
        r = o.forward(x)
        assert l.shape == r["logits"].shape
        d = np.abs(l - r["logits"]).reshape(len(x), -1).max(1)
        res[label] = (float(d.max()), (names + [f"wav{i:02d}" for i in range(len(wavs))])[int(d.argmax())], float(np.abs(p - r["probs"]).max()))
    print(f"{tag} ({m.effective_precision()}): " + "; ".join(f"{k}: worst max|dlogit| {v[0]:.2e} on {v[1]}, max|dprob| {v[2]:.2e}" for k, v in res.items()))
    assert all(v[0] < 8e-4 and v[2] < 2.5e-4 for v in res.values())


@pytest.mark.parametrize("tag,name,seed,att,act", HELDOUT_MODELS + (("xs205", "uit_xs", 205, "bneck", "relu"),))
def test_random_search_over_adversarial_families(tag, name, seed, att, act):
    """Round 5: a seeded random search (~12 s per checkpoint) over the adversarial families -- every round draws
    one new member of each family (other periods, levels, pitches, positions) as one batch, alternating with mixed batches that put
    them next to loud noise (another batch-global floor) -- through the default mode and the C oracle; the worst input per
    checkpoint is reported and must stay inside the gate."""
    import time
    from uit_mobile_amd.synth import adversarial_pcm_i16
    sd = make_trained_like_state_dict(name, seed, attention=att)
    m = getattr(models, name)(outputdim=537, target_length=102)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval().prepare("cuda")
    o = Oracle(sd, act=act)
    worst, n, rounds, t0 = (0.0, None), 0, 0, time.time()
    while time.time() - t0 < 12.0 and rounds < 40:
        x, names = adversarial_pcm_i16(1000 + rounds, 1)
        if rounds % 2:
            x = np.concatenate([x, synth_pcm_i16(2, 16000, seed=rounds)])
            names = names + ["noise", "tone"]
        p, l = run(m, x)
        r = o.forward(x)
        d = np.abs(l.reshape(r["logits"].shape) - r["logits"]).reshape(len(x), -1).max(1)
        worst = max(worst, (float(d.max()), names[int(d.argmax())]))
        assert np.abs(p - r["probs"]).max() < 2.5e-4
        n, rounds = n + len(x), rounds + 1
    print(f"random search {tag} ({m.effective_precision()}): {n} clips in {rounds} rounds, worst max|dlogit| {worst[0]:.2e} on {worst[1]}")
    assert worst[0] < 9e-4


def test_auto_escalates_instead_of_falling_back_to_nearest(golden):
    """Round 5 (VERDICT r4 item 1a / ADVICE r4): the parity mode is a guarantee, not a report.  XS seed 207 is the checkpoint where
    nearest-rounded fp16 weights measure 1.2e-3.  (a) 'auto' with a healthy calibration stays 'f16c' inside the budget.  (b) A
    SABOTAGED calibration (calib_accept_pct = 1: the acceptance check refuses every calibrated image) no longer leaves the handle on
    the nearest image: the installed image is over the 6e-4 budget, 'auto' rebuilds the handle as 'f16l6', warns, reports it -- and
    the result is parity-grade, while an explicit 'f16c' under the same sabotage warns and answers bit for bit like 'f16'.  (c) A
    budget nobody can meet makes 'auto' MEASURE the split-weight build too and keep whichever is better (here the calibrated image: the
    rest is activation rounding); the full-attention GELU family escalates to 'f16x2'.  (d) At init
    scale a refused calibration is harmless (nearest is inside the budget): no escalation."""
    import warnings
    from uit_mobile_amd.models.uit import UitkParityWarning
    x = held_out_inputs(golden)
    xd = torch.from_numpy(x).cuda()
    sd = make_trained_like_state_dict("uit_xs", 207)
    tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    ref = Oracle(sd).forward(x)

    def make(prec, fac=models.uit_xs, w=tsd, **opts):
        m = fac(outputdim=537, target_length=102, precision=prec)
        m.load_state_dict(w, strict=True)
        m = m.cuda().eval()
        for k, v in opts.items():
            m.set_option(k, v)
        return m

    def err_of(m):
        with torch.no_grad():
            p, l = m(xd, return_logits=True)
        return float(np.abs(l.cpu().numpy().reshape(ref["logits"].shape) - ref["logits"]).max()), p.clone()

    # (a)
    with warnings.catch_warnings():
        warnings.simplefilter("error", UitkParityWarning)
        m = make("auto")
        e_auto, _ = err_of(m)
    rep = m.calibration_report()
    assert m.effective_precision() == "f16c" and not rep["escalated"] and rep["accepted"] and rep["max_calibrated"] <= rep["budget"] == pytest.approx(6e-4)
    e_near, p_near = err_of(make("f16"))
    # (b)
    with pytest.warns(UitkParityWarning, match="f16l6"):
        m = make("auto", calib_accept_pct=1)                               # the default check (full clips only): the split build is inside the budget
        e_esc, _ = err_of(m)
    rep = m.calibration_report()
    print(f"sabotaged calibration, XS seed 207: nearest {e_near:.2e}, auto (healthy) {e_auto:.2e}, auto (sabotaged -> {m.effective_precision()}) {e_esc:.2e}; report {rep}")
    assert m.effective_precision() == "f16l6" and rep["escalated"] and not rep["accepted"] and rep["max_nearest"] > rep["budget"]
    assert rep["self_check"] is not None and rep["self_check"][0] < rep["budget"]
    assert e_near > 9e-4 and e_esc < 8e-4 and e_auto < 7e-4
    # round 6, option calib_short_check (what prepare() switches on for a short-clip deployment): the held-out clips are also scored at
    # half length (12 of 24 tokens: less averaging).  There the split-weight build of this checkpoint is over the budget too, so the
    # ladder goes on -- measured: to 'f32', 3.0e-4 against the oracle
    with pytest.warns(UitkParityWarning, match="f16l6"):
        m = make("auto", calib_accept_pct=1, calib_short_check=1)
        e_esc2, _ = err_of(m)
    rep = m.calibration_report()
    print(f"   with the half-length leg of the check: auto (sabotaged -> {m.effective_precision()}) {e_esc2:.2e}; ladder {rep.get('ladder')}")
    assert m.effective_precision() in ("f16l6", "f32") and rep["escalated"] and "f16l6" in rep["ladder"] and e_esc2 < 8e-4
    assert rep["within_budget"] and rep["self_check"][0] <= rep["budget"]
    with pytest.warns(UitkParityWarning, match="does not escalate"):
        m = make("f16c", calib_accept_pct=1)
        e_c, p_c = err_of(m)
    assert m.effective_precision() == "f16" and torch.equal(p_c, p_near) and e_c == e_near
    # (c) a budget no fp16 mode can meet: the accepted calibrated image is over it, the split-weight build is measured on the same clips -- and,
    # being no better than the calibrated image (what is left is activation rounding), is NOT taken: the report says so.  With the fp32
    # rung switched off the calibrated image stays; with it (the default) the ladder ends in 'f32', the library's plain fp32 forward
    def make_nofp32(prec, **opts):
        m = models.uit_xs(outputdim=537, target_length=102, precision=prec, fp32_fallback=False)
        m.load_state_dict(tsd, strict=True)
        m = m.cuda().eval()
        for k, v in opts.items():
            m.set_option(k, v)
        return m
    with pytest.warns(UitkParityWarning, match="no better"):
        m = make_nofp32("auto", calib_budget_e6=1)
        err_of(m)
    rep = m.calibration_report()
    assert m.effective_precision() == "f16c" and rep["accepted"] and not rep["escalated"] and rep["within_budget"] is False
    assert rep["self_check_split"][0] >= rep["max_calibrated"]
    with pytest.warns(UitkParityWarning, match="'f32'"):
        m = make("auto", calib_budget_e6=1)
        e_f32, _ = err_of(m)
    rep = m.calibration_report()
    print(f"a budget of 1e-6: auto -> {m.effective_precision()}: {e_f32:.2e} against the C oracle on the held-out inputs; report {rep}")
    assert m.effective_precision() == "f32" and rep["escalated"] and rep["self_check"][0] < 5e-5 and e_f32 < 4e-4
    sdg = make_trained_like_state_dict("audio_transformer_h128_d4_m3", 206, attention="full")
    with pytest.warns(UitkParityWarning, match="f16x2"):
        m = make("auto", models.audio_transformer_h128_d4_m3, {k: torch.from_numpy(np.asarray(v)) for k, v in sdg.items()}, calib_accept_pct=1)
        with torch.no_grad():
            p, l = m(xd[:8], return_logits=True)
    assert m.effective_precision() in ("f16x2", "f32") and "f16x2" in m.calibration_report()["ladder"]      # (the sabotage also refuses the split build's centring: if
    rg = Oracle(sdg, act="gelu").forward(x[:8])                                                               #  that leaves it over the budget, the ladder goes on to 'f32')
    assert np.abs(l.cpu().numpy().reshape(rg["logits"].shape) - rg["logits"]).max() < TOL
    # (d)
    with warnings.catch_warnings():
        warnings.simplefilter("error", UitkParityWarning)
        m = make("auto", w={k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xs", 13).items()}, calib_accept_pct=1)
        with torch.no_grad():
            m(xd[:4])
    assert m.effective_precision() == "f16" and not m.calibration_report()["escalated"]


def test_prepare_takes_the_set_up_out_of_forward_and_allows_graph_capture():
    """Round 5: model.prepare(device, (B, T)) = handle + weight calibration + workspace + front-end tuning + one
    warm-up forward; after it a forward of that shape is enqueue-only and can be CAPTURED into a HIP graph whose replay gives the eager
    bits.  An unprepared model warns once (UitkSetupWarning) when its first forward has to calibrate; the pack-time entry points refuse
    to run under stream capture (UITK_ERR_INVALID_ARG before any work is enqueued)."""
    import ctypes
    import time
    import warnings
    from uit_mobile_amd import _lib
    from uit_mobile_amd.models.uit import UitkSetupWarning
    sd = make_trained_like_state_dict("uit_xxxs", 203)
    tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    x = torch.from_numpy(synth_pcm_i16(4096, 16000, seed=12)).cuda()
    m = models.uit_xxxs(outputdim=537, target_length=102)
    m.load_state_dict(tsd, strict=True)
    with warnings.catch_warnings():
        warnings.simplefilter("error", UitkSetupWarning)
        m = m.cuda().eval().prepare("cuda", (4096, 16000), dtype=torch.int16)
        t0 = time.perf_counter()
        with torch.no_grad():
            eager = m(x)
        dt = time.perf_counter() - t0                      # enqueue only: nothing is calibrated, tuned or allocated by the library
        torch.cuda.synchronize()
    assert dt < 5e-3, dt
    eager = eager.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.no_grad():
            m(x)                                           # (torch's own warm-up rule for graph capture: one run on the side stream)
    torch.cuda.current_stream().wait_stream(s)
    w, keep = m._host_weights()
    hptr, o2, L = m._handle(x.device), (ctypes.c_float * 2)(), _lib.load()
    with torch.cuda.graph(g):
        with torch.no_grad():
            out = m(x)
        # a pack-time entry point inside the capture: refused before it enqueues or allocates anything
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc_check = L.uitk_self_check(hptr, ctypes.byref(w), stream, o2)
        rc_cal = L.uitk_calibrate_weights(hptr, ctypes.byref(w), None, _lib.IN_I16, 0, 0, stream)
    assert rc_check == _lib.ERR_INVALID_ARG and rc_cal == _lib.ERR_INVALID_ARG
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    ref = Oracle(sd).forward(x[:6].cpu().numpy())           # rows 0..5 of a 4096-clip batch: the floor is the batch's, so compare loosely
    assert np.abs(out[:6].cpu().numpy() - ref["probs"]).max() < 5e-3
    m2 = models.uit_xxxs(outputdim=537, target_length=102)
    m2.load_state_dict(tsd, strict=True)
    with pytest.warns(UitkSetupWarning):
        with torch.no_grad():
            m2.cuda().eval()(x[:4])
    with warnings.catch_warnings():
        warnings.simplefilter("error", UitkSetupWarning)   # once per model
        with torch.no_grad():
            m2(x[:4])


def test_two_stage_state_keeps_its_handle_alive_and_rejects_a_rebuilt_one():
    """frontend() hands the raw model pointer to encode() through its state; if the handle was rebuilt in between
    (refresh / calibrate / load_state_dict / .to / an option that re-packs) the old pointer used to be freed -> use-after-free.  The
    state now carries the _Handle object (alive until used) and encode() refuses a state whose handle is no longer the live one."""
    m, sd = build("uit_xxxs", 7)
    x = torch.from_numpy(synth_pcm_i16(6, 16000, seed=5)).cuda()
    with torch.no_grad():
        want = m(x).clone()
        st = m.frontend(x)
        assert torch.equal(m.encode(st), want)
        st = m.frontend(x)
        m.refresh()
        with pytest.raises(RuntimeError, match="rebuilt"):
            m.encode(st)
        st = m.frontend(x)
        m.set_option("calib_accept_pct", 104)              # re-packs under the new rule
        with pytest.raises(RuntimeError, match="rebuilt"):
            m.encode(st)
        assert torch.equal(m.encode(m.frontend(x)), want)



def test_c_abi_consumer_in_plain_c(tmp_path):
    """The drop-in boundary is a C ABI (include/uitk.h): a consumer written in plain C99 (tests/native/cabi_consumer.c: gcc, no C++, no
    Python, no torch; the HIP runtime API only for its device buffers) creates a model from raw float arrays, calibrates, runs
    uitk_forward on int16 clips -- and writes, bit for bit, what the Python wrapper returns for the same checkpoint and clips."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    sd = make_trained_like_state_dict("uit_xxxs", 205)
    pcm = synth_pcm_i16(6, 16000, seed=77)
    order = ["front_end.0.spectrogram.window", "front_end.0.mel_scale.fb", "init_bn.1.weight", "init_bn.1.bias", "init_bn.1.running_mean",
             "init_bn.1.running_var", "patch_embed.proj.weight", "patch_embed.proj.bias", "time_pos_embed", "freq_pos_embed", "norm.weight",
             "norm.bias", "outputlayer.0.weight", "outputlayer.0.bias", "outputlayer.1.weight", "outputlayer.1.bias", "cls_token", "token_pos_embed"]
    for b in range(4):
        order += [f"blocks.{b}.{k}" for k in ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
                                              "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")]
    blob = tmp_path / "ckpt.bin"
    with open(blob, "wb") as f:
        f.write(np.array([4, 537, pcm.shape[0], pcm.shape[1]], np.int32).tobytes())
        for k in order:
            f.write(np.ascontiguousarray(np.asarray(sd[k]), dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(pcm).tobytes())
    exe = tmp_path / "cabi_consumer"
    lib = ROOT / "uit_mobile_amd"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{ROOT / 'include'}",
                    str(ROOT / "tests" / "native" / "cabi_consumer.c"), "-o", str(exe), f"-L{lib}", "-luitk", "-L/opt/rocm/lib", "-lamdhip64",
                    f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe), str(blob), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    print(r.stdout.strip(), r.stderr.strip())
    assert r.returncode == 0, r.stderr
    out = np.fromfile(tmp_path / "out.bin", np.float32)
    n = pcm.shape[0] * 537
    m, _ = build_trained("uit_xxxs", 205)
    with torch.no_grad():
        p, l = m(torch.from_numpy(pcm).cuda(), return_logits=True)
    assert np.array_equal(out[:n].reshape(-1, 537), p.cpu().numpy()) and np.array_equal(out[n:2 * n].reshape(-1, 537), l.cpu().numpy().reshape(-1, 537))
    rep = m.calibration_report()
    assert out[2 * n + 4] == 1.0 and np.isclose(out[2 * n + 1], rep["max_calibrated"], rtol=0, atol=0)
    # the ladder in C (round 5): a budget no fp16 mode meets -> the consumer follows the handle rung by rung (f16l6, then f32) and writes
    # what the wrapper's 'auto' writes under the same budget (which ends on the same last rung)
    r = subprocess.run([str(exe), str(blob), str(tmp_path / "out2.bin"), "1"], capture_output=True, text=True, timeout=300)
    print(r.stdout.strip(), r.stderr.strip())
    assert r.returncode == 0 and "rebuilt in precision 4" in r.stdout and "rebuilt in precision 8" in r.stdout, r.stderr
    out2 = np.fromfile(tmp_path / "out2.bin", np.float32)
    import warnings
    m8 = models.uit_xxxs(outputdim=537, target_length=102)
    m8.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m8 = m8.cuda().eval()
    m8.set_option("calib_budget_e6", 1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.no_grad():
            p8 = m8(torch.from_numpy(pcm).cuda())
    assert m8.effective_precision() == "f32" and np.array_equal(out2[:n].reshape(-1, 537), p8.cpu().numpy())


def test_inference_cli_counterpart_config1(golden, tmp_path, capsys):
    """BASELINE config #1 plumbing: `inference.py -m <ckpt> water_000.wav` through the counterpart CLI (own WAV reader,
    {'model','config'} checkpoint format of inference.py:42-48, label lookup, 'Keyword:' prefix)."""
    from uit_mobile_amd import inference
    from uit_mobile_amd.wav import write_wav_pcm16
    g = golden("samples_uit_xs")
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xs", int(g["seed"])).items()}
    torch.save({"model": sd, "config": {"model": "uit_xs", "num_classes": 537, "model_args": {"target_length": 102}}},
               tmp_path / "ckpt.pt")
    write_wav_pcm16(tmp_path / "water_000.wav", g["water_000_pcm"])
    inference.main([str(tmp_path / "water_000.wav"), "-m", str(tmp_path / "ckpt.pt"), "-k", "5"])
    out = capsys.readouterr().out.strip().splitlines()
    labels = inference.load_label_map()
    want = g["water_000_probs"]
    top = np.argsort(-want)[:5]
    assert out[0].startswith("=====") and len(out) == 6
    for line, idx in zip(out[1:], top):
        name = labels[int(idx)] if idx <= 526 else f"Keyword: {labels[int(idx)]}"
        assert line.startswith(name[:30])
        assert abs(float(line.split()[-1]) - want[idx]) < 1.1e-3


def test_evaluator_batch_step_pad_rule(golden, tmp_path):
    from uit_mobile_amd.evaluate import Evaluator
    g = golden("samples_uit_xs")
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xs", int(g["seed"])).items()}
    torch.save(sd, tmp_path / "bare.pt")                   # bare state_dict format (inference.py:37-40)
    ev = Evaluator(str(tmp_path / "bare.pt"))
    pcm = torch.from_numpy(g["85b877b5_nohash_0_pcm"].astype(np.float32) / 32768.0)[None]
    out, _ = ev._inference((pcm, torch.zeros(1, 537), torch.tensor([16000]), ["x"]), pad=True)
    assert np.abs(out.cpu().numpy()[0] - g["gsc_padded_probs"]).max() < TOL


def test_sharded_wrapper_single_rank_matches_plain_forward():
    """world_size 1: the amax hook runs between the two kernels (in-place on the device slots) and must not change
    the result; the gather is the identity."""
    from uit_mobile_amd.sharded import ShardedUiT
    m, sd = build("uit_xxxs", 31)
    pcm = synth_pcm_i16(24, 16000, seed=12)
    x = torch.from_numpy(pcm).cuda()
    plain = m(x)
    sharded = ShardedUiT(m)(x)
    torch.cuda.synchronize()
    assert torch.equal(plain, sharded)
    slots = m.amax_tensor(x.device)
    assert slots.shape == (64,) and not torch.isnan(slots).any() and torch.isfinite(slots.max())


def _rccl_worker(rank, world, port, q, gather):
    import os
    import torch.distributed as dist
    from uit_mobile_amd.sharded import ShardedUiT, shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        m, sd = build("uit_xxxs", 32)
        m = m.to(f"cuda:{rank}")
        pcm = synth_pcm_i16(16, 16000, seed=13)               # clip 1 is the loud tone, clip 2 has a silent tail
        lo, hi = shard_range(16, rank, world)
        out = ShardedUiT(m, gather=gather)(torch.from_numpy(pcm[lo:hi]).cuda())
        torch.cuda.synchronize()
        q.put((rank, out.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL)")
@pytest.mark.parametrize("gather", ["rccl", "peer"])
def test_two_rank_rccl_is_bitwise_the_single_rank_result(gather):
    """R-invariance: batch shards + in-place MAX all-reduce of the amax slots + gather == the single-rank GPU
    result BIT FOR BIT (shards start on even rows; the silent tail of clip 2 sits on the top_db floor set by the loud
    clip 1, which lives on the other rank), and within 1e-3 of the oracle."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, 29400 + os.getpid() % 500, q, gather)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    m, sd = build("uit_xxxs", 32)
    pcm = synth_pcm_i16(16, 16000, seed=13)
    single, _ = run(m, pcm)
    ref = Oracle(sd).forward(pcm)
    for r in range(2):
        assert res[r].shape == (16, 537)
        assert np.array_equal(res[r], single)
        assert np.abs(res[r] - ref["probs"]).max() < TOL


def _rccl_one_rank_worker(port, q):
    import os
    import torch.distributed as dist
    from uit_mobile_amd.sharded import ShardedUiT
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        m, _ = build("uit_xxxs", 32)
        x = torch.from_numpy(synth_pcm_i16(8, 16000, seed=13)).cuda()
        want = m(x).clone()
        st = m.frontend(x)
        slots = m.amax_tensor(x.device)
        before = slots.clone()
        dist.all_reduce(slots, op=dist.ReduceOp.MAX, async_op=True).wait()     # RCCL on the library-owned slots (foreign pointer)
        p = m.encode(st)
        out = torch.empty((8, 537), device="cuda")
        dist.all_gather_into_tensor(out, p)
        torch.cuda.synchronize()
        q.put((bool(torch.equal(before, slots)), bool(torch.equal(out, want)), bool(torch.equal(ShardedUiT(m)(x), want))))
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_accept_the_library_owned_amax_slots():
    """What bench.py does at N > 1, on a one-rank RCCL group (a 1-GPU box cannot host two RCCL ranks): the in-place MAX
    all-reduce runs on the device slots uitk_amax_ptr hands out -- memory torch's allocator does not own -- and the all-gather
    on the encoder's output."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(29300 + os.getpid() % 500, q))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=60)
    assert res == (True, True, True)


def _one_gpu_worker(rank, world, port, q, rows_per_rank, mode):
    """Several ranks sharing ONE GPU: gloo for the control plane (RCCL refuses two ranks on one device), the amax MAX all-reduce
    staged through the host, the outputs gathered by PeerGather's device-to-device copies into IPC-mapped buffers.
    mode: 'probs' (fp32 payload), 'f16' (float16 payload), 'topk' (device-side top-5 before the gather), 'pipelined' (two gathers in
    flight, every handle waited after the next gather was enqueued: the pattern the consumption fence exists for)."""
    import os
    import torch.distributed as dist
    from uit_mobile_amd.sharded import ShardedUiT
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, sd = build("uit_xxxs", 32, prec="f16l6")
        lo = rank * rows_per_rank
        kw = {"f16": dict(gather_dtype=torch.float16), "topk": dict(topk=5)}.get(mode, {})
        sh = ShardedUiT(m, gather="peer", **kw)
        outs = []
        if mode == "pipelined":
            batches = [synth_pcm_i16(world * rows_per_rank, 16000, seed=13 + i) for i in range(5)]
            prev = None
            for pcm in batches:                                # handle n is waited after gather n+1 has been enqueued
                h = sh(torch.from_numpy(pcm[lo:lo + rows_per_rank]).cuda(), async_gather=True)
                if prev is not None:
                    outs.append(prev.wait().clone().cpu().numpy())
                prev = h
            outs.append(prev.wait().clone().cpu().numpy())
        else:
            pcm = synth_pcm_i16(world * rows_per_rank, 16000, seed=13)
            for rep in range(3):                               # repeated: the gather buffers rotate, the amax slot sets alternate
                out = sh(torch.from_numpy(pcm[lo:lo + rows_per_rank]).cuda())
                torch.cuda.synchronize()
                outs.append(tuple(o.cpu().numpy() for o in out) if isinstance(out, tuple) else out.cpu().numpy())
        q.put((rank, outs))
        dist.barrier()                                         # keep the IPC-mapped buffers alive until every rank has read
    finally:
        dist.destroy_process_group()


def _run_ranks_on_one_gpu(world, rows_per_rank, mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() * 7 + world * 13 + rows_per_rank) % 500
    procs = [ctx.Process(target=_one_gpu_worker, args=(r, world, port, q, rows_per_rank, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    return res


@pytest.mark.parametrize("world,rows_per_rank", [(2, 8), (2, 6), (4, 4)])
def test_ranks_on_one_gpu_peer_gather_is_bitwise_the_single_rank_result(world, rows_per_rank):
    """The N>1 data path on a 1-GPU box: 2 or 4 processes, batch shards, MAX all-reduce of the amax slots, copy-engine gather
    through IPC-mapped peer buffers -> every rank holds the full [B, 537] output, bit for bit the single-rank result
    (R-invariance; shards start on even rows) and within 1e-3 of the oracle.  Parity mode (f16l6)."""
    res = _run_ranks_on_one_gpu(world, rows_per_rank, "probs")
    m, sd = build("uit_xxxs", 32, prec="f16l6")
    pcm = synth_pcm_i16(world * rows_per_rank, 16000, seed=13)
    single, _ = run(m, pcm)
    ref = Oracle(sd).forward(pcm)
    for r in range(world):
        for out in res[r]:
            assert out.shape == (world * rows_per_rank, 537)
            assert np.array_equal(out, single)
            assert np.abs(out - ref["probs"]).max() < TOL


@pytest.mark.parametrize("mode", ["f16", "topk", "pipelined"])
def test_ranks_on_one_gpu_reduced_payloads_and_pipelined_gathers(mode):
    """The payload cuts that de-risk the 8-GPU gather and the consumption fence of PeerGather, on two ranks
    sharing the GPU: float16 probabilities (1 074 B per clip instead of 2 148; |error| <= 2.5e-4), the device-side top-5 gathered
    instead of the probabilities (40 B per clip; equals torch.topk of the single-rank output), and five pipelined steps in which
    every handle is waited only after the next gather has been enqueued (results must be those of the single-rank run)."""
    world, rows = 2, 8
    res = _run_ranks_on_one_gpu(world, rows, mode)
    m, _ = build("uit_xxxs", 32, prec="f16l6")
    if mode == "pipelined":
        for i in range(5):
            single, _ = run(m, synth_pcm_i16(world * rows, 16000, seed=13 + i))
            for r in range(world):
                assert np.array_equal(res[r][i], single), (r, i)
        return
    single, _ = run(m, synth_pcm_i16(world * rows, 16000, seed=13))
    for r in range(world):
        for out in res[r]:
            if mode == "f16":
                assert out.dtype == np.float16 and out.shape == single.shape
                assert np.array_equal(out, single.astype(np.float16))
            else:
                v, i = out
                tv, ti = torch.from_numpy(single).topk(5)
                assert np.array_equal(v, tv.numpy()) and np.array_equal(i, ti.numpy())


def test_host_clip_streamer_matches_direct_calls():
    """8f row 1: pinned int16 staging + async H2D overlapped with compute gives the same outputs, in order."""
    from uit_mobile_amd.streaming import HostClipStreamer
    m, sd = build("uit_xxxs", 33)
    batches = [torch.from_numpy(synth_pcm_i16(n, 16000, seed=40 + i)) for i, n in enumerate((32, 32, 17, 32))]
    st = HostClipStreamer(m, 32, 16000)
    outs = [o.clone() for o in st.run(batches)]
    torch.cuda.synchronize()
    assert [o.shape[0] for o in outs] == [32, 32, 17, 32]
    orc = Oracle(sd)
    for b, o in zip(batches, outs):
        # each batch is one forward call (its own batch-global top_db floor): the oracle on the same int16 PCM
        assert np.abs(o.cpu().numpy() - orc.forward(b.numpy())["probs"]).max() < TOL
    # non-pinned float32 batches, no per-batch synchronisation by the consumer: the host copy into a pinned staging buffer
    # must wait for the H2D copy that still reads it (two batches earlier)
    fb = [b.float() / 32768.0 for b in batches] * 3
    st32 = HostClipStreamer(m, 32, 16000, dtype=torch.float32)
    outs32 = list(st32.run(fb))
    torch.cuda.synchronize()
    for i, o in enumerate(outs32):
        assert torch.equal(o, outs[i % 4])


def test_multi_producer_feeder_through_the_streamer_vs_oracle():
    """SURVEY 8f-1 (round 3): PinnedBatchFeeder -- producer threads filling pinned int16 batches with the collate's zero padding --
    in front of HostClipStreamer: outputs in batch order, equal to direct calls bit for bit and within 1e-3 of the oracle, including
    a ragged last batch and clips shorter than the batch length."""
    from uit_mobile_amd.streaming import HostClipStreamer, PinnedBatchFeeder
    m, sd = build("uit_xxxs", 21, prec="f16l6")
    B, T, nb = 12, 16000, 5
    clips = [synth_pcm_i16(B, T, seed=70 + i) for i in range(nb)]
    lens = np.random.default_rng(2).integers(12000, T + 1, size=(nb, B))
    lens[:, 0] = T

    def fill(i, out):
        for r in range(B):
            PinnedBatchFeeder.collate(out[r], clips[i][r, :lens[i, r]])
    outs = [o.cpu().numpy() for o in HostClipStreamer(m, B, T).run(PinnedBatchFeeder(fill, nb, B, T, producers=3))]
    assert len(outs) == nb
    o = Oracle(sd)
    for i in range(nb):
        padded = clips[i].copy()
        for r in range(B):
            padded[r, lens[i, r]:] = 0
        direct, _ = run(m, padded)
        assert np.array_equal(outs[i], direct)
        if i in (0, nb - 1):
            assert np.abs(outs[i] - o.forward(padded)["probs"]).max() < TOL


def test_feeder_buffers_are_not_recycled_under_a_running_copy():
    """the feeder hands a pinned buffer back to its producers once three later batches have been requested -- safe only
    because staging batch k waits on the HOST for the H2D copy of batch k-2 (HostClipStreamer._stage).  Distinct batches, the ring
    pre-filled before the consumer starts, large batches (a copy takes far longer than a loop iteration) and NO synchronisation per
    output: without the wait the producers overwrite buffers whose DMA is still queued and clips of later batches show up in
    earlier results."""
    from uit_mobile_amd.streaming import HostClipStreamer, PinnedBatchFeeder
    m, sd = build("uit_xxxs", 21)
    B, T, nb = 2048, 16000, 12
    base = synth_pcm_i16(B, T, seed=5)

    def fill(i, out):                                       # batch i: the base batch rolled by i clips -> every batch is distinct
        np.copyto(out, np.roll(base, i, axis=0))
    feeder = PinnedBatchFeeder(fill, nb, B, T, producers=4, depth=4)
    import time
    time.sleep(0.5)                                         # the ring is full before the consumer takes its first batch
    with torch.no_grad():
        outs = [o for o in HostClipStreamer(m, B, T).run(feeder)]      # device tensors: nothing waits for the device in between
        torch.cuda.synchronize()
        want, _ = run(m, base)
    for i, o in enumerate(outs):
        got = o.cpu().numpy()
        # row r of batch i is clip (r - i) mod B; clip pairs keep their wave when i is even: compare exactly there, closely otherwise
        ref = np.roll(want, i, axis=0)
        assert np.abs(got - ref).max() < (1e-6 if i % 2 == 0 else 1e-4), i


def test_clip_store_through_feeder_and_streamer_vs_oracle(tmp_path):
    """a directory of clips of different lengths and dtypes (int16 as stored, float32 as the reference's loader returns
    them) -> ClipStore -> pinned batches -> streamer: every clip's probabilities within 1e-3 of the oracle run on the padded batch,
    the short last batch included."""
    from uit_mobile_amd.dataset import ClipStore, to_int16
    from uit_mobile_amd.streaming import HostClipStreamer
    m, sd = build("uit_xxxs", 22, prec="f16l6")
    B, T, n = 4, 16000, 10
    src = synth_pcm_i16(n, T, seed=90)
    lens = np.random.default_rng(4).integers(9000, T + 1, n)
    lens[0] = T
    for i in range(n):
        c = src[i, :lens[i]]
        np.save(tmp_path / f"c{i:02d}.npy", c if i % 2 else (c / 32768.0).astype(np.float32))
    store = ClipStore.from_npy_dir(tmp_path)
    outs = [o.cpu().numpy() for o in HostClipStreamer(m, B, T).run(store.feeder(B, T, producers=2))]
    o = Oracle(sd)
    for (first, count, names), probs in zip(store.batch_index(B), outs):
        padded = np.zeros((B, T), np.int16)
        for j in range(count):
            padded[j, :lens[first + j]] = src[first + j, :lens[first + j]]
        assert list(store.lengths(first)) == list(lens[first:first + count]) and names[0] == f"c{first:02d}"
        assert np.abs(probs[:count] - o.forward(padded)["probs"][:count]).max() < TOL


@pytest.mark.parametrize("T", [16001, 15999, 4099])
def test_odd_clip_lengths_vs_oracle(T):
    """Odd sample counts: rows are no longer pair-aligned (the paired-sample fast path must switch off), frame counts
    and reflect padding at the right edge change."""
    m, sd = build("uit_xxxs", 34)
    pcm = synth_pcm_i16(5, T, seed=T)
    ref = Oracle(sd).forward(pcm)
    p, l = run(m, pcm)
    assert np.abs(l - ref["logits"]).max() < TOL and np.abs(p - ref["probs"]).max() < TOL
    p32, _ = run(m, pcm.astype(np.float32) / 32768.0)
    assert np.abs(p32 - ref["probs"]).max() < TOL


def test_non_default_hop_size_vs_oracle():
    """UITBase kwargs reach the front-end (models/uit.py:289): hop 128 -> 126 frames for 1 s -> two crops."""
    sd = make_state_dict("uit_xxxs", 35)
    m = models.uit_xxxs(outputdim=537, target_length=102, hop_size=128)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    pcm = synth_pcm_i16(6, 16000, seed=77)
    o = Oracle(sd, hop=128)
    assert m.num_crops(16000) == o.num_crops(16000) == 2
    ref = o.forward(pcm)
    p, l = run(m, pcm)
    assert np.abs(l - ref["logits"]).max() < TOL and np.abs(p - ref["probs"]).max() < TOL
    mel = m.melspec_db(torch.from_numpy(pcm).cuda()).cpu().numpy()
    assert_mel_close(mel, ref["mel_db"], scale=2.0)


@pytest.mark.parametrize("kind", ["dense", "wide", "shifted"])
def test_filterbank_from_state_dict_is_honoured(kind):
    """front_end.0.mel_scale.fb is a state_dict buffer: whatever it holds must be what the front-end multiplies by.
    HTK triangles take the on-chip banded path (52 steps per 4-mel group); a dense matrix or bands wider than that walk
    the generic path (up to 260 steps), a shifted filterbank moves every band."""
    sd = dict(make_state_dict("uit_xxxs", 36))
    fb = np.array(sd["front_end.0.mel_scale.fb"], dtype=np.float32)          # [257, 64]
    rng = np.random.default_rng(5)
    if kind == "dense":
        fb = (fb + 0.002 * rng.random(fb.shape)).astype(np.float32)          # every (bin, mel) pair non-zero
    elif kind == "wide":
        wide = np.zeros_like(fb)
        for mcol in range(64):                                               # box filters 80 bins wide
            lo = min(3 * mcol, 257 - 80)
            wide[lo:lo + 80, mcol] = 1.0 / 80
        fb = wide
    else:
        fb = np.roll(fb, 5, axis=0)                                           # bands start 5 bins later, last band wraps to bin 0
    sd["front_end.0.mel_scale.fb"] = fb
    m = models.uit_xxxs(outputdim=537, target_length=102)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    pcm = synth_pcm_i16(5, 16000, seed=91)
    ref = Oracle(sd).forward(pcm)
    mel = m.melspec_db(torch.from_numpy(pcm).cuda()).cpu().numpy()
    assert_mel_close(mel, ref["mel_db"], scale=2.0)
    p, l = run(m, pcm)
    assert np.abs(l - ref["logits"]).max() < TOL and np.abs(p - ref["probs"]).max() < TOL


def test_two_stage_api_interleaved_replicas_match_forward():
    """model.frontend / model.encode (uitk_frontend / uitk_encode) with another replica's front-end enqueued in between
    (what bench.py does to hide the amax all-reduce) give bit-identical results to forward(); reserve() pre-allocates."""
    ma, sd = build("uit_xxxs", 37)
    mb, _ = build("uit_xxxs", 37)
    xa = torch.from_numpy(synth_pcm_i16(9, 16000, seed=1)).cuda()
    xb = torch.from_numpy(synth_pcm_i16(5, 16384, seed=2)).cuda()          # other length: two crops
    ma.reserve(9, 16000); mb.reserve(5, 16384)
    want_a, want_b = ma(xa).clone(), mb(xb).clone()
    for _ in range(3):                                                     # repeated: the amax slot sets alternate per call
        sa = ma.frontend(xa)
        sb = mb.frontend(xb)
        pa = ma.encode(sa)
        sa2 = ma.frontend(xa)
        pb, lb = mb.encode(sb, return_logits=True)
        pa2 = ma.encode(sa2)
        assert torch.equal(pa, want_a) and torch.equal(pa2, want_a) and torch.equal(pb, want_b)
        assert lb.shape == (5, 2, 537)
    with pytest.raises(RuntimeError):
        ma.reserve(4, 16000, "cpu")
    # uitk_encode must follow a uitk_frontend of the SAME (B, T): anything else would read stale operands
    st = ma.frontend(xa)
    with pytest.raises(RuntimeError):
        ma.encode((st[0] - 1, st[1], st[2]))
    with pytest.raises(RuntimeError):
        ma.encode((st[0], 16160, st[2]))
    assert torch.equal(ma.encode(st), want_a)
    ma.melspec_db(xa)                                                      # re-uses the amax slots: invalidates the pending state
    with pytest.raises(RuntimeError):
        ma.encode(st)


def test_device_topk_matches_host_topk():
    """uitk_topk (inference.py:57 `output.topk(k)` on the device): values bit-equal, indices equal; ties -> lower index."""
    from uit_mobile_amd.postprocess import topk
    g = torch.Generator().manual_seed(3)
    for B, C, k in ((1, 537, 3), (37, 537, 5), (8, 527, 10), (5, 64, 64), (3, 1000, 7), (4097, 537, 1)):
        p = torch.rand(B, C, generator=g)
        vals, idx = topk(p.cuda(), k)
        wv, wi = p.topk(k)
        assert vals.dtype == torch.float32 and idx.dtype == torch.int64
        assert torch.equal(vals.cpu(), wv) and torch.equal(idx.cpu(), wi)
    p = torch.zeros(2, 537)
    p[0, [400, 7, 300]] = 0.5                                  # three-way tie
    p[1, :] = 0.25
    vals, idx = topk(p.cuda(), 4)
    assert idx.cpu().tolist() == [[7, 300, 400, 0], [0, 1, 2, 3]]
    assert vals.cpu().tolist() == [[0.5, 0.5, 0.5, 0.0], [0.25] * 4]
    v1, i1 = topk(torch.rand(537, generator=g).cuda(), 3)      # 1-D row, as inference.py passes it
    assert v1.shape == (3,) and i1.shape == (3,)
    with pytest.raises(ValueError):
        topk(p.cuda(), 538)                                    # k > C: torch raises too
    with pytest.raises(RuntimeError):
        topk(p, 3)                                             # host tensor: no CPU path


def test_device_kws_transform_matches_reference_loop():
    """uitk_kws_transform vs the loop restatement of _output_transform_xiaoai (evaluate.py:212-229): bit-exact."""
    from conftest import kws_loop_restatement
    from uit_mobile_amd.evaluate import output_transform_kws
    g = torch.Generator().manual_seed(0)
    for B, C in ((64, 537), (3, 527), (1000, 537), (5, 600)):
        y_pred = torch.rand(B, C, generator=g)
        y_pred[::3, 527:] *= 0.1                               # rows without a firing keyword
        y_pred[1, :527] = 0.0                                  # all-equal fillers
        if B > 4:
            y_pred[4, [10, 200]] = 0.999                       # tied maximal fillers: both kept, the first one cleared
        y = torch.zeros(B, C)
        y[torch.arange(B), torch.randint(0, C, (B,), generator=g)] = 1.0
        for thr in (0.2, 0.5):
            a, ya = output_transform_kws(y_pred.cuda(), y.cuda(), thr)
            b, yb = kws_loop_restatement(y_pred, y, thr)
            assert torch.equal(a.cpu(), b) and torch.equal(ya.cpu(), yb)


@pytest.mark.parametrize("gather,launcher,ranks", [("rccl", "torchrun", 2), ("peer", "torchrun", 2), ("auto", "self", 2), ("auto", "self", 8)])
def test_bench_two_rank_pipeline_runs_on_one_gpu(gather, launcher, ranks):
    """bench.py's N > 1 code path (shards, pipelined front-end / amax all-reduce / encoder / asynchronous gathers, barrier
    + max-over-ranks timing) exercised end to end with two ranks on one GPU over gloo -- RCCL itself needs one GPU per rank;
    what is checked is that the pipeline runs, every gather completes, rank 0 prints one well-formed JSON line and the sampled rows
    of the last timed step match the oracle.  launcher 'self' = `python bench.py --gpus N ...` as the driver types it: the script
    starts its own ranks as a child torch.distributed.run (round 4; it used to exit with "launch with torch.distributed.run").
    The 8-rank case is the driver's N = 8 command line with every rank on the one GPU: eight handles, the 8-way peer fan-out with its
    self-test, eight-way amax reduction -- everything but RCCL itself."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    per = 512 if ranks == 2 else 256
    args = [str(root / "bench.py"), "--gpus", str(ranks), "--steps", "6", "--warmup", "2", "--batch-per-gpu", str(per), "--backend", "gloo",
            "--all-ranks-on-device0", "--gather", gather]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(29700 + os.getpid() % 200)] + args
    else:
        cmd = [sys.executable] + args
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(root), env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["steps"] == 6 and d["config"]["global_batch"] == ranks * per and d["value"] > 0
    assert d["scaling"] == "weak" and "cpu_baseline" not in d and d["rccl_ranks"] == ranks
    assert d["gather"] == ("rccl" if gather == "rccl" else "peer") and ("peer" in d["config"]["parallelism"]) == (gather != "rccl")
    assert d["precision"] == "auto" and d["effective_precision"] == "f16c" and d["max_abs_dprob"] is not None and d["max_abs_dprob"] < TOL, d.get("parity_check")


def test_bench_watchdog_falls_back_to_rccl_when_the_peer_self_test_stalls():
    """Round 5: `python bench.py --gpus 8` with every rank on the one GPU and a stall INJECTED into rank 3's
    peer-gather set-up (the stand-in for a hung hipIpcOpenMemHandle / peer copy on a real 8-GPU node): the workers' own watchdogs
    fire when the set-up budget runs out, the supervisor kills what is left of the attempt and starts eight FRESH workers with
    --gather rccl, whose JSON line comes out -- with the supervisor's record of both attempts -- inside the budget."""
    import json
    import subprocess
    import sys
    import time
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    # (uit_xxxs in precision f16: no calibration, a 4-block pack -- what is tested here is the watchdog, and eight ranks that share one
    # GPU and one host take long enough over their set-up as it is)
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "2", "--batch-per-gpu", "256", "--backend", "gloo",
           "--model", "uit_xxxs", "--precision", "f16", "--all-ranks-on-device0", "--gather", "auto", "--inject-stall", "peer_gather:3",
           "--setup-timeout", "100", "--run-timeout", "120"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(root), env=env)
    dt = time.time() - t0
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    print(f"watchdog run: {dt:.0f} s, rc {out.returncode}; stderr tail: {out.stderr[-600:]}")
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    att = d["watchdog"]["attempts"]
    assert [a["gather"] for a in att] == ["auto", "rccl"] and att[0]["error"] and att[1]["error"] is None
    print("attempts:", att)
    assert att[0]["last_phase"]["3"].startswith("peer_gather") and att[0]["seconds"] < 100 + 25
    assert d["gather"] == "rccl" and d["n_gpus"] == 8 and d["value"] > 0 and d["max_abs_dprob"] < TOL
    assert len(d["rank_devices"]) == 8 and all(r["effective_precision"] == "f16" for r in d["rank_devices"])


def test_token_ladder_reaches_f16ax_on_heavy_tailed_checkpoints(golden):
    """Round 5: the checkpoint family with the STRUCTURE of trained weights (synth.make_structured_trained_like_state_dict: heavy tails, a
    low-rank component, uneven rows) broke pooling 'token' with full-width attention: six of 24 checkpoints over 1e-3 in every fp16 mode
    (seed 223: 3.3e-3 in 'f16a', 2.2e-3 in 'f16x2'; profiles/round5/auto_sweep_token_structured.log) -- the fp16 rounding of the KEYS, then
    values and queries, of all tokens (tools/err_attrib_token.py).  Now 'f16a' carries q, k, v of every tile as hi + lo, and an 'f16a'
    handle over the budget is rebuilt as 'f16ax' (the same on split weights).  (a) seed 223: 'auto' ends in 'f16ax', inside the budget and
    the gate on the wide held-out set; an explicit 'f16a' warns and stays.  (b) BNeckAttention, uit_xxs seed 234: 'f16ax' from uitk_create's
    image alone (no calibration) -- the case in which the first build of this kernel was wrong (profiles/round5/f16ax_bisect.log; round 6
    found why: an asm v_max3_f32 whose output register an MFMA in flight still had to write -- profiles/round6/f16ax_race.log)."""
    import warnings
    from uit_mobile_amd.models.uit import UitkParityWarning
    import uit_mobile_amd.models.uit as U
    from uit_mobile_amd.synth import make_structured_trained_like_state_dict
    x, names = held_out_inputs(golden, with_names=True)
    xd = torch.from_numpy(x).cuda()

    def run(factory, sd, act, prec):
        m = getattr(models, factory)(outputdim=537, target_length=102, pooling="token", precision=prec)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        m = m.cuda().eval()
        with torch.no_grad():
            lg = m(xd, return_logits=True)[1].cpu().numpy()
        return m, lg

    # (a)
    sd = make_structured_trained_like_state_dict("audio_transformer_h128_d4_m3", 223, attention="full")
    ref = Oracle(sd, act="gelu", pooling="token").forward(x)["logits"]
    with pytest.warns(UitkParityWarning, match="f16ax"):
        m, lg = run("audio_transformer_h128_d4_m3", sd, "gelu", "auto")
    d = np.abs(lg.reshape(ref.shape) - ref).reshape(len(x), -1).max(1)
    rep = m.calibration_report()
    print(f"structured seed 223, full attention + GELU, pooling token: auto -> {m.effective_precision()}: worst max|dlogit| {d.max():.2e} ({names[int(d.argmax())]}); report {rep}")
    assert m.effective_precision() == "f16ax" and rep["escalated"] and rep["within_budget"] and rep["self_check"][0] <= rep["budget"]
    assert d.max() < 7e-4
    with pytest.warns(UitkParityWarning, match="does not escalate"):
        ma, lga = run("audio_transformer_h128_d4_m3", sd, "gelu", "f16a")
    da = np.abs(lga.reshape(ref.shape) - ref).max()
    print(f"   explicit f16a: {da:.2e} (over the budget, reported: {ma.calibration_report()['within_budget']})")
    assert ma.effective_precision().startswith("f16a") and ma.calibration_report()["within_budget"] is False and da > d.max()
    # (b)
    sd = make_structured_trained_like_state_dict("uit_xxs", 234)
    ref = Oracle(sd, act="relu", pooling="token").forward(x)["logits"]
    keep = U._CALIBRATING
    U._CALIBRATING = tuple(p for p in keep if p != "f16ax")              # uitk_create's image: no statistics, mu = 0 in the centring table
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m, lg = run("uit_xxs", sd, "relu", "f16ax")
    finally:
        U._CALIBRATING = keep
    d = np.abs(lg.reshape(ref.shape) - ref).reshape(len(x), -1).max(1)
    print(f"structured seed 234, uit_xxs, pooling token, f16ax without calibration: worst max|dlogit| {d.max():.2e} ({names[int(d.argmax())]})")
    assert np.isfinite(lg).all() and d.max() < 3e-4


@pytest.mark.parametrize("pooling", ["mean", "token"])
def test_degenerate_whole_batches(pooling):
    """Batches in which EVERY clip is degenerate -- nothing in them sets a top_db floor above the amin clamp, or everything clips: all
    digital silence (float32 and int16; every mel bin at 10 log10(1e-10) = -100 dB, the batch amax included), one lone silent clip
    (B = 1), full-scale DC (+1.0: only bin 0 and the window's leakage carry power), full-scale Nyquist alternation, int16 rails
    (-32768 everywhere).  Against the C oracle at trained-like magnitudes; finite everywhere."""
    m, sd = build_trained("uit_xxs", 209, pooling=pooling)
    o = Oracle(sd, pooling=pooling)
    T_ = 16000
    alt = np.where(np.arange(T_) % 2 == 0, 1.0, -1.0).astype(np.float32)
    cases = {"silence f32 x4": np.zeros((4, T_), np.float32), "silence i16 x3": np.zeros((3, T_), np.int16), "silence B=1": np.zeros((1, T_), np.float32),
             "DC +1.0 x2": np.ones((2, T_), np.float32), "Nyquist +-1.0 x2": np.stack([alt, -alt]), "int16 rail x2": np.full((2, T_), -32768, np.int16)}
    for what, x in cases.items():
        p, l = run(m, x)
        r = o.forward(x)
        dl, dp = np.abs(l.reshape(r["logits"].shape) - r["logits"]).max(), np.abs(p - r["probs"]).max()
        print(f"pooling {pooling}, {what}: |logit|max {np.abs(r['logits']).max():.1f} max|dlogit| {dl:.2e} max|dprob| {dp:.2e}")
        assert np.isfinite(l).all() and np.isfinite(p).all() and dl < 1e-3 and dp < 2.5e-4, what


@pytest.mark.parametrize("factory,att,act,pooling", [("uit_xxs", "bneck", "relu", "mean"), ("uit_xxs", "bneck", "relu", "token"), ("uit_xxxs", "bneck", "relu", "dm"),
                                                      ("audio_transformer_h128_d4_m3", "full", "gelu", "mean"), ("audio_transformer_h128_d4_m3", "full", "gelu", "token")])
def test_f32_rung_matches_the_oracle(golden, factory, att, act, pooling):
    """Precision 'f32' (round 5): the plain fp32 forward of the library (csrc/calibrate.hip + a head kernel) behind the product front-end --
    the last rung of the parity ladder.  Against the C oracle at trained-like magnitudes through every shape of the path: 1 s clips (wide
    held-out set), 0.5 s clips (masked time patches), 4-crop clips (crop reduction), int16 input; far inside the tolerance (the rung reads
    the front-end's fp32 dB frames, not its 16-bit operands: what is left are the two fp32 FFTs -- the bin-edge tone pairs)."""
    x, names = held_out_inputs(golden, with_names=True)
    sd = make_trained_like_state_dict(factory, 210, attention=att)
    m = getattr(models, factory)(outputdim=537, target_length=102, pooling=pooling, precision="f32")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    o = Oracle(sd, act=act, pooling=pooling)
    # (134 clips = 3216+ tokens in the first case: the register-tiled linear of calibrate.hip; the small cases run its 16 x 16 one)
    for what, xi in (("1 s", np.concatenate([x, x[::-1]])), ("0.5 s", np.ascontiguousarray(x[:24, 4000:12000])), ("4 s (4 crops)", np.ascontiguousarray(np.tile(x[:6], (1, 4))))):
        r = o.forward(xi)
        with torch.no_grad():
            if pooling == "dm":
                p = m(torch.from_numpy(xi).cuda()).cpu().numpy()
                dl = 0.0
            else:
                p, l = m(torch.from_numpy(xi).cuda(), return_logits=True)
                p, dl = p.cpu().numpy(), float(np.abs(l.cpu().numpy().reshape(r["logits"].shape) - r["logits"]).max())
        dp = float(np.abs(p - r["probs"]).max())
        print(f"f32 {factory} pooling {pooling}, {what}: max|dlogit| {dl:.2e} max|dprob| {dp:.2e}")
        assert dl < 3e-4 and dp < 1e-4, what
    assert m.effective_precision() == "f32" and m.calibration_report() is None
    sc = m.self_check()
    assert sc is not None and sc[0] < 5e-5, sc


def test_auto_ends_in_f32_on_an_attention_sink_checkpoint(golden):
    """A third checkpoint family (synth.make_sink_trained_like_state_dict): an attention sink in every block and head -- six outlier keys
    every query looks for, scores of tens of units.  A sink key's fp16 rounding error q . dk is proportional to the score it produces and
    centring does not shrink an outlier (tools/err_attrib_sink.py: the activation floor of uit_xs goes 2.4e-4 -> 1.0e-3 at gain 4): no fp16
    mode of the library is inside the budget (tools/auto_sweep.py ... sink4: every checkpoint flagged).  The ladder goes on: 'auto' ends in
    'f32' -- slow, warned about, and right: against an fp64 evaluation of the SAME mel spectrogram the rung is at 1e-5, closer than the
    fp32 oracle is to fp64; what is left against the oracle is the two fp32 FFTs' round-off floors, amplified by this checkpoint
    (tools/fe_split.py).  With fp32_fallback=False 'auto' stays on the best fp16 build and says so."""
    import sys
    from uit_mobile_amd.models.uit import UitkParityWarning
    from uit_mobile_amd.synth import DEPTH, make_sink_trained_like_state_dict
    sys.path[:0] = [str(ROOT / "tools")]
    import err_attrib_centring as E
    x = held_out_inputs(golden)[:32]
    xd = torch.from_numpy(x).cuda()
    sd = make_sink_trained_like_state_dict("uit_xxs", 216, gain=4.0)
    tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    ref = Oracle(sd).forward(x)["logits"]

    def run_(**kw):
        m = models.uit_xxs(outputdim=537, target_length=102, **kw)
        m.load_state_dict(tsd, strict=True)
        m = m.cuda().eval()
        with torch.no_grad():
            l = m(xd, return_logits=True)[1].cpu().numpy().reshape(ref.shape)
        return m, l, float(np.abs(l - ref).max())

    with pytest.warns(UitkParityWarning, match="'f32'"):
        m, l, e = run_()
    rep = m.calibration_report()
    # the fp64 evaluation of the mel spectrogram the rung itself read
    with torch.no_grad():
        mel = m.melspec_db(xd).cpu().numpy().astype(np.float64)
    sc = np.asarray(sd["init_bn.1.weight"], np.float64) / np.sqrt(np.asarray(sd["init_bn.1.running_var"], np.float64) + 1e-5)
    mel_bn = torch.from_numpy((mel - np.asarray(sd["init_bn.1.running_mean"], np.float64)[None, :, None]) * sc[None, :, None]
                              + np.asarray(sd["init_bn.1.bias"], np.float64)[None, :, None])
    truth = E.forward(sd, mel_bn, DEPTH["uit_xxs"], "relu", "bneck", "mean", ()).numpy()
    e64 = float(np.abs(l.reshape(truth.shape) - truth).max())
    print(f"sink checkpoint (uit_xxs seed 216, gain 4): auto -> {m.effective_precision()}, ladder {rep['ladder']}: {e64:.2e} against fp64 on its own mel, "
          f"{e:.2e} against the C oracle (front-end difference); library: f16c {rep['max_calibrated']:.2e}, self check of the rung {rep['self_check'][0]:.1e}")
    assert m.effective_precision() == "f32" and rep["escalated"] and rep["within_budget"] and rep["max_calibrated"] > 1e-3
    assert e64 < 2e-5 and e < 1.5e-3
    with pytest.warns(UitkParityWarning, match="fp32_fallback=False"):
        m2, l2, e2 = run_(fp32_fallback=False)
    print(f"   fp32_fallback=False: stays on {m2.effective_precision()}: {e2:.2e} against the C oracle")
    assert m2.effective_precision() != "f32" and m2.calibration_report()["within_budget"] is False and e2 > e


@pytest.mark.parametrize("factory,att,precision", [("uit_xxs", "bneck", "f16ax"), ("uit_xxs", "bneck", "f16a"), ("audio_transformer_h128_d4_m3", "full", "f16ax"),
                                                      ("uit_xxs", "bneck", "f32")])
def test_token_ladder_rungs_are_bit_stable_under_load(factory, att, precision):
    """The rungs added late in round 5 -- pooling 'token' in 'f16a' / 'f16ax', and 'f32' -- twenty times on 1024 different clips, with a
    SECOND model running the default kernels on another stream at the same time (different clocks, different contention in every repeat):
    every output bit identical to the first run, finite, and the whole batch at the oracle.  The first build of the BNeckAttention
    'f16ax' kernel was wrong in a way that depended on timing between the waves of a workgroup (profiles/round5/f16ax_bisect.log; cause,
    round 6: profiles/round6/f16ax_race.log -- the build check test_no_mfma_hazard_that_the_compiler_cannot_see is its static guard): a
    result that is merely right on a quiet GPU would not be enough."""
    from uit_mobile_amd.synth import make_structured_trained_like_state_dict
    import uit_mobile_amd.models.uit as U
    sd = make_structured_trained_like_state_dict(factory, 234, attention=att)
    pooling = "mean" if precision == "f32" else "token"
    act = "relu" if att == "bneck" else "gelu"
    keep = U._CALIBRATING
    U._CALIBRATING = tuple(p for p in keep if p != "f16ax")              # (f16ax on uitk_create's image: the case that was wrong)
    try:
        m = getattr(models, factory)(outputdim=537, target_length=102, pooling=pooling, precision=precision)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        m = m.cuda().eval()
        pcm = synth_pcm_i16(1024, 16000, seed=97)
        x = torch.from_numpy(pcm).cuda()
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m.prepare("cuda", sample=x)
    finally:
        U._CALIBRATING = keep
    noise = models.uit_xs(outputdim=537, target_length=102, precision="f16")
    noise.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xs", 3).items()})
    noise = noise.cuda().eval()
    xn = torch.from_numpy(synth_pcm_i16(4096, 16000, seed=5)).cuda()
    side = torch.cuda.Stream()
    with torch.no_grad():
        noise(xn)
        first = m(x, return_logits=True)[1].clone()
        torch.cuda.synchronize()
        differing = 0
        for rep in range(20):
            with torch.cuda.stream(side):
                for _ in range(1 + rep % 3):
                    noise(xn)
            differing += int(not torch.equal(first.view(torch.int32), m(x, return_logits=True)[1].view(torch.int32)))      # (bit patterns: NaN == NaN)
            torch.cuda.synchronize()
    ref = Oracle(sd, act=act, pooling=pooling).forward(pcm)["logits"]           # (the whole batch: the top_db floor is batch-global)
    d = float(np.abs(first.cpu().numpy().reshape(ref.shape) - ref).max())
    print(f"{factory} {precision}: 20 repeats beside a second model on another stream: {differing} differ; 1024 clips vs oracle {d:.2e}")
    assert differing == 0 and bool(torch.isfinite(first).all()) and d < 8e-4


@pytest.mark.gpu
def test_fuzz_over_instantiations_against_the_oracle():
    """Round 6: random combinations of factory x pooling x precision x clip length (short / full / odd / multi-crop) x batch size x
    input dtype x workgroup shape against the C oracle on trained-like weights (tools/fuzz_instantiations.py; 2 000 cases on record in
    profiles/round6/fuzz_instantiations.log).  After the hazard clean-up of the encoder's inline asm every instantiation's ISA changed;
    the tests above walk the families one at a time, this walks combinations nobody wrote down.  Fixed seed, time-boxed."""
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
    import fuzz_instantiations
    rows, bad = fuzz_instantiations.run(cases=100, seed=6, budget_s=90, verbose=False)
    print(f"{len(rows)} random cases; " + ("all inside their gates" if not bad else f"{len(bad)} outside"))
    assert len(rows) >= 40 and not bad, bad[:5]


@pytest.mark.gpu
def test_a_short_clip_deployment_is_judged_on_short_clips(golden):
    """Round 6 (option calib_short_check): prepare() for clips of fewer than six time patches switches the half-length leg of the held-out
    check on -- the library then scores its 48 held-out clips at 0.5 s too (12 of 24 tokens, masked attention) and every maximum it
    reports, the budget decision included, is over both lengths; a model prepared for 1 s clips keeps round 5's check (it must not be sent
    up the ladder for what it would do on clips it never sees).  Estimate with the leg >= estimate without (a maximum over a superset),
    and the short clips themselves inside the gate against the C oracle."""
    sd = make_trained_like_state_dict("uit_xs", 207)
    tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}

    def make():
        m = models.uit_xs(outputdim=537, target_length=102)
        m.load_state_dict(tsd, strict=True)
        return m.cuda().eval()

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        full = make().prepare("cuda", (8, 16000))
        short = make().prepare("cuda", (8, 8000))
    assert "calib_short_check" not in full._options and short._options.get("calib_short_check") == 1
    sc_full, sc_short = full.self_check(), short.self_check()
    rep = short.calibration_report()
    print(f"uit_xs seed 207: held-out estimate, full clips only {sc_full[0]:.2e}; with the half-length leg {sc_short[0]:.2e} -> {short.effective_precision()}; report {rep}")
    assert sc_short[0] >= sc_full[0] * 0.999 or short.effective_precision() != full.effective_precision()
    x = held_out_inputs(golden)[:24, 4000:12000]
    ref = Oracle(sd).forward(np.ascontiguousarray(x))["logits"]
    with torch.no_grad():
        lg = short(torch.from_numpy(np.ascontiguousarray(x)).cuda(), return_logits=True)[1].cpu().numpy()
    d = float(np.abs(lg.reshape(ref.shape) - ref).max())
    print(f"   24 clips of 0.5 s against the oracle: {d:.2e}")
    assert d < 8e-4
