"""Pins the CPU oracle (oracle/uit_oracle.c) against the fixtures produced by the reference's own
models/uit.py on torch CPU (tests/golden/make_golden.py).  CPU only.

Tolerances: the oracle is fp32 like the reference; differences are summation-order noise.
  mel dB   : graded by depth below the clip's peak, 5e-4 .. 0.25 dB (fp32 FFT round-off floor, see
             conftest.assert_mel_close)
  tokens   : 2e-5 abs per stage (5e-3 for the pure-tone clip)
  logits   : 1e-4 abs, probs 5e-5 abs   (the product gate vs the oracle is 1e-3)
"""
import numpy as np
import pytest

from oracle.oracle import Oracle
from uit_mobile_amd.synth import make_state_dict
from conftest import assert_mel_close

MODELS = ["uit_xxxs", "uit_xxs", "uit_xs"]


def _oracle(name, seed, scale=1.0, **kw):
    return Oracle(make_state_dict(name, int(seed), float(scale), **{k: v for k, v in kw.items() if k == "outputdim"}),
                  **{k: v for k, v in kw.items() if k != "outputdim"})


@pytest.mark.parametrize("name", MODELS)
def test_synth_1s_all_stages(golden, name):
    g = golden(f"synth1s_{name}")
    o = _oracle(name, g["seed"])
    r = o.forward(g["pcm"], want_stages=True)
    assert r["mel_db"].shape == g["mel_db"].shape == (8, 64, 101)
    assert_mel_close(r["mel_db"], g["mel_db"])
    # the batch contains digital silence: the floor must be batch-global (amax - 120), SURVEY 8a row 3
    assert g["mel_db"].min() == pytest.approx(g["mel_db"].max() - 120.0, abs=1e-4)
    assert r["mel_db"].min() == pytest.approx(r["mel_db"].max() - 120.0, abs=1e-4)
    ns = g["stages"].shape[0]
    assert np.abs(r["stages"][0] - g["stages"][0]).max() < 2e-5          # noise clip
    # clip 1 is a full-scale pure tone: nearly all of its bins sit on the FFT round-off floor (see
    # assert_mel_close), which BN + patch-embed turn into ~1e-3 token differences; logits still agree to 1e-4
    assert np.abs(r["stages"][1] - g["stages"][1]).max() < 5e-3
    assert np.abs(r["logits"] - g["logits"]).max() < 1e-4
    assert np.abs(r["probs"] - g["probs"]).max() < 5e-5


def test_int16_and_float_inputs_agree(golden):
    g = golden("synth1s_uit_xxxs")
    o = _oracle("uit_xxxs", g["seed"])
    a = o.forward(g["pcm"][:3])["probs"]
    b = o.forward(g["pcm"][:3].astype(np.float32) / 32768.0)["probs"]
    assert np.array_equal(a, b)


def test_scaled_weights(golden):
    g = golden("synth1s_uit_xs_scale3")
    r = _oracle("uit_xs", g["seed"], g["scale"]).forward(g["pcm"])
    assert np.abs(r["logits"] - g["logits"]).max() < 3e-4
    assert np.abs(r["probs"] - g["probs"]).max() < 1e-4   # x3 weights amplify the pure-tone clip (index 1)


@pytest.mark.parametrize("name", ["uit_xs", "uit_xxxs"])
def test_reference_sample_clips(golden, name):
    """BASELINE config #1 inputs: water_000 (16384 samples -> 103 frames -> 2 crops) and the GSC clip."""
    g = golden(f"samples_{name}")
    o = _oracle(name, g["seed"])
    for key, ncrops in (("water_000", 2), ("water_003", 2), ("85b877b5_nohash_0", 1)):
        pcm = g[f"{key}_pcm"][None]
        assert o.num_crops(pcm.shape[1]) == ncrops
        r = o.forward(pcm)
        assert np.abs(r["logits"][0] - g[f"{key}_logits"]).max() < 1e-4
        assert np.abs(r["probs"][0] - g[f"{key}_probs"]).max() < 5e-5
    # evaluate.py:56-64 pad rule: 16000 -> 16160 samples (102 frames, single crop)
    assert int(g["gsc_pad_diff"]) == 160
    r = o.forward(g["gsc_padded_pcm"][None])
    assert o.num_crops(16160) == 1
    assert np.abs(r["probs"][0] - g["gsc_padded_probs"]).max() < 5e-5


@pytest.mark.parametrize("avg", ["mean", "max"])
def test_long_clip_multicrop(golden, avg):
    """313 frames -> crops at 0,102,204 and the re-cropped tail (last 102 frames), models/uit.py:474-488."""
    g = golden(f"long_{avg}_uit_xxxs")
    o = _oracle("uit_xxxs", g["seed"], eval_avg=avg)
    assert o.num_crops(g["pcm"].shape[1]) == 4
    r = o.forward(g["pcm"])
    assert_mel_close(r["mel_db"], g["mel_db"])
    assert np.abs(r["logits"] - g["logits"]).max() < 1e-4
    assert np.abs(r["probs"] - g["probs"]).max() < 5e-5


def test_short_clip_fewer_time_patches(golden):
    """8000 samples -> 51 frames -> 3 time patches -> 12 tokens, time_pos_embed[..., :3] (models/uit.py:382)."""
    g = golden("short_uit_xxxs")
    r = _oracle("uit_xxxs", g["seed"]).forward(g["pcm"], want_stages=True)
    assert g["stages"].shape[-2] == 12 and r["stages"].shape[-2] == 12
    assert np.abs(r["stages"][:1] - g["stages"]).max() < 2e-4
    assert np.abs(r["probs"] - g["probs"]).max() < 5e-5


def test_527_classes(golden):
    g = golden("synth1s_uit_xxs_527")
    r = _oracle("uit_xxs", g["seed"], outputdim=527).forward(g["pcm"])
    assert r["probs"].shape == (3, 527)
    assert np.abs(r["probs"] - g["probs"]).max() < 5e-5


@pytest.mark.parametrize("name", ["uit_xxxs", "uit_xs"])
def test_torch_port_matches_reference_fixtures(golden, name):
    """oracle/torch_port.py (the op-for-op torch-CPU restatement used as bench.py's cpu_baseline)."""
    import torch
    from oracle.torch_port import TorchPort
    g = golden(f"synth1s_{name}")
    tp = TorchPort(make_state_dict(name, int(g["seed"])))
    p, l = tp.forward(torch.from_numpy(g["pcm"].astype(np.float32) / 32768.0), return_logits=True)
    assert np.abs(l.numpy() - g["logits"]).max() < 2e-5
    assert np.abs(p.numpy() - g["probs"]).max() < 1e-5
    g = golden("long_max_uit_xxxs")
    tp = TorchPort(make_state_dict("uit_xxxs", int(g["seed"])), eval_avg="max")
    p = tp.forward(torch.from_numpy(g["pcm"].astype(np.float32) / 32768.0))
    assert np.abs(p.numpy() - g["probs"]).max() < 1e-5


# ---------------------------------------------------------------- round-2 fixtures (trained-checkpoint magnitudes, 10 crops)
@pytest.mark.parametrize("name", ["uit_xs", "uit_xxxs"])
def test_trained_like_checkpoint(golden, name):
    """The oracle at trained-checkpoint magnitudes: reference logits beyond -10 (tests/golden/make_golden.py main_round2,
    weights from synth_weights.make_trained_like_state_dict loaded into the reference's own UITBase)."""
    from conftest import centred
    from uit_mobile_amd.synth import make_trained_like_state_dict
    g = golden(f"trained_{name}")
    assert np.abs(g["logits"]).max() >= 10.0
    r = Oracle(make_trained_like_state_dict(name, int(g["seed"]))).forward(g["pcm"], want_stages=True)
    assert np.abs(r["logits"] - g["logits"]).max() < 2e-4          # fp32 summation-order noise at |logit| ~ 10
    assert np.abs(r["probs"] - g["probs"]).max() < 5e-5
    assert np.abs(centred(r["stages"][0]) - centred(g["stages"][0])).max() < 2e-4


def test_trained_like_checkpoint_on_sample_clips(golden):
    from uit_mobile_amd.synth import make_trained_like_state_dict
    g, pcm = golden("trained_samples_uit_xs"), golden("samples_uit_xs")
    o = Oracle(make_trained_like_state_dict("uit_xs", int(g["seed"])))
    for key in ("water_000", "water_003", "85b877b5_nohash_0"):
        r = o.forward(pcm[f"{key}_pcm"][None])
        assert np.abs(r["logits"][0] - g[f"{key}_logits"]).max() < 2e-4
        assert np.abs(r["probs"][0] - g[f"{key}_probs"]).max() < 5e-5


def test_trained_like_short_clip(golden):
    from uit_mobile_amd.synth import make_trained_like_state_dict
    g = golden("trained_short_uit_xxxs")
    r = Oracle(make_trained_like_state_dict("uit_xxxs", int(g["seed"]))).forward(g["pcm"])
    assert np.abs(r["logits"] - g["logits"]).max() < 2e-4
    assert np.abs(r["probs"] - g["probs"]).max() < 5e-5


@pytest.mark.parametrize("avg", ["mean", "max"])
def test_ten_crop_clip(golden, avg):
    """160 000 samples -> 1001 frames -> 10 crops, the last one re-cropped to the final 102 frames (models/uit.py:468-488)."""
    g = golden("long10_uit_xxxs")
    o = _oracle("uit_xxxs", g["seed"], eval_avg=avg)
    assert o.num_crops(g["pcm"].shape[1]) == 10
    r = o.forward(g["pcm"])
    assert np.abs(r["logits"] - g["logits"]).max() < 1e-4
    assert np.abs(r["probs"] - g[f"probs_{avg}"]).max() < 5e-5


VARIANTS = [("audio_transformer_h128_d4_m3_relu", "relu"), ("audio_transformer_h128_d4_m3", "gelu"),
            ("audio_transformer_h128_d6_m3_relu", "relu"), ("audio_transformer_h128_d6_m3", "gelu")]


@pytest.mark.parametrize("name,act", VARIANTS)
def test_other_factories_full_attention_and_gelu(golden, name, act):
    """full-width Attention (models/uit.py:124-178) and the exact-erf GELU MLP of the non-bottleneck factories
    (:514-579), d4 at init scale, d6 at trained-like magnitudes, against the imported reference (make_golden.py --variants)."""
    from conftest import centred
    from uit_mobile_amd.synth import make_trained_like_state_dict
    g = golden(f"variant_{name}")
    mk = make_trained_like_state_dict if bool(g["trained"]) else make_state_dict
    r = Oracle(mk(name, int(g["seed"]), attention="full"), act=act).forward(g["pcm"], want_stages=True)
    assert np.abs(r["logits"] - g["logits"]).max() < 3e-4
    assert np.abs(r["probs"] - g["probs"]).max() < 5e-5
    assert np.abs(centred(r["stages"][0]) - centred(g["stages"][0])).max() < 2e-4


def test_pooling_dm(golden):
    """pooling='dm' (models/uit.py:405-412) against the imported reference: 3XS (full, short, two-crop clips), trained-like XS."""
    from uit_mobile_amd.synth import make_trained_like_state_dict
    g = golden("pooling_dm")
    o = Oracle(make_state_dict("uit_xxxs", int(g["xxxs_seed"])), pooling="dm")
    for pcm, want in ((g["pcm"], g["xxxs_probs"]), (g["xxxs_short_pcm"], g["xxxs_short_probs"]), (g["xxxs_crops_pcm"], g["xxxs_crops_probs"])):
        assert np.abs(o.forward(pcm)["probs"] - want).max() < 5e-5
    o = Oracle(make_trained_like_state_dict("uit_xs", int(g["xs_trained_seed"])), pooling="dm")
    assert np.abs(o.forward(g["pcm"])["probs"] - g["xs_trained_probs"]).max() < 5e-5


def test_pooling_token(golden):
    """pooling='token' (models/uit.py:389-392, 399-401; the UITBase default) against the imported reference."""
    from uit_mobile_amd.synth import make_trained_like_state_dict
    g = golden("pooling_token")
    for tag, name, trained, att, act in (("xxxs", "uit_xxxs", False, "bneck", "relu"), ("xs_trained", "uit_xs", True, "bneck", "relu"),
                                         ("h128", "audio_transformer_h128_d4_m3", False, "full", "gelu")):
        sd = dict((make_trained_like_state_dict if trained else make_state_dict)(name, int(g[f"{tag}_seed"]), attention=att))
        sd["cls_token"] = g[f"{tag}_cls_token"]
        o = Oracle(sd, act=act, pooling="token")
        r = o.forward(g["pcm"])
        assert np.abs(r["logits"] - g[f"{tag}_logits"]).max() < 2e-4 and np.abs(r["probs"] - g[f"{tag}_probs"]).max() < 5e-5
        if tag == "xxxs":
            assert np.abs(o.forward(g["xxxs_short_pcm"])["logits"] - g["xxxs_short_logits"]).max() < 1e-4


def test_other_kwargs_and_center_false(golden):
    """Round 3: the oracle's restatement of qkv_bias=False / LayerScale / init_bn=False (GlobalNormer) and of center=False against
    the fixtures the imported reference produced for those kwargs (tests/golden/make_golden.py --kwargs)."""
    from uit_mobile_amd.synth import make_kwargs_variant_state_dict, make_state_dict
    g = golden("kwargs_uit_xxxs")
    r = Oracle(make_kwargs_variant_state_dict("uit_xxxs", int(g["seed"]))).forward(g["pcm"])
    assert np.abs(r["logits"] - g["logits"]).max() < 2e-4 and np.abs(r["probs"] - g["probs"]).max() < 5e-5
    g = golden("nocenter_uit_xxxs")
    o = Oracle(make_state_dict("uit_xxxs", int(g["seed"]), 2.0), center=False)
    for tag, crops in (("1s", 1), ("2crop", 2)):
        r = o.forward(g[f"{tag}_pcm"])
        assert r["logits"].shape[1] == crops == g[f"{tag}_logits"].shape[1]
        assert np.abs(r["logits"] - g[f"{tag}_logits"]).max() < 3e-4 and np.abs(r["probs"] - g[f"{tag}_probs"]).max() < 1e-4
        assert r["mel_db"].shape == g[f"{tag}_mel_db"].shape


def test_win_size_shorter_than_n_fft(golden):
    """Round 4: win_size = 400 < n_fft = 512 (models/uit.py:290, 301-303 -> torch.stft zero-pads the window to n_fft, centred): the
    oracle's restatement against the fixtures of the imported reference (centre and no-centre framing, one and two crops), mel included."""
    from uit_mobile_amd.synth import hann_periodic, make_state_dict
    g = golden("win400_uit_xxxs")
    sd = dict(make_state_dict("uit_xxxs", int(g["seed"]), 2.0))
    sd["front_end.0.spectrogram.window"] = hann_periodic(400)
    for tag, crops, center in (("1s", 1, True), ("2crop", 2, True), ("nocenter", 1, False)):
        r = Oracle(sd, center=center).forward(g[f"{tag}_pcm"])
        assert r["logits"].shape[1] == crops == g[f"{tag}_logits"].shape[1] and r["mel_db"].shape == g[f"{tag}_mel_db"].shape
        assert_mel_close(r["mel_db"], g[f"{tag}_mel_db"])
        assert np.abs(r["logits"] - g[f"{tag}_logits"]).max() < 3e-4 and np.abs(r["probs"] - g[f"{tag}_probs"]).max() < 1e-4


HELDOUT_MODELS = (("xs207", "uit_xs", 207, "bneck", "relu"), ("xxxs203", "uit_xxxs", 203, "bneck", "relu"),
                  ("d4gelu206", "audio_transformer_h128_d4_m3", 206, "full", "gelu"))


@pytest.mark.parametrize("tag,name,seed,att,act", HELDOUT_MODELS)
def test_wide_held_out_set_under_the_reference(golden, tag, name, seed, att, act):
    """Round 5: the oracle on the WIDE held-out set -- all eleven recordings of the reference's samples/ (native length, one forward
    call each, as inference.py does) and one member of every adversarial family (impulses, clicks, DC offsets, hard clipping, +-1 LSB
    dither, Nyquist alternation, speech-like stacks at -40 dBFS ... as one batch) -- against what the IMPORTED reference answered on
    trained-like checkpoints (make_golden.py --heldout).  This is what lets the GPU sweeps use the oracle on such inputs."""
    from uit_mobile_amd.synth import adversarial_pcm_i16, make_trained_like_state_dict
    g = golden("heldout_wide")
    adv, names = adversarial_pcm_i16(0, 1)
    assert np.array_equal(adv, g["adv_pcm"]), "the seeded adversarial generator no longer reproduces the committed clips"
    o = Oracle(make_trained_like_state_dict(name, seed, attention=att), act=act)
    worst = 0.0
    for i in range(int(g["wav_count"])):
        r = o.forward(g[f"wav{i:02d}_pcm"][None])
        assert r["logits"][0].shape == g[f"{tag}_wav{i:02d}_logits"].shape
        worst = max(worst, float(np.abs(r["logits"][0] - g[f"{tag}_wav{i:02d}_logits"]).max()))
        assert np.abs(r["probs"][0] - g[f"{tag}_wav{i:02d}_probs"]).max() < 5e-5
    r = o.forward(g["adv_pcm"])
    d = np.abs(r["logits"] - g[f"{tag}_adv_logits"]).reshape(len(names), -1).max(1)
    print(f"{tag}: oracle vs reference, recordings {worst:.2e}; adversarial worst {d.max():.2e} ({names[int(d.argmax())]})")
    # fp32 summation-order noise at trained magnitudes (|logit| up to 12.9).  Stationary tones (bin_edge_tones, square, hard_clipped)
    # leave most of their bins on the fp32 FFT round-off floor, where ANY two fp32 FFTs differ -- pocketfft vs the oracle's radix-2
    # included (conftest.assert_mel_close) -- and trained-like weights turn that into up to ~2e-4 on a logit: the reference itself is
    # defined no better than that on such clips.
    assert worst < 2e-4 and d.max() < 4e-4
    assert np.abs(r["probs"] - g[f"{tag}_adv_probs"]).max() < 5e-5


# ---- round 6: geometries outside the published factories' (the library's general engine, csrc/general.hip, is checked against this oracle)
def general_oracle(case):
    from general_cases import state_dict_args
    from uit_mobile_amd.synth import make_general_state_dict
    k = case["kwargs"]
    return Oracle(make_general_state_dict(**state_dict_args(case)), target_length=k["target_length"], eval_avg=k.get("eval_avg", "mean"),
                  hop=k.get("hop_size", 160), heads=k["num_heads"], act=case["act"], pooling=k["pooling"], center=k.get("center", True))


@pytest.mark.parametrize("tag", ["mels128_fft1024", "patch8_token", "fft400_dm", "tl256_h4", "default_uitbase"])
def test_general_geometries_against_the_reference(golden, tag):
    """The oracle is parametric in every UITBase geometry kwarg; these fixtures (the reference's own UITBase on torch CPU) pin it there:
    128 mels / n_fft 1024; 8 x 8 patches, 3 heads, cls token; n_fft 400 without centring, 40 mels, pooling 'dm', max over crops;
    target_length 256, 4 heads, f_min / f_max / hop / win away from the defaults; UITBase() with the constructor's own defaults (768 wide,
    12 blocks, 12 heads, 85.7 M parameters) on 10 s clips."""
    from general_cases import GENERAL_CASES
    case = GENERAL_CASES[tag]
    g, o = golden(f"general_{tag}"), general_oracle(case)
    for i, (B, T) in enumerate(case["clips"]):
        r = o.forward(g[f"c{i}_pcm"])
        if f"c{i}_mel_db" in g:
            assert r["mel_db"].shape == g[f"c{i}_mel_db"].shape
            assert_mel_close(r["mel_db"], g[f"c{i}_mel_db"])
        # clip 1 of every batch is a full-scale pure tone: most of its bins sit on the fp32 FFT round-off floor, where any two FFT
        # algorithms differ (conftest.assert_mel_close; the same clip in test_synth_1s_all_stages) -- 4e-4 on logits there, 1e-5 elsewhere
        tone = np.arange(B) == 1
        dp = np.abs(r["probs"] - g[f"c{i}_probs"]).reshape(B, -1).max(1)
        assert dp[~tone].max() < 2e-5 and dp.max() < 1e-4, (tag, i, dp)
        if f"c{i}_logits" in g:
            dl = np.abs(r["logits"] - g[f"c{i}_logits"]).reshape(B, -1).max(1)
            assert dl[~tone].max() < 1e-4 and dl.max() < 6e-4, (tag, i, dl)
