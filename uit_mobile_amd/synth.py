"""Deterministic synthetic checkpoints and PCM in the reference's state_dict layout (numpy only): what bench.py, the smoke test and
the test-suite feed the product and the oracle with.

No pretrained UiT checkpoint is reachable from the build container (zenodo, models/uit.py:643-653), so parity
is pinned on seeded weights.  The SAME function is used by make_golden.py (to load the weights into the
reference's own UITBase with strict=True -- which also pins the key set / shapes) and by the tests and bench
(to feed the oracle and the HIP path), so the weights never have to be committed.

Distributions follow the reference init (models/uit.py:329-336, 369-376; Conv2d default init) but with
non-trivial biases, LayerNorm gains and BatchNorm running stats so every folded term is exercised; `scale`
multiplies the matmul weights to mimic trained magnitudes.
"""
import math

import numpy as np

DEPTH = {"uit_xs": 12, "uit_xxs": 6, "uit_xxxs": 4,
         "audio_transformer_h128_d4_m3": 4, "audio_transformer_h128_d4_m3_relu": 4,
         "audio_transformer_h128_d6_m3": 6, "audio_transformer_h128_d6_m3_relu": 6}


def hann_periodic(n):
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)).astype(np.float32)


def melscale_fbanks(n_freqs=257, f_min=0.0, f_max=8000.0, n_mels=64, sample_rate=16000):
    """float32 restatement of torchaudio.functional.melscale_fbanks (htk, norm=None)."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs, dtype=np.float32)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2, dtype=np.float32)
    f_pts = (700.0 * (np.power(np.float32(10.0), m_pts / np.float32(2595.0)) - 1.0)).astype(np.float32)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up)).astype(np.float32)


def make_state_dict(name="uit_xs", seed=0, scale=1.0, outputdim=537, target_length=102, dim=128, n_mels=64, attention="bneck"):
    """attention: 'bneck' (BNeckAttention, inner = dim/4: the uit_* family) or 'full' (Attention, inner = dim: the
    audio_transformer_h128_* factories, models/uit.py:514-579)."""
    depth = DEPTH[name]
    rng = np.random.default_rng(seed)
    gt, gf, inner, hidden = target_length // 16, n_mels // 16, (dim if attention == "full" else dim // 4), int(dim * 3.0)

    def tn(*shape, std=0.02):
        return (np.clip(rng.standard_normal(shape), -2.0, 2.0) * std * scale).astype(np.float32)

    def nrm(*shape, std=1.0, mean=0.0):
        return (mean + std * rng.standard_normal(shape)).astype(np.float32)

    sd = {}
    sd["cls_token"] = nrm(1, 1, dim, std=1e-6)
    sd["token_pos_embed"] = nrm(1, dim, std=0.02)
    sd["time_pos_embed"] = nrm(1, dim, 1, gt, std=0.02)
    sd["freq_pos_embed"] = nrm(1, dim, gf, 1, std=0.02)
    sd["front_end.0.spectrogram.window"] = hann_periodic(512)
    sd["front_end.0.mel_scale.fb"] = melscale_fbanks(257, 0.0, 8000.0, n_mels, 16000)
    sd["init_bn.1.weight"] = nrm(n_mels, std=0.2, mean=1.0)
    sd["init_bn.1.bias"] = nrm(n_mels, std=0.1)
    sd["init_bn.1.running_mean"] = nrm(n_mels, std=10.0, mean=-30.0)
    sd["init_bn.1.running_var"] = (200.0 + 400.0 * rng.random(n_mels)).astype(np.float32)
    sd["init_bn.1.num_batches_tracked"] = np.array(1000, dtype=np.int64)
    bound = 1.0 / 16.0
    sd["patch_embed.proj.weight"] = ((rng.random((dim, 1, 16, 16)) * 2 - 1) * bound * scale).astype(np.float32)
    sd["patch_embed.proj.bias"] = ((rng.random(dim) * 2 - 1) * bound).astype(np.float32)
    for b in range(depth):
        p = f"blocks.{b}."
        sd[p + "norm1.weight"] = nrm(dim, std=0.3, mean=1.0)
        sd[p + "norm1.bias"] = nrm(dim, std=0.1)
        sd[p + "attn.qkv.weight"] = tn(3 * inner, dim)
        sd[p + "attn.qkv.bias"] = nrm(3 * inner, std=0.1)
        sd[p + "attn.proj.weight"] = tn(dim, inner)
        sd[p + "attn.proj.bias"] = nrm(dim, std=0.1)
        sd[p + "norm2.weight"] = nrm(dim, std=0.3, mean=1.0)
        sd[p + "norm2.bias"] = nrm(dim, std=0.1)
        sd[p + "mlp.fc1.weight"] = tn(hidden, dim)
        sd[p + "mlp.fc1.bias"] = nrm(hidden, std=0.1)
        sd[p + "mlp.fc2.weight"] = tn(dim, hidden)
        sd[p + "mlp.fc2.bias"] = nrm(dim, std=0.1)
    sd["norm.weight"] = nrm(dim, std=0.3, mean=1.0)
    sd["norm.bias"] = nrm(dim, std=0.1)
    sd["outputlayer.0.weight"] = nrm(dim, std=0.3, mean=1.0)
    sd["outputlayer.0.bias"] = nrm(dim, std=0.1)
    sd["outputlayer.1.weight"] = tn(outputdim, dim)
    sd["outputlayer.1.bias"] = nrm(outputdim, std=0.1)
    return sd


def synth_pcm_i16(B, T, seed=0):
    """Bench/test input generator: uniform noise in [-0.5,0.5) as int16 PCM; every 8th clip is a
    loud 1 kHz-ish tone burst, every 8th+1 clip is digital silence over its last quarter (exercises the floor)."""
    rng = np.random.default_rng(1000 + seed)
    x = (rng.random((B, T), dtype=np.float32) - 0.5)
    t = np.arange(T, dtype=np.float32)
    for b in range(B):
        if b % 8 == 1:
            x[b] = 0.9 * np.sin(2 * np.pi * (440.0 + 37.0 * b) * t / 16000.0)
        if b % 8 == 2:
            x[b, T - T // 4:] = 0.0
        if b % 8 == 3:
            x[b] *= 0.001
    return np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)


def make_trained_like_state_dict(name="uit_xs", seed=0, outputdim=537, target_length=102, dim=128, n_mels=64, attention="bneck"):
    """A seeded checkpoint with the MAGNITUDES of a trained UiT (no trained weights are reachable from the build
    container; the reference's published outputs -- `Speech 1.0000`, `Keyword: on 0.9999`, README.md:124-127 -- show
    logits beyond +-9, and AudioSet taggers sit at logits around -5..-9 for the hundreds of absent classes):

      * classifier: weights std 0.1 (row norm ~1.1), biases -4.5 +- 2  =>  reference logits span about -15..+4;
      * matmul weights 3-4x the init scale (sharper attention: q.k logits of a few units; MLP updates comparable to the
        residual stream instead of a small perturbation of it), non-zero biases everywhere;
      * LayerNorm gains far from 1 (0.3..2.5, a few negative) and shifts of a few tenths;
      * a residual stream with structure the init never has: a COMMON per-token offset (every channel shifted the same way by
        the positional embeddings and the projection biases, so the per-token mean is several times the per-token standard
        deviation of the bulk) plus three "massive activation" channels (tens of units, as trained ViTs have);
      * init_bn running statistics with the spectral tilt of real log-mel data (-18 dB at the bottom to -48 dB at the top,
        variance 120..520) and gains/shifts away from (1, 0).
    Same key set / shapes as make_state_dict (loaded with strict=True into the reference's own UITBase by make_golden.py)."""
    depth = DEPTH[name]
    rng = np.random.default_rng(7000 + seed)
    gt, gf, inner, hidden = target_length // 16, n_mels // 16, (dim if attention == "full" else dim // 4), int(dim * 3.0)

    def nrm(*shape, std=1.0, mean=0.0):
        return (mean + std * rng.standard_normal(shape)).astype(np.float32)

    def gain(n):
        g = np.exp(rng.normal(0.0, 0.45, n)).astype(np.float32)          # log-normal gains 0.3..2.5
        g[rng.integers(0, n, 2)] *= -1.0                                  # a couple of negative gains
        return g

    massive = rng.choice(dim, 3, replace=False)
    sd = {}
    sd["cls_token"] = nrm(1, 1, dim, std=0.02)
    sd["token_pos_embed"] = nrm(1, dim, std=0.1)
    sd["time_pos_embed"] = nrm(1, dim, 1, gt, std=0.25, mean=2.0)        # common offset: same sign on every channel
    sd["freq_pos_embed"] = nrm(1, dim, gf, 1, std=0.25, mean=2.0)
    sd["front_end.0.spectrogram.window"] = hann_periodic(512)
    sd["front_end.0.mel_scale.fb"] = melscale_fbanks(257, 0.0, 8000.0, n_mels, 16000)
    tilt = np.linspace(-18.0, -48.0, n_mels)
    sd["init_bn.1.weight"] = nrm(n_mels, std=0.25, mean=1.1)
    sd["init_bn.1.bias"] = nrm(n_mels, std=0.3)
    sd["init_bn.1.running_mean"] = (tilt + 3.0 * rng.standard_normal(n_mels)).astype(np.float32)
    sd["init_bn.1.running_var"] = (120.0 + 400.0 * rng.random(n_mels)).astype(np.float32)
    sd["init_bn.1.num_batches_tracked"] = np.array(250000, dtype=np.int64)
    sd["patch_embed.proj.weight"] = nrm(dim, 1, 16, 16, std=0.05)
    pe_b = nrm(dim, std=0.3, mean=0.8)
    pe_b[massive] += np.array([28.0, -22.0, 17.0], np.float32)            # massive-activation channels
    sd["patch_embed.proj.bias"] = pe_b
    for b in range(depth):
        p = f"blocks.{b}."
        sd[p + "norm1.weight"] = gain(dim)
        sd[p + "norm1.bias"] = nrm(dim, std=0.3)
        sd[p + "attn.qkv.weight"] = nrm(3 * inner, dim, std=0.08)
        sd[p + "attn.qkv.bias"] = nrm(3 * inner, std=0.3)
        sd[p + "attn.proj.weight"] = nrm(dim, inner, std=0.08)
        sd[p + "attn.proj.bias"] = nrm(dim, std=0.2, mean=0.15)           # keeps feeding the common offset
        sd[p + "norm2.weight"] = gain(dim)
        sd[p + "norm2.bias"] = nrm(dim, std=0.3)
        sd[p + "mlp.fc1.weight"] = nrm(hidden, dim, std=0.07)
        sd[p + "mlp.fc1.bias"] = nrm(hidden, std=0.3, mean=-0.2)
        sd[p + "mlp.fc2.weight"] = nrm(dim, hidden, std=0.06)
        fc2_b = nrm(dim, std=0.2, mean=0.15)
        if b == 1:
            fc2_b[massive] += np.array([35.0, 30.0, -40.0], np.float32)   # ... and grow in an early block, as observed
        sd[p + "mlp.fc2.bias"] = fc2_b
    sd["norm.weight"] = gain(dim)
    sd["norm.bias"] = nrm(dim, std=0.3)
    sd["outputlayer.0.weight"] = gain(dim)
    sd["outputlayer.0.bias"] = nrm(dim, std=0.3)
    sd["outputlayer.1.weight"] = nrm(outputdim, dim, std=0.1)
    sd["outputlayer.1.bias"] = nrm(outputdim, std=2.0, mean=-4.5)
    return sd


def make_general_state_dict(embed_dim=128, depth=2, mlp_ratio=3.0, attention="bneck", n_mels=64, n_fft=512, win_size=512, f_min=0.0, f_max=8000.0,
                            patch_size=16, target_length=102, outputdim=537, seed=0):
    """Checkpoint of a UITBase built with ANY geometry (models/uit.py:253-363) -- the state_dict key set and shapes the reference's
    constructor gives those kwargs (loaded with strict=True into the reference's own UITBase by make_golden.py --general), magnitudes
    after make_trained_like_state_dict (matmul weights scaled with 1 / sqrt(fan-in) so that activations stay O(1..10) at any width).
    Feeds the general engine (csrc/general.hip), the oracle and the reference alike."""
    rng = np.random.default_rng(13000 + seed)
    dim, P = embed_dim, patch_size
    gt, gf, inner, hidden = target_length // P, n_mels // P, (dim if attention == "full" else dim // 4), int(dim * mlp_ratio)

    def nrm(*shape, std=1.0, mean=0.0):
        return (mean + std * rng.standard_normal(shape)).astype(np.float32)

    def gain(n):
        g = np.exp(rng.normal(0.0, 0.4, n)).astype(np.float32)
        g[rng.integers(0, n, 2)] *= -1.0
        return g

    sd = {}
    sd["cls_token"] = nrm(1, 1, dim, std=0.3)
    sd["token_pos_embed"] = nrm(1, dim, std=0.3)
    sd["time_pos_embed"] = nrm(1, dim, 1, gt, std=0.25, mean=1.0)
    sd["freq_pos_embed"] = nrm(1, dim, gf, 1, std=0.25, mean=1.0)
    sd["front_end.0.spectrogram.window"] = hann_periodic(win_size)
    sd["front_end.0.mel_scale.fb"] = melscale_fbanks(n_fft // 2 + 1, f_min, f_max, n_mels, 16000)
    tilt = np.linspace(-18.0, -48.0, n_mels)
    sd["init_bn.1.weight"] = nrm(n_mels, std=0.25, mean=1.1)
    sd["init_bn.1.bias"] = nrm(n_mels, std=0.3)
    sd["init_bn.1.running_mean"] = (tilt + 3.0 * rng.standard_normal(n_mels)).astype(np.float32)
    sd["init_bn.1.running_var"] = (120.0 + 400.0 * rng.random(n_mels)).astype(np.float32)
    sd["init_bn.1.num_batches_tracked"] = np.array(250000, dtype=np.int64)
    sd["patch_embed.proj.weight"] = nrm(dim, 1, P, P, std=0.8 / P)
    sd["patch_embed.proj.bias"] = nrm(dim, std=0.3, mean=0.5)
    for b in range(depth):
        p = f"blocks.{b}."
        sd[p + "norm1.weight"] = gain(dim)
        sd[p + "norm1.bias"] = nrm(dim, std=0.3)
        sd[p + "attn.qkv.weight"] = nrm(3 * inner, dim, std=0.9 / math.sqrt(dim))
        sd[p + "attn.qkv.bias"] = nrm(3 * inner, std=0.3)
        sd[p + "attn.proj.weight"] = nrm(dim, inner, std=0.5 / math.sqrt(inner))
        sd[p + "attn.proj.bias"] = nrm(dim, std=0.2, mean=0.1)
        sd[p + "norm2.weight"] = gain(dim)
        sd[p + "norm2.bias"] = nrm(dim, std=0.3)
        sd[p + "mlp.fc1.weight"] = nrm(hidden, dim, std=0.8 / math.sqrt(dim))
        sd[p + "mlp.fc1.bias"] = nrm(hidden, std=0.3, mean=-0.2)
        sd[p + "mlp.fc2.weight"] = nrm(dim, hidden, std=1.2 / math.sqrt(hidden))
        sd[p + "mlp.fc2.bias"] = nrm(dim, std=0.2, mean=0.1)
    sd["norm.weight"] = gain(dim)
    sd["norm.bias"] = nrm(dim, std=0.3)
    sd["outputlayer.0.weight"] = gain(dim)
    sd["outputlayer.0.bias"] = nrm(dim, std=0.3)
    sd["outputlayer.1.weight"] = nrm(outputdim, dim, std=1.1 / math.sqrt(dim))
    sd["outputlayer.1.bias"] = nrm(outputdim, std=2.0, mean=-3.0)
    return sd


def make_kwargs_variant_state_dict(name="uit_xxxs", seed=0, scale=2.0, outputdim=537):
    """Checkpoint of a model built with the OTHER kwargs of UITBase (models/uit.py:253-293, Block :206-243): qkv_bias=False (no
    attn.qkv.bias keys), init_values=<x> (LayerScale: blocks.N.ls1.gamma / ls2.gamma, gains 0.2..1.2) and init_bn=False (no init_bn
    keys: the reference's GlobalNormer(-10, 20) takes their place).  Otherwise make_state_dict's distributions."""
    sd = dict(make_state_dict(name, seed, scale, outputdim))
    rng = np.random.default_rng(9000 + seed)
    for k in [k for k in sd if k.startswith("init_bn.") or k.endswith("attn.qkv.bias")]:
        del sd[k]
    for b in range(DEPTH[name]):
        sd[f"blocks.{b}.ls1.gamma"] = (0.2 + rng.random(128)).astype(np.float32)
        sd[f"blocks.{b}.ls2.gamma"] = (0.2 + rng.random(128)).astype(np.float32)
    return sd


def make_structured_trained_like_state_dict(name="uit_xs", seed=0, outputdim=537, target_length=102, dim=128, n_mels=64, attention="bneck"):
    """A SECOND family of trained-like checkpoints (round 5): make_trained_like_state_dict's magnitudes, biases, LayerNorm gains and
    massive-activation channels, but matmul weights with the STRUCTURE gradient descent leaves and a Gaussian draw lacks:

      * heavy tails: Student-t (4 degrees of freedom) entries -- a handful of weights per row several times the bulk;
      * a low-rank component (rank 8, geometric spectrum) carrying half of every matrix's energy: correlated rows, so that rounding
        errors of different outputs meet the same operand directions;
      * output rows of very different norm (log-normal, sigma 0.5), and the input columns of the massive-activation channels 3x the rest.
    Every matrix is rescaled to the RMS the first family gives it, so logit magnitudes (and the tolerance's meaning) stay comparable.
    Used by tools/auto_sweep.py to check that the default mode's margins are not a property of Gaussian weights."""
    sd = dict(make_trained_like_state_dict(name, seed, outputdim, target_length, dim, n_mels, attention))
    rng = np.random.default_rng(11000 + seed)
    massive = np.argsort(-np.abs(sd["patch_embed.proj.bias"]))[:3]
    for k in [k for k in sd if k.endswith(("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight")) or k == "patch_embed.proj.weight"]:
        w0 = np.asarray(sd[k], np.float64)
        shape = w0.shape
        w0 = w0.reshape(shape[0], -1)
        rms = np.sqrt((w0 ** 2).mean())
        o, i = w0.shape
        bulk = rng.standard_t(4, size=(o, i)) / np.sqrt(2.0)
        r = min(8, o, i)
        u, v = np.linalg.qr(rng.standard_normal((o, r)))[0], np.linalg.qr(rng.standard_normal((i, r)))[0]
        sv = 0.7 ** np.arange(r)
        low = (u * sv) @ v.T
        low *= np.sqrt((bulk ** 2).sum() / (low ** 2).sum())             # half of the energy
        w = (bulk + low) * np.exp(rng.normal(0.0, 0.5, (o, 1)))
        if k.endswith(("attn.qkv.weight", "mlp.fc1.weight")):              # operands of the residual stream: the massive channels' columns
            w[:, massive] *= 3.0
        sd[k] = (w * (rms / np.sqrt((w ** 2).mean()))).reshape(shape).astype(np.float32)
    return sd


def make_sink_trained_like_state_dict(name="uit_xs", seed=0, outputdim=537, target_length=102, dim=128, n_mels=64, attention="bneck", gain=2.0):
    """A THIRD family (round 5): make_trained_like_state_dict with an ATTENTION SINK in every block and head, as trained transformers
    have them: the six tokens of frequency row 0 carry a key component along a direction u (written into W_k through the direction of
    that row's positional embedding), and every query looks for u (q bias): scores of a few units x `gain` on those keys for all
    queries.  A sink key is an outlier -- centring on the all-token mean does not shrink it -- so its fp16 rounding error, q . dk, is
    large; what this does to the activation floor: tools/err_attrib_sink.py."""
    sd = {k: np.array(v) for k, v in make_trained_like_state_dict(name, seed, outputdim, target_length, dim, n_mels, attention).items()}
    rng = np.random.default_rng(13000 + seed)
    inner = dim if attention == "full" else dim // 4
    hd = inner // 2
    fp = sd["freq_pos_embed"][0, :, :, 0].astype(np.float64)
    p0 = fp[:, 0] - fp.mean(1)
    p0 /= np.linalg.norm(p0)
    for b in range(DEPTH[name]):
        W, bias = sd[f"blocks.{b}.attn.qkv.weight"], sd[f"blocks.{b}.attn.qkv.bias"]
        g1 = float(np.abs(sd[f"blocks.{b}.norm1.weight"]).mean())
        for h in range(2):
            u = rng.standard_normal(hd)
            u /= np.linalg.norm(u)
            bias[h * hd:(h + 1) * hd] += (gain * 6.0 * u).astype(np.float32)
            W[inner + h * hd:inner + (h + 1) * hd] += (gain * 8.0 * np.outer(u, p0) / g1).astype(np.float32)
    return sd


# ---------------------------------------------------------------------------------------------- adversarial held-out signals
# Round 5: the signal families a calibration fit on noise / tones / chirps / harmonic stacks leaves
# unconstrained.  Every family is a function (rng, T) -> float64 [T] in full-scale units; adversarial_pcm_i16 draws `per_family`
# members of each (a fixed seed = the fixed held-out set of the parity tests; the random search of the GPU tests walks other seeds).
# Deliberately NOT the library's own acceptance clips (csrc/api.cpp: builtin_heldout_set) -- same kinds, other parameters.
_LSB = 1.0 / 32768.0


def _speechlike(rng, T, level):
    t = np.arange(T) / 16000.0
    f0 = rng.uniform(90.0, 260.0)
    ph = 2 * np.pi * (f0 * t + rng.uniform(0.3, 1.5) * np.sin(2 * np.pi * rng.uniform(3.0, 7.0) * t))
    formants = [(rng.uniform(300, 900), 250.0, 1.0), (rng.uniform(1100, 2500), 450.0, 0.6), (rng.uniform(2600, 3600), 650.0, 0.3)]
    acc = np.zeros(T)
    for h in range(1, 22):
        fh = h * f0
        if fh > 7600:
            break
        g = 0.1 + sum(a * np.exp(-0.5 * ((fh - c) / w) ** 2) for c, w, a in formants)
        acc += g / h * np.sin(h * ph)
    env = (0.5 + 0.5 * np.sin(2 * np.pi * rng.uniform(2.5, 5.5) * t + rng.uniform(0, 6.28))) ** 2
    return level * env * acc / max(1e-9, np.abs(acc).max())


def _family_impulse_train(rng, T):
    x = np.zeros(T)
    per = int(rng.integers(40, 900))
    idx = np.arange(int(rng.integers(0, per)), T, per)
    x[idx] = rng.uniform(0.2, 1.0) * np.where(np.arange(len(idx)) % 2 == 0, 1.0, rng.choice([-1.0, 1.0]))
    return x


def _family_single_click(rng, T):
    x = np.zeros(T)
    x[int(rng.integers(0, T))] = rng.choice([-1.0, 1.0]) * rng.choice([1.0, 0.5, 8 * _LSB, _LSB])
    return x


def _family_dc_offset(rng, T):
    t = np.arange(T) / 16000.0
    return rng.choice([-0.5, 0.5]) + rng.uniform(0.0, 0.3) * np.sin(2 * np.pi * rng.uniform(80, 4000) * t) + rng.uniform(0.0, 0.02) * rng.standard_normal(T)


def _family_hard_clipped(rng, T):
    t = np.arange(T) / 16000.0
    return np.clip(rng.uniform(1.5, 8.0) * np.sin(2 * np.pi * rng.uniform(60, 2500) * t + rng.uniform(0, 6.28)), -1.0, 1.0)


def _family_lsb_dither(rng, T):
    return rng.integers(-1, 2, T) * _LSB * rng.choice([1, 1, 2])


def _family_nyquist(rng, T):
    a = rng.choice([0.5, 0.9, 0.05, _LSB, 3 * _LSB])
    return a * np.where(np.arange(T) % 2 == 0, 1.0, -1.0) + (rng.integers(-1, 2, T) * _LSB if a < 1e-3 else 0.0)


def _family_speechlike_quiet(rng, T):
    return _speechlike(rng, T, 0.01)                                     # -40 dBFS


def _family_speechlike(rng, T):
    return _speechlike(rng, T, rng.uniform(0.1, 0.7))


def _family_clipped_noise(rng, T):
    return np.clip(rng.uniform(1.0, 4.0) * rng.standard_normal(T) * 0.5, -1.0, 1.0)


def _family_square(rng, T):
    t = np.arange(T) / 16000.0
    return rng.uniform(0.2, 0.95) * np.sign(np.sin(2 * np.pi * rng.uniform(30, 1500) * t + 1e-3))


def _family_silence_then_burst(rng, T):
    x = np.zeros(T)
    on = int(T * rng.uniform(0.55, 0.97))
    x[on:] = rng.uniform(0.3, 0.95) * (rng.random(T - on) - 0.5) * 2
    return x if rng.random() < 0.5 else x[::-1].copy()


def _family_exp_sweep(rng, T):
    t = np.arange(T) / 16000.0
    f0, f1 = sorted(rng.uniform(20, 7900, 2))
    k = np.log(max(f1, f0 + 1.0) / f0) / t[-1]
    return rng.uniform(0.05, 0.9) * np.sin(2 * np.pi * f0 * (np.exp(k * t) - 1.0) / k)


def _family_bin_edge_tones(rng, T):
    t = np.arange(T) / 16000.0
    k = int(rng.integers(2, 250))
    return 0.45 * np.sin(2 * np.pi * (k + 0.5) * 31.25 * t) + 0.45 * np.sin(2 * np.pi * (k + 0.5 + rng.choice([0.5, 1.0, 8.0])) * 31.25 * t + rng.uniform(0, 6.28))


def _family_resonances(rng, T):
    t = np.arange(T) / 16000.0
    tt = np.mod(t, rng.uniform(0.08, 0.4))
    return 0.6 * sum(np.exp(-rng.uniform(5, 40) * tt) * np.sin(2 * np.pi * rng.uniform(150, 5000) * tt) / (r + 1) for r in range(3))


ADVERSARIAL_FAMILIES = [("impulse_train", _family_impulse_train), ("single_click", _family_single_click), ("dc_offset", _family_dc_offset),
                        ("hard_clipped", _family_hard_clipped), ("lsb_dither", _family_lsb_dither), ("nyquist", _family_nyquist),
                        ("speechlike_-40dBFS", _family_speechlike_quiet), ("speechlike", _family_speechlike), ("clipped_noise", _family_clipped_noise),
                        ("square", _family_square), ("silence_then_burst", _family_silence_then_burst), ("exp_sweep", _family_exp_sweep),
                        ("bin_edge_tones", _family_bin_edge_tones), ("resonances", _family_resonances)]


def adversarial_pcm_i16(seed=0, per_family=1, T=16000):
    """-> (pcm int16 [len(ADVERSARIAL_FAMILIES) * per_family, T], names): `per_family` seeded members of every family, family-major."""
    rng = np.random.default_rng(31000 + seed)
    rows, names = [], []
    for name, fn in ADVERSARIAL_FAMILIES:
        for j in range(per_family):
            rows.append(np.clip(np.round(np.asarray(fn(rng, T), np.float64) * 32768.0), -32768, 32767).astype(np.int16))
            names.append(f"{name}#{seed}.{j}")
    return np.stack(rows), names
