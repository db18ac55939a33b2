"""Test-only stand-ins that let the reference's `models/uit.py` be imported in the build container.

TEST INFRASTRUCTURE -- used only by `tests/golden/make_golden.py` (fixture generation, this container
only; /root/reference does not exist on the GPU box).  Nothing in the product package imports this.

The reference imports three packages that are not installed here:

* ``loguru``   (models/uit.py:5)  -- logging only, no arithmetic.
* ``timm``     (models/uit.py:8-9) -- ``to_2tuple``, ``DropPath`` (identity in eval), ``trunc_normal_``.
* ``torchaudio.transforms`` (models/uit.py:7, call site 298-308) -- MelSpectrogram + AmplitudeToDB.
  This one carries arithmetic.  torchaudio==0.13.0 (requirements.txt:17) is an un-vendored third-party
  dependency, so its published algorithm is restated below on top of ``torch.stft`` (which is what
  torchaudio itself bottoms out in).  The reference holds no test pinning this boundary, so the
  front-end is "parity unpinned by the reference"; it is cross-checked in make_golden.py against the
  independent numpy implementation in ``transformers.audio_utils``.

Module/buffer names mirror real torchaudio so the state_dict key set is the real one
(``front_end.0.spectrogram.window``, ``front_end.0.mel_scale.fb``).
"""
import math
import sys
import types

import torch
import torch.nn as nn


# --------------------------------------------------------------------------- torchaudio 0.13 restatement
def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
    """torchaudio.functional.melscale_fbanks (0.13): HTK mel scale, triangular filters, norm=None."""
    assert norm is None and mel_scale == "htk"
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]                          # (n_mels + 1)
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)      # (n_freqs, n_mels + 2)
    zero = torch.zeros(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(zero, torch.min(down_slopes, up_slopes))
    return fb


class Spectrogram(nn.Module):
    def __init__(self, n_fft, win_length, hop_length, center=True, pad_mode="reflect", power=2.0):
        super().__init__()
        self.n_fft, self.win_length, self.hop_length = n_fft, win_length, hop_length
        self.center, self.pad_mode, self.power = center, pad_mode, power
        self.register_buffer("window", torch.hann_window(win_length))   # periodic=True default

    def forward(self, waveform):
        shape = waveform.size()
        waveform = waveform.reshape(-1, shape[-1])
        spec = torch.stft(waveform, n_fft=self.n_fft, hop_length=self.hop_length,
                          win_length=self.win_length, window=self.window, center=self.center,
                          pad_mode=self.pad_mode, normalized=False, onesided=True,
                          return_complex=True)
        spec = spec.reshape(shape[:-1] + spec.shape[-2:])
        return spec.abs().pow(self.power)


class MelScale(nn.Module):
    def __init__(self, n_mels, sample_rate, f_min, f_max, n_stft):
        super().__init__()
        f_max = f_max if f_max is not None else float(sample_rate // 2)
        self.register_buffer("fb", melscale_fbanks(n_stft, f_min, f_max, n_mels, sample_rate))

    def forward(self, specgram):
        return torch.matmul(specgram.transpose(-1, -2), self.fb).transpose(-1, -2)


class MelSpectrogram(nn.Module):
    def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, f_min=0.0,
                 f_max=None, pad=0, n_mels=128, power=2.0, normalized=False, center=True,
                 pad_mode="reflect", **_):
        super().__init__()
        win_length = win_length if win_length is not None else n_fft
        hop_length = hop_length if hop_length is not None else win_length // 2
        self.spectrogram = Spectrogram(n_fft, win_length, hop_length, center, pad_mode, power)
        self.mel_scale = MelScale(n_mels, sample_rate, f_min, f_max, n_fft // 2 + 1)

    def forward(self, waveform):
        return self.mel_scale(self.spectrogram(waveform))


def amplitude_to_DB(x, multiplier, amin, db_multiplier, top_db=None):
    """torchaudio.functional.amplitude_to_DB (0.13), including the packed-channel reshape that makes the
    top_db floor batch-global for the 3-D [B, F, T] tensor the reference passes."""
    x_db = multiplier * torch.log10(torch.clamp(x, min=amin))
    x_db -= multiplier * db_multiplier
    if top_db is not None:
        shape = x_db.size()
        packed_channels = shape[-3] if x_db.dim() > 2 else 1
        x_db = x_db.reshape(-1, packed_channels, shape[-2], shape[-1])
        x_db = torch.max(x_db, (x_db.amax(dim=(-3, -2, -1)) - top_db).view(-1, 1, 1, 1))
        x_db = x_db.reshape(shape)
    return x_db


class AmplitudeToDB(nn.Module):
    def __init__(self, stype="power", top_db=None):
        super().__init__()
        self.multiplier = 10.0 if stype == "power" else 20.0
        self.amin, self.ref_value, self.top_db = 1e-10, 1.0, top_db
        self.db_multiplier = math.log10(max(self.amin, self.ref_value))

    def forward(self, x):
        return amplitude_to_DB(x, self.multiplier, self.amin, self.db_multiplier, self.top_db)


# --------------------------------------------------------------------------- install + import
def _install():
    loguru = types.ModuleType("loguru")

    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None
    loguru.logger = _Logger()
    sys.modules.setdefault("loguru", loguru)

    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    layers = types.ModuleType("timm.models.layers")
    helpers = types.ModuleType("timm.models.layers.helpers")

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    class DropPath(nn.Module):                       # identity in eval; never instantiated for uit_* (rate 0)
        def __init__(self, drop_prob=0.0):
            super().__init__()

        def forward(self, x):
            return x

    def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return torch.nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)

    helpers.to_2tuple = to_2tuple
    layers.DropPath, layers.trunc_normal_, layers.helpers = DropPath, trunc_normal_, helpers
    timm_models.layers = layers
    timm.models = timm_models
    for name, mod in (("timm", timm), ("timm.models", timm_models),
                      ("timm.models.layers", layers), ("timm.models.layers.helpers", helpers)):
        sys.modules.setdefault(name, mod)

    ta = types.ModuleType("torchaudio")
    tr = types.ModuleType("torchaudio.transforms")
    tr.MelSpectrogram, tr.AmplitudeToDB, tr.Spectrogram, tr.MelScale = (
        MelSpectrogram, AmplitudeToDB, Spectrogram, MelScale)
    ta.transforms = tr
    sys.modules.setdefault("torchaudio", ta)
    sys.modules.setdefault("torchaudio.transforms", tr)


def import_reference_uit(ref_root="/root/reference"):
    """Import the reference's own models/uit.py (unmodified, from where it lies) and return the module."""
    sys.dont_write_bytecode = True                   # never write __pycache__ into the read-only tree
    _install()
    import importlib.util
    spec = importlib.util.spec_from_file_location("_reference_uit", f"{ref_root}/models/uit.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
