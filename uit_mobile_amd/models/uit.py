"""Host-side mirror of the reference's `models/uit.py` model API, backed by libuitk.so (HIP, gfx950).

What is kept from the reference interface (citations into /root/reference/models/uit.py):
  * factories `uit_xs / uit_xxs / uit_xxxs(**kwargs)` (:581-635) and `PRETRAINED_CHECKPOINTS` (:639-655);
  * `UITBase(**kwargs)` kwargs (:253-293) incl. `outputdim`, `target_length`, `eval_avg`, front-end knobs;
  * the exact `state_dict()` key set / shapes, `load_state_dict(sd, strict=True)` with the positional-embedding
    resize rule (:416-450), `.eval()`, `.to(device)`, attributes `target_length`, `hop_size`, `n_mels`,
    `outputdim`, `patch_embed.grid_size` (read by evaluate.py:57-62, 255-259);
  * `model(wave[B, T]) -> probs[B, outputdim]` (:452-493, eval branch: crops, mean/max over crops).

What is different: the torch sub-modules below only HOLD parameters (so checkpoints load unchanged); all
arithmetic of `forward` runs in two HIP kernels (csrc/frontend.hip, csrc/encoder.hip).  Training-mode branches
(mixup, wav/spec transforms, patch-out) are not implemented; `forward` in training mode raises.  There is no CPU
path: a non-HIP input raises RuntimeError.
"""
import ctypes
import math
import operator
import warnings

import numpy as np
import torch
import torch.nn as nn

from .. import _lib

_VERSION_OF = operator.attrgetter("_version")
_OPTIONS = ('fe_asym_pm', 'fe_wg_per_cu', 'enc_waves', 'split_pe', 'enc_generic_clamp', 'fe_tail_skip', 'calib_accept_pct', 'calib_budget_e6', 'calib_short_check')      # uitk_set_option names
_CALIB_OPTIONS = ('calib_accept_pct', 'calib_budget_e6', 'calib_short_check')            # (meaningful on a calibrating handle only)
# 'auto' (the factories' default, round 5) and 'f16c' both START as the f16 kernel on weights with calibrated rounding; 'auto' then holds the
# handle to the library's absolute deviation budget and rebuilds it in the family's split-weight precision when it is over (UITBase._handle)
# 'fp8c' / 'fp8w' (round 5, BASELINE config #5): the fp8 kernel on weights GPTQ-rounded to the MX e4m3 grid / the f16 kernel on fp8-grid weights
# ("fp8 weights" read literally), both calibrated like 'f16c'; 'fp8' keeps its weights rounded to nearest (the reference point)
_PRECISIONS = {'auto': _lib.PREC_F16, 'f16': _lib.PREC_F16, 'f16c': _lib.PREC_F16, 'bf16': _lib.PREC_BF16, 'fp8': _lib.PREC_FP8, 'fp8c': _lib.PREC_FP8,
               'fp8w': _lib.PREC_FP8W, 'f16x2': _lib.PREC_F16X2, 'f16l6': _lib.PREC_F16L6, 'f16a': _lib.PREC_F16A, 'f16ax': _lib.PREC_F16AX, 'f32': _lib.PREC_F32}
_PRECISION_NAMES = {_lib.PREC_F16: 'f16', _lib.PREC_BF16: 'bf16', _lib.PREC_FP8: 'fp8', _lib.PREC_F16X2: 'f16x2', _lib.PREC_F16L6: 'f16l6', _lib.PREC_F16A: 'f16a', _lib.PREC_F16AX: 'f16ax', _lib.PREC_F32: 'f32'}
# 'f16a' (round 5, pooling 'token' only): 'f16c' + the MLP's activation operands of the cls column tile as hi + lo terms -- what 'auto' escalates
# to there (the cls row is read without a mean over tokens: activation rounding, not weight rounding, is what is over budget)
_CALIBRATING = ('auto', 'f16c', 'fp8c', 'fp8w', 'f16a', 'f16ax')    # precisions whose handle is built with uitk_calibrate_weights
# 'f16ax' (round 5): 'f16a' on the split weights of 'f16x2' -- where 'auto' goes when an 'f16a' handle is over the budget
# 'f32' (round 5): the plain fp32 forward of the library (vector ALUs; ~20x slower) -- the last rung of every ladder: where 'auto' ends when no
# fp16 mode is inside the budget (checkpoints with strong attention sinks); any family, any pooling

__all__ = ["UITBase", "UitkSetupWarning", "UitkParityWarning", "uit_xs", "uit_xxs", "uit_xxxs", "PRETRAINED_CHECKPOINTS",
           "audio_transformer_h128_d4_m3", "audio_transformer_h128_d4_m3_relu",
           "audio_transformer_h128_d6_m3", "audio_transformer_h128_d6_m3_relu"]


# ----------------------------------------------------------------------------- parameter holders
def _melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') -- the buffer MelScale registers."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


class _Buffers(nn.Module):
    def __init__(self, **bufs):
        super().__init__()
        for k, v in bufs.items():
            self.register_buffer(k, v)


class _MelSpectrogramParams(nn.Module):
    """Holds `spectrogram.window` and `mel_scale.fb` under torchaudio's names (front_end.0.*)."""

    def __init__(self, n_fft, win_length, f_min, f_max, n_mels, sample_rate):
        super().__init__()
        self.spectrogram = _Buffers(window=torch.hann_window(win_length))
        self.mel_scale = _Buffers(fb=_melscale_fbanks(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate))


class _PatchEmbedParams(nn.Module):
    def __init__(self, input_size, patch_size, patch_stride, embed_dim):
        super().__init__()
        self.input_size = input_size
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (input_size[0] // patch_stride, input_size[1] // patch_stride)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(1, embed_dim, kernel_size=patch_size, stride=patch_stride)


class _AttnParams(nn.Module):
    """BNeckAttention (uit.py:89-122: inner = dim/4) or Attention (uit.py:124-178: inner = dim) parameter shapes."""

    def __init__(self, dim, qkv_bias, full=False):
        super().__init__()
        inner = dim if full else dim // 4
        self.qkv = nn.Linear(dim, inner * 3, bias=qkv_bias)
        self.proj = nn.Linear(inner, dim)


class _MlpParams(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _LayerScaleParams(nn.Module):
    """LayerScale (uit.py:78-85): x * gamma.  Holds the parameter only; the gain is folded into the preceding linear when the
    weights are handed to the C library (exact algebra)."""

    def __init__(self, dim, init_values):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class _BlockParams(nn.Module):
    def __init__(self, dim, hidden, qkv_bias, full_attention=False, init_values=None):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _AttnParams(dim, qkv_bias, full_attention)
        self.ls1 = _LayerScaleParams(dim, init_values) if init_values else nn.Identity()      # uit.py:230-231
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _MlpParams(dim, hidden)
        self.ls2 = _LayerScaleParams(dim, init_values) if init_values else nn.Identity()      # uit.py:240-241


class _Handle:
    """Owns one uitk_model* (per device).  `mode`: what the handle actually runs -- {'precision', 'requested', 'escalated', 'reason',
    'check'} (UITBase.calibration_report)."""

    def __init__(self, ptr, general=False):
        self.ptr = ptr
        self.mode = None
        self.general = general      # a uitk_general* (the general fp32 engine, csrc/general.hip) instead of a uitk_model*

    def __del__(self):
        try:
            if self.ptr:
                L = _lib.load()
                (L.uitk_general_destroy if self.general else L.uitk_destroy)(self.ptr)
        except Exception:
            pass


class _DevPtr:
    """Zero-copy torch view of a raw device pointer (amax slots) via __cuda_array_interface__."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class UitkSetupWarning(UserWarning):
    """Set-up work (weight calibration, front-end tuning) ran inside a forward call: see UITBase.prepare."""


class UitkParityWarning(UserWarning):
    """The parity mode escalated to split weights, or an explicit 'f16c' runs outside its budget: see UITBase.calibration_report."""


class UITBase(nn.Module):
    def __init__(self, outputdim=527, patch_size=16, patch_stride=16, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=True, drop_rate=0., attn_drop_rate=0., drop_path_rate=0., init_bn=True,
                 norm_layer=None, act_layer=None, init_values=None, target_length=1012, pooling='token',
                 wavtransforms=None, spectransforms=None, time_patch_out=None, freq_patch_out=None,
                 block_type='Block', attention_type='Attention', eval_avg='mean', precision=None, fe_autotune=True,
                 check_weights=True, fp32_fallback=True, **kwargs):
        super().__init__()
        assert pooling in ('mean', 'token', 'dm')
        self.outputdim = outputdim
        self.pooling = pooling
        self.embed_dim = embed_dim
        self.patch_stride = patch_stride
        self.patch_size = patch_size
        self.n_mels = kwargs.get('n_mels', 64)
        n_fft = kwargs.get('n_fft', 512)
        self.hop_size = kwargs.get('hop_size', 160)
        self.win_size = kwargs.get('win_size', 512)
        f_min = kwargs.get('f_min', 0)
        f_max = kwargs.get('f_max', 8000)
        self.center = kwargs.get('center', True)
        self.eval_avg = eval_avg
        self.time_patch_out = time_patch_out
        self.freq_patch_out = freq_patch_out
        self.target_length = target_length
        self.precision = precision
        self.fe_autotune = bool(fe_autotune)
        # every call compares a cheap version stamp of the weights (~15 us) and re-packs after an in-place edit; False: a caller that never
        # touches the weights after the first call (benchmark loops) skips the comparison (round 5: a constructor kwarg, no environment)
        self.check_weights = bool(check_weights)
        # precision 'auto': may the ladder end in 'f32' (the library's plain fp32 forward, ~20x slower than the MFMA encoder) when no fp16 mode
        # is inside the budget?  False: stay on the best fp16 build, warn, report within_budget = False (throughput over parity)
        self.fp32_fallback = bool(fp32_fallback)
        self.depth = depth
        self.num_heads = num_heads
        self.mlp_hidden = int(embed_dim * mlp_ratio)
        self.n_fft = n_fft
        # what the HIP kernels are built for: the uit_* family (uit.py:581-635: BNeckAttention + ReLU) and the other
        # factories of models/uit.py (:514-579: full-width Attention, ReLU or the default exact GELU); mean pooling
        unsupported = []

        if attention_type not in ('BNeckAttention', 'Attention'): unsupported.append("attention_type must be 'BNeckAttention' or 'Attention'")
        if act_layer is not None and act_layer not in (nn.ReLU, nn.GELU): unsupported.append("act_layer must be nn.ReLU or nn.GELU / None")
        self.attention_full = attention_type == 'Attention'
        self.act_gelu = act_layer is None or act_layer is nn.GELU
        # round 3: init_bn=False (the reference's GlobalNormer(-10, 20), uit.py:310-316), LayerScale (init_values, uit.py:230-243) and
        # qkv_bias=False are exact pack-time folds (see _host_weights); center=False is a front-end option (uitk_config.center)
        self.use_init_bn, self.init_values, self.qkv_bias = bool(init_bn), init_values, bool(qkv_bias)
        if not init_bn and self.n_mels != 64: unsupported.append("init_bn=False is defined for n_mels=64 only (as in the reference)")
        if patch_stride != patch_size: unsupported.append("patch_stride must equal patch_size (non-overlapping patches)")
        # The MFMA kernels are specialised to the geometry of the published factories.  Every OTHER geometry UITBase admits (n_mels, n_fft,
        # win_size, hop_size, patch_size, target_length, embed_dim, num_heads, mlp_ratio: uit.py:253-293) runs on the library's GENERAL
        # engine (round 6; csrc/general.hip): plain fp32 on the device in the reference's order of operations -- the general-geometry
        # sibling of precision 'f32': slower than the MFMA path by an order of magnitude, and right.  No calibration, tuning or options.
        self.general = not (embed_dim == 128 and num_heads == 2 and self.mlp_hidden == 384 and self.n_mels == 64 and n_fft == 512
                            and patch_size == 16 and target_length // 16 == 6 and 1 <= self.hop_size <= 512 and 1 <= self.win_size <= 512)
        if block_type != 'Block': unsupported.append("block_type must be 'Block'")
        if unsupported:
            raise NotImplementedError("uit_mobile_amd does not implement this UITBase configuration: " + "; ".join(unsupported))
        if precision is not None and precision not in _PRECISIONS:
            raise ValueError(f"Unknown precision ({precision})")
        if self.general:
            if precision not in (None, 'auto', 'f32'):
                raise NotImplementedError(f"precision='{precision}' is an MFMA-kernel mode, built for the geometry of the published factories (embed 128, 2 heads, "
                                          "MLP x 3, 64 mels, n_fft 512, 16 x 16 patches, 6 time patches); this configuration runs on the general fp32 "
                                          "engine: precision None / 'auto' / 'f32'")
            precision = 'f32'
            self.precision = precision
        if precision is None:
            # default = THE PARITY MODE, for every family and pooling (round 5: 'auto').  It starts as 'f16c' -- single fp16 weights with
            # CALIBRATED rounding: the kernels of 'f16', the weights rounded against operand statistics (uitk_calibrate_weights) of the
            # library's own calibration signals or of `model.calibrate(x)`: the accuracy of split weights at the plain fp16 rate (DESIGN.md
            # 2.1) -- and is HELD to the library's absolute budget on its held-out clips (option calib_budget_e6, 6e-4 on logits): a handle
            # whose calibration was refused or whose installed image is over budget is rebuilt in the family's split-weight precision
            # ('f16l6' for BNeckAttention + ReLU, 'f16x2' otherwise), with a warning and an entry in calibration_report().  An explicit
            # 'f16c' never escalates: it warns when its calibration is refused or over budget.
            precision = 'auto'
            self.precision = precision
        if precision not in _PRECISIONS:
            raise ValueError(f"Unknown precision ({precision})")
        if precision in ('fp8', 'fp8c') and (self.attention_full or self.act_gelu or pooling != 'mean'):
            raise NotImplementedError("precision='fp8' / 'fp8c' are built for the uit_* family (BNeckAttention + ReLU, mean pooling) only")
        if precision == 'f16l6' and (self.attention_full or self.act_gelu):
            raise NotImplementedError("precision='f16l6' is built for BNeckAttention + ReLU (any pooling)")
        if precision in ('f16a', 'f16ax') and pooling != 'token':
            raise NotImplementedError("precision='f16a' / 'f16ax' (split activation operands of the cls tile) are built for pooling='token'")
        if pooling != 'mean' and precision not in ('auto', 'f16', 'f16c', 'f16l6', 'f16x2', 'fp8w', 'f16a', 'f16ax', 'f32'):
            raise NotImplementedError("pooling='dm' / 'token' are built for precision='auto' / 'f16' / 'f16c' / 'f16l6' / 'f16x2' / 'fp8w'")

        self.front_end = nn.Sequential(
            _MelSpectrogramParams(n_fft, self.win_size, f_min, f_max, self.n_mels, 16000), nn.Identity())
        # init_bn=False: GlobalNormer has no parameters (no init_bn.* keys in the state_dict, as in the reference)
        self.init_bn = nn.Sequential(nn.Identity(), nn.BatchNorm2d(self.n_mels, momentum=0.01), nn.Identity()) if init_bn else nn.Identity()
        self.patch_embed = _PatchEmbedParams((self.n_mels, target_length), patch_size, patch_stride, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.token_pos_embed = nn.Parameter(torch.randn(1, embed_dim) * .02)
        self.time_pos_embed = nn.Parameter(torch.randn(1, embed_dim, 1, self.patch_embed.grid_size[1]) * .02)
        self.freq_pos_embed = nn.Parameter(torch.randn(1, embed_dim, self.patch_embed.grid_size[0], 1) * .02)
        self.blocks = nn.Sequential(*[_BlockParams(embed_dim, self.mlp_hidden, qkv_bias, self.attention_full, init_values) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.outputlayer = nn.Sequential(nn.LayerNorm(embed_dim), nn.Linear(embed_dim, outputdim))
        self.apply(self.init_weights)
        nn.init.normal_(self.cls_token, std=1e-6)
        self._handles = {}          # device index -> (_Handle, version stamp)
        self._options = {}          # uitk_set_option settings, re-applied to every handle this model creates
        self._calib_input = None    # calibrate(x): the caller's calibration clips (None = the library's own signals)
        self._tensors = None        # cached list of parameter / buffer tensors for the per-call stamp
        self._warned_setup = False  # UitkSetupWarning is given once per model
        self._warned_general = False

    def init_weights(self, module):
        if isinstance(module, nn.Linear):
            nn.init.trunc_normal_(module.weight, std=.02)
            if module.bias is not None:
                nn.init.zeros_(module.bias)
        elif isinstance(module, nn.LayerNorm):
            nn.init.constant_(module.bias, 0)
            nn.init.constant_(module.weight, 1.0)

    # ------------------------------------------------------------------ checkpoints (uit.py:416-450)
    def load_state_dict(self, state_dict, strict=True):
        if 'time_pos_embed' in state_dict and self.time_pos_embed.shape != state_dict['time_pos_embed'].shape:
            state_dict = dict(state_dict)
            self.change_pos_embedding(state_dict)
        out = super().load_state_dict(state_dict, strict=strict)
        self._handles.clear()
        return out

    def change_pos_embedding(self, state_dict):
        tt, tf = self.time_pos_embed.shape[-1], self.freq_pos_embed.shape[-2]
        pt, pf = state_dict['time_pos_embed'], state_dict['freq_pos_embed']
        if tt <= pt.shape[-1]:
            state_dict['time_pos_embed'] = pt[..., :tt]
        else:
            state_dict['time_pos_embed'] = torch.nn.functional.interpolate(pt, size=(1, tt), align_corners=False, mode='bilinear')
        if tf <= pf.shape[-2]:
            state_dict['freq_pos_embed'] = pf[:, :, :tf, :]
        else:
            state_dict['freq_pos_embed'] = torch.nn.functional.interpolate(pf, size=(tf, 1), align_corners=False, mode='bilinear')

    # ------------------------------------------------------------------ C-ABI plumbing
    def _stamp(self):
        """Identity of the weights the uploaded image was packed from, cheap enough for every call (~15 us: one pass over a cached
        tensor list): the sum of the tensors' version counters + the storage pointers of the first and last tensor.  In-place edits
        under no_grad (`p.mul_()`, `p.copy_()`, nn.init.*, vector_to_parameters, a sub-module's load_state_dict) bump a counter and are
        seen; `load_state_dict`, `_apply` (.to / .cuda / .half) and `refresh()` drop the images outright.  NOT seen: writes through
        `p.data` or a numpy view (they bypass the counter) -- call `refresh()` after those."""
        ts = self._tensors
        if ts is None:
            ts = self._tensors = list(self.parameters()) + list(self.buffers())
        return (sum(map(_VERSION_OF, ts)), ts[0].data_ptr(), ts[-1].data_ptr(), len(ts))

    def refresh(self):
        """Drop the uploaded parameter images: the next call re-packs (and re-calibrates) the current weights.  Needed only after
        editing a parameter through `p.data` / a numpy view; everything else is noticed by itself (see _stamp)."""
        self._handles.clear()
        self._tensors = None

    def calibrate(self, x=None):
        """Choose the calibration clips of precision 'f16c' (calibrated rounding, uitk_calibrate_weights): `x` = [B, T] device tensor
        (float32 / int16) of representative audio covering target_length frames, at most the first 1024 clips are used; None = the
        library's own synthetic signals (the default).  Takes effect when the parameter image is (re)built -- now, for live handles.
        Results stay within the parity tolerance either way (tests/test_gpu_parity.py::test_calibrated_rounding_*): calibration only
        decides in which direction each fp16 weight is rounded."""
        self._not_general("calibrate()")
        if self.precision not in _CALIBRATING:
            raise ValueError("calibrate() applies to the calibrated precisions 'auto' / 'f16c' / 'f16a' / 'f16ax' and 'fp8c' / 'fp8w'")
        if x is not None:
            x, _ = self._prep_input(x)
        self._calib_input = x
        self._handles.clear()
        return self

    def set_option(self, name, value):
        """Explicit tuning / test option of the C library (uitk_set_option: 'fe_asym_pm', 'fe_wg_per_cu', 'enc_waves', 'split_pe',
        'enc_generic_clamp', 'fe_tail_skip').  None changes a result bit.  Applied to the live handles and to every handle created later."""
        self._not_general("set_option()")
        if str(name) not in _OPTIONS:                    # (validated here too: without a live handle nothing else would reject it)
            raise ValueError(f"uitk_set_option: unknown option: {name}")
        for h, _ in self._handles.values():
            _lib.check(_lib.load().uitk_set_option(h.ptr, str(name).encode(), int(value)), "uitk_set_option")
        self._options[str(name)] = int(value)            # (remembered only once the live handles have accepted it)
        if str(name) in _CALIB_OPTIONS:                  # they decide what a calibration installs: the images are rebuilt under the new rule
            self._handles.clear()

    def get_option(self, name):
        """Current value of a library option on the live handle(s) (uitk_get_option) -- e.g. 'fe_asym_pm' after the first full-size call:
        the front-end pass split measured on this device.  {device: value}."""
        import ctypes
        self._not_general("get_option()")
        out = {}
        for dev, (h, _) in self._handles.items():
            v = ctypes.c_int(0)
            _lib.check(_lib.load().uitk_get_option(h.ptr, str(name).encode(), ctypes.byref(v)), "uitk_get_option")
            out[dev] = v.value
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if hasattr(self, '_handles'):
            self._handles.clear()
            self._tensors = None                          # .to() / .cuda() replace the tensors
        return out

    def _fn(self, name):
        """The C entry point `name` of the engine this model runs on: uitk_<name> (MFMA kernels) or uitk_general_<name>."""
        return getattr(_lib.load(), ("uitk_general_" if self.general else "uitk_") + name)

    def _not_general(self, what):
        if self.general:
            raise NotImplementedError(f"{what} belongs to the MFMA kernels (the published factories' geometry); this model runs on the general "
                                      "fp32 engine, which rounds nothing, calibrates nothing and has no options")

    def _config(self, precision=None):
        return _lib.Config(depth=self.depth, embed_dim=self.embed_dim, num_heads=self.num_heads,
                           mlp_hidden=self.mlp_hidden, outputdim=self.outputdim, target_length=self.target_length,
                           n_mels=self.n_mels, n_fft=self.n_fft, hop_size=self.hop_size, win_size=self.win_size,
                           patch_size=self.patch_size, center=int(bool(self.center)),
                           eval_avg={'mean': _lib.AVG_MEAN, 'max': _lib.AVG_MAX}.get(self.eval_avg, -1),
                           precision=_PRECISIONS[precision or self.precision], top_db=120.0,
                           attention_full=int(self.attention_full), act_gelu=int(self.act_gelu),
                           pooling={'mean': 0, 'dm': 1, 'token': 2}[self.pooling], fe_autotune=int(self.fe_autotune))

    def _host_weights(self):
        """-> (Weights struct, keep-alive list) with fp32 host pointers in reference layout."""
        sd = self.state_dict()
        keep = []

        def arr(a):
            a = np.ascontiguousarray(a, dtype=np.float32)
            keep.append(a)
            return a.ctypes.data

        def ptr(key):
            return arr(sd[key].detach().to('cpu', torch.float32).numpy())

        def f64(key):
            return sd[key].detach().to('cpu', torch.float64).numpy()

        def block_ptr(b, f, k):
            """per-block tensor with the kwargs folds: qkv_bias=False -> zero bias; LayerScale gamma -> rows of proj / fc2 and their biases"""
            key = f"blocks.{b}.{k}"
            if f == "qkv_b" and not self.qkv_bias:
                return arr(np.zeros(sd[f"blocks.{b}.attn.qkv.weight"].shape[0]))
            ls = {"proj_w": "ls1", "proj_b": "ls1", "fc2_w": "ls2", "fc2_b": "ls2"}.get(f)
            if ls is not None and self.init_values:
                g = f64(f"blocks.{b}.{ls}.gamma")
                return arr(f64(key) * (g[:, None] if f.endswith("_w") else g))
            return ptr(key)

        blocks = (_lib.BlockWeights * self.depth)()
        names = {"norm1_w": "norm1.weight", "norm1_b": "norm1.bias", "qkv_w": "attn.qkv.weight", "qkv_b": "attn.qkv.bias",
                 "proj_w": "attn.proj.weight", "proj_b": "attn.proj.bias", "norm2_w": "norm2.weight",
                 "norm2_b": "norm2.bias", "fc1_w": "mlp.fc1.weight", "fc1_b": "mlp.fc1.bias",
                 "fc2_w": "mlp.fc2.weight", "fc2_b": "mlp.fc2.bias"}
        for b in range(self.depth):
            for f, k in names.items():
                setattr(blocks[b], f, block_ptr(b, f, k))
        keep.append(blocks)
        if self.use_init_bn:
            bn = dict(bn_weight=ptr("init_bn.1.weight"), bn_bias=ptr("init_bn.1.bias"),
                      bn_mean=ptr("init_bn.1.running_mean"), bn_var=ptr("init_bn.1.running_var"))
        else:       # GlobalNormer(-10, 20): (x + 10) / (2 * 20)  ==  eval BatchNorm with weight 1, bias 0, mean -10, var + eps = 1600
            M = self.n_mels
            bn = dict(bn_weight=arr(np.ones(M)), bn_bias=arr(np.zeros(M)), bn_mean=arr(np.full(M, -10.0)),
                      bn_var=arr(np.full(M, 1600.0 - 1e-5)))
        w = _lib.Weights(window=ptr("front_end.0.spectrogram.window"), mel_fb=ptr("front_end.0.mel_scale.fb"),
                         **bn,
                         pe_w=ptr("patch_embed.proj.weight"), pe_b=ptr("patch_embed.proj.bias"),
                         time_pos=ptr("time_pos_embed"), freq_pos=ptr("freq_pos_embed"),
                         norm_w=ptr("norm.weight"), norm_b=ptr("norm.bias"),
                         head_ln_w=ptr("outputlayer.0.weight"), head_ln_b=ptr("outputlayer.0.bias"),
                         head_w=ptr("outputlayer.1.weight"), head_b=ptr("outputlayer.1.bias"), blocks=blocks,
                         cls_token=ptr("cls_token"), token_pos=ptr("token_pos_embed"))
        return w, keep

    def packed_image(self, calib_stats=None):
        """Host-only: the packed parameter image uitk_create would upload (bytes).  No GPU needed.  calib_stats: float64 array of
        uitk_calib_stats_doubles values -> the image uitk_calibrate_weights would build from those statistics (uitk_pack_weights_calibrated)."""
        self._not_general("packed_image()")
        L = _lib.load()
        cfg = self._config()
        w, keep = self._host_weights()
        need = ctypes.c_size_t()
        if calib_stats is None:
            pack = lambda out, cap: L.uitk_pack_weights(ctypes.byref(cfg), ctypes.byref(w), out, cap, ctypes.byref(need))
        else:
            st = np.ascontiguousarray(calib_stats, dtype=np.float64)
            if st.size != L.uitk_calib_stats_doubles(ctypes.byref(cfg)):
                raise ValueError(f"calib_stats must hold {L.uitk_calib_stats_doubles(ctypes.byref(cfg))} doubles (include/uitk.h), got {st.size}")
            pack = lambda out, cap: L.uitk_pack_weights_calibrated(ctypes.byref(cfg), ctypes.byref(w), ctypes.c_void_p(st.ctypes.data), out, cap, ctypes.byref(need))
        _lib.check(pack(None, 0), "uitk_pack_weights")
        buf = (ctypes.c_char * need.value)()
        _lib.check(pack(buf, need.value), "uitk_pack_weights")
        return bytes(buf)

    def calib_stats(self, device=None):
        """The operand statistics the live handle's calibration measured (float64, layout of include/uitk.h: uitk_calib_stats_doubles)."""
        self._not_general("calib_stats()")
        L = _lib.load()
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        h = self._handle(device)
        cfg = self._config()
        out = np.empty(L.uitk_calib_stats_doubles(ctypes.byref(cfg)), np.float64)
        _lib.check(L.uitk_get_calib_stats(h, ctypes.c_void_p(out.ctypes.data), out.size), "uitk_get_calib_stats")
        return out

    def calibration_report(self, device=None):
        """What the live handle runs and what the acceptance check of its calibration measured on the library's held-out clips
        (uitk_get_calib_check): {'max_nearest', 'max_calibrated', 'rms_nearest', 'rms_calibrated', 'accepted'} -- deviations of the
        pre-sigmoid logits (probabilities for pooling 'dm') from the unrounded network, for the nearest-rounded and the calibrated
        parameter image; accepted False = the calibrated image was refused -- plus (round 5) 'precision' (what the handle actually runs:
        'f16c', 'f16' after a refusal inside the budget, or the split-weight precision 'auto' escalated to), 'requested', 'escalated',
        'reason', 'budget' (the absolute budget the installed image was held to) and, after an escalation, 'self_check' = (max, rms) of
        the rebuilt handle on the same clips (uitk_self_check).  None when no check has run (precisions without calibration, or
        calib_accept_pct = 0)."""
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        mode = self._handle_obj(device).mode
        if mode.get("check") is None:
            return None
        return {**mode["check"], **{k: v for k, v in mode.items() if k != "check"}}

    def effective_precision(self, device=None):
        """The precision the live handle on `device` actually runs ('auto' resolves to 'f16c', or to 'f16l6' / 'f16x2' after an escalation)."""
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        return self._handle_obj(device).mode["precision"]

    def self_check(self, device=None):
        """(max, rms) deviation of the live handle -- ANY precision -- on the library's held-out clips against the fp32 forward of the
        unrounded network (uitk_self_check): the oracle-free estimate of this handle's parity margin.  A set-up operation (allocates,
        synchronises); None when no single-crop clip length covers target_length."""
        self._not_general("self_check()")
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        h = self._handle_obj(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        w, keep = self._host_weights()
        out = (ctypes.c_float * 2)()
        with torch.cuda.device(idx):
            stream = ctypes.c_void_p(torch.cuda.current_stream(idx).cuda_stream)
            _lib.check(_lib.load().uitk_self_check(h.ptr, ctypes.byref(w), stream, out), "uitk_self_check")
        return None if out[0] < 0 else (float(out[0]), float(out[1]))

    def _create(self, L, w, precision):
        """uitk_create on the current device in `precision` + this model's options -> _Handle (owned: a failure destroys it)."""
        cfg = self._config(precision)
        out = ctypes.c_void_p()
        if self.general:
            _lib.check(L.uitk_general_create(ctypes.byref(cfg), ctypes.byref(w), ctypes.byref(out)), "uitk_general_create")
            handle = _Handle(out.value, general=True)
            handle.mode = {"precision": "f32", "requested": self.precision, "escalated": False, "reason": None, "check": None, "engine": "general"}
            return handle
        _lib.check(L.uitk_create(ctypes.byref(cfg), ctypes.byref(w), ctypes.byref(out)), "uitk_create")
        handle = _Handle(out.value)
        for name, value in self._options.items():
            _lib.check(L.uitk_set_option(out, name.encode(), value), "uitk_set_option")
        handle.mode = {"precision": precision, "requested": self.precision, "escalated": False, "reason": None, "check": None}
        return handle

    def _handle(self, device):
        return self._handle_obj(device).ptr

    @staticmethod
    def _calibrate_handle(L, handle, w, x, stream):
        """uitk_calibrate_weights on `handle`: the caller's clips `x` (device tensor) or, None, the library's own signals."""
        if x is None:
            _lib.check(L.uitk_calibrate_weights(handle.ptr, ctypes.byref(w), None, _lib.IN_I16, 0, 0, stream), "uitk_calibrate_weights")
        else:
            _lib.check(L.uitk_calibrate_weights(handle.ptr, ctypes.byref(w), ctypes.c_void_p(x.data_ptr()),
                                                _lib.IN_I16 if x.dtype == torch.int16 else _lib.IN_F32, x.shape[0], x.shape[1], stream),
                       "uitk_calibrate_weights")

    def _handle_obj(self, device, _from_forward=False):
        """The live _Handle for `device`: the cached one while the weights' stamp is unchanged, else a new build (_build_handle)."""
        if self.eval_avg not in ('mean', 'max'):
            raise ValueError(f'Unknown Eval average function ({self.eval_avg})')      # uit.py:488
        idx = device.index if device.index is not None else torch.cuda.current_device()
        h = self._handles.get(idx)
        stamp = self._stamp() if (self.check_weights or h is None) else None
        if h is not None and (stamp is None or h[1] == stamp):
            return h[0]
        if self.general and not self._warned_general:
            self._warned_general = True
            warnings.warn("uit_mobile_amd: this configuration is outside the geometry of the published factories (embed 128, 2 heads, MLP x 3, 64 mels, "
                          "n_fft 512, 16 x 16 patches, 6 time patches): it runs on the library's GENERAL fp32 engine -- on the device, in the reference's "
                          "order of operations, an order of magnitude slower than the MFMA kernels", UitkSetupWarning, stacklevel=4)
        if self.precision in _CALIBRATING and _from_forward and not self._warned_setup:
            self._warned_setup = True
            warnings.warn("uit_mobile_amd: this forward call builds the parameter image first (weight calibration + acceptance check, "
                          "~0.2 s, device-synchronous; front-end tuning on the first full-size batch): call model.prepare(device, (B, T)) "
                          "once at set-up to keep that out of the first forward (required before stream capture)", UitkSetupWarning, stacklevel=4)
        handle = self._build_handle(idx)
        self._handles[idx] = (handle, stamp)
        return handle

    def _build_handle(self, idx):
        """uitk_create (+ uitk_calibrate_weights and the parity ladder for the calibrated precisions) on device `idx` -> _Handle."""
        L = _lib.load()
        w, keep = self._host_weights()
        with torch.cuda.device(idx):
            # 'auto' starts as 'f16c' -- except for pooling 'token', whose parity mode is 'f16a' from the start: the head reads ONE row, the
            # MLP's activation rounding on that row is not averaged over 24 tokens, and the library's held-out clips cannot bound it
            # for every input (tools/auto_sweep.py, 24 unseen checkpoints of uit_xxs: 'f16c' inside the library's budget and still 1.08e-3
            # on one of the reference's recordings; 'f16a': profiles/round5/auto_sweep_token_f16a_v2.log)
            start_as = self.precision if self.precision != 'auto' else ('f16a' if self.pooling == 'token' else 'f16c')
            handle = self._create(L, w, start_as)
            if self.precision in _CALIBRATING:            # calibrated rounding: the caller's clips, else the library's own signals
                x = self._calib_input if self._calib_input is None else self._calib_input.to(torch.device("cuda", idx))
                stream = ctypes.c_void_p(torch.cuda.current_stream(idx).cuda_stream)
                self._calibrate_handle(L, handle, w, x, stream)
                handle = self._hold_to_budget(L, w, handle, start_as, x, stream, idx)
        return handle

    def _hold_to_budget(self, L, w, handle, start_as, x, stream, idx):
        """The guarantee of the parity mode (round 5), after uitk_calibrate_weights on `handle`: read what the library measured on its
        held-out clips and what it asks for (option parity_escalate_to); 'auto' climbs the ladder the library names -- 'f16l6' / 'f16x2';
        pooling 'token': 'f16a', then 'f16ax'; last, for every family: 'f32', the plain fp32 forward -- calibrating each fp16 build (operand
        centring), measuring it on the same clips (uitk_self_check), taking it when it is better, until one is inside the budget; an
        explicit 'f16c' / 'f16a' warns and stays.  Returns the _Handle to install; its .mode is the report."""
        rep = (ctypes.c_float * 5)()
        _lib.check(L.uitk_get_calib_check(handle.ptr, rep), "uitk_get_calib_check")
        esc, budget = ctypes.c_int(-1), ctypes.c_int(0)
        _lib.check(L.uitk_get_option(handle.ptr, b"parity_escalate_to", ctypes.byref(esc)), "uitk_get_option")
        _lib.check(L.uitk_get_option(handle.ptr, b"calib_budget_e6", ctypes.byref(budget)), "uitk_get_option")
        check = None if rep[4] < 0 else {"max_nearest": rep[0], "max_calibrated": rep[1], "rms_nearest": rep[2], "rms_calibrated": rep[3],
                                         "accepted": rep[4] > 0}
        handle.mode.update(check=check, budget=budget.value * 1e-6 * (0.25 if self.pooling == 'dm' else 1.0), within_budget=esc.value < 0)
        refused = check is not None and not check["accepted"]
        if refused:                                       # the nearest-rounded image stayed
            handle.mode["precision"] = {'f16c': 'f16', 'fp8c': 'fp8'}.get(start_as, start_as + ' (nearest on its grid)')
        if not (esc.value >= 0 or refused):
            return handle
        dev_now = None if check is None else (check["max_nearest"] if refused else check["max_calibrated"])
        if self.precision not in ('auto', 'f16c', 'f16a'):
            # any other explicit calibrating precision ('f16ax', 'fp8c', 'fp8w'): no ladder, but never silent -- a refusal leaves
            # the nearest image on its grid, and a parity precision over its budget ('f16ax': the library names the next rung) says so
            if esc.value >= 0 or refused:
                what = ("the acceptance check refused the calibrated image (calibrated %.2e vs nearest %.2e on the held-out clips): the handle runs the "
                        "nearest-rounded image" % (check["max_calibrated"], check["max_nearest"])) if refused else \
                       ("held-out deviation %s over the budget of %.1e" % ("n/a" if dev_now is None else "%.2e" % dev_now, handle.mode["budget"]))
                handle.mode["reason"] = what
                warnings.warn(f"uit_mobile_amd: precision='{self.precision}' on cuda:{idx}: {what}; an explicit '{self.precision}' does not escalate"
                              + (f" (the library names '{_PRECISION_NAMES[esc.value]}' as the next rung)" if esc.value >= 0 else "")
                              + " -- use precision='auto' (the default) for the guaranteed parity mode", UitkParityWarning, stacklevel=6)
            return handle
        why = ("calibration refused by the acceptance check (calibrated %.2e vs nearest %.2e on the held-out clips)" % (check["max_calibrated"], check["max_nearest"])
               if refused else "held-out deviation of the calibrated image over the budget") if check is not None else "the held-out check failed"
        if dev_now is not None:
            why += "; installed image: %.2e against a budget of %.1e" % (dev_now, handle.mode["budget"])
        handle.mode["reason"] = why
        if self.precision != 'auto':
            warnings.warn(f"uit_mobile_amd: precision='{self.precision}' on cuda:{idx}: {why}; an explicit '{self.precision}' does not escalate"
                          + (" and now runs the NEAREST-rounded fp16 image (= precision 'f16')" if refused and self.precision == 'f16c' else "")
                          + " -- use precision='auto' (the default) for the guaranteed parity mode", UitkParityWarning, stacklevel=6)
            return handle
        if esc.value < 0:                                 # refused, yet inside the budget (harmless at init scale): reported, not escalated
            return handle
        # the ladder: build what the library names, of the same weights -- the SAME calibration call gives a split build the operand centring
        # of the attention (its weights are not re-rounded) and the library's verdict on it (the next rung, if it is over budget too) --
        # measure it on the same clips, keep it if it is better, go on while it is over the budget and the library names another rung
        cur, cur_dev, to_code, path = handle, dev_now, esc.value, [handle.mode["precision"]]
        while to_code >= 0:
            to = _PRECISION_NAMES[to_code]
            nxt = self._create(L, w, to)
            nxt_code = -1
            if to != 'f32':                               # (the fp32 rung has nothing to calibrate)
                self._calibrate_handle(L, nxt, w, x, stream)
                e2 = ctypes.c_int(-1)
                _lib.check(L.uitk_get_option(nxt.ptr, b"parity_escalate_to", ctypes.byref(e2)), "uitk_get_option")
                nxt_code = e2.value
            out2 = (ctypes.c_float * 2)()
            _lib.check(L.uitk_self_check(nxt.ptr, ctypes.byref(w), stream, out2), "uitk_self_check")
            sc = None if out2[0] < 0 else (float(out2[0]), float(out2[1]))      # (a negative value = "the check does not apply"; NaN is a measurement)
            if sc is not None and not math.isfinite(sc[0]):
                # a rung that answers NaN / Inf on a held-out clip is WORSE than anything finite (`nan >= cur_dev` is False, so
                # the rung used to be taken): never installed, reported, and the ladder goes on
                why += "; the build '%s' is not finite on the held-out clips (not taken)" % to
                cur.mode.setdefault("self_check_split", sc)
            elif sc is not None and cur_dev is not None and math.isfinite(cur_dev) and sc[0] >= cur_dev and to != 'f32':
                # no better: what is over budget is not what that build splits -- it is not taken (the next rung still is tried)
                why += "; the build '%s' measures %.2e there (no better)" % (to, sc[0])
                cur.mode.setdefault("self_check_split", sc)
            else:
                nxt.mode.update(check=handle.mode["check"], budget=handle.mode["budget"], escalated=True, self_check=sc)
                cur, cur_dev = nxt, (sc[0] if sc is not None else None)
                path.append(to)
            if cur_dev is not None and cur_dev <= handle.mode["budget"]:
                break
            to_code = nxt_code if nxt_code >= 0 else (_lib.PREC_F32 if to != 'f32' else -1)
            if to_code == _lib.PREC_F32 and not self.fp32_fallback:
                why += "; the fp32 rung is switched off (fp32_fallback=False)"
                break
        inside = cur_dev is None or (math.isfinite(cur_dev) and cur_dev <= handle.mode["budget"])
        cur.mode.update(reason=why, within_budget=inside, ladder=path)
        if cur is handle:
            warnings.warn(f"uit_mobile_amd: precision='auto' on cuda:{idx}: {why}: kept", UitkParityWarning, stacklevel=6)
        else:
            to = path[-1]
            warnings.warn(f"uit_mobile_amd: precision='auto' on cuda:{idx}: {why} -> rebuilding the handle in precision '{to}' "
                          + ("(the plain fp32 forward: the last rung, ~20x slower than the MFMA encoder -- and right)" if to == 'f32'
                             else "(split activation operands + split weights: the top of pooling 'token''s fp16 ladder, slower)" if to == 'f16ax'
                             else "(split activation operands of the cls tile, slower)" if to == 'f16a' else "(split weights: parity-grade, slower)")
                          + ("" if inside else "; that build measures %.2e there: STILL OVER the budget" % cur_dev)
                          + " [ladder: " + " -> ".join(path) + "]", UitkParityWarning, stacklevel=6)
        return cur                                        # (the handles not taken are destroyed when their last reference goes)

    @staticmethod
    def _prep_input(x):
        if not isinstance(x, torch.Tensor):
            raise TypeError("expected a torch.Tensor [B, T]")
        if not x.is_cuda:
            raise RuntimeError("uit_mobile_amd runs on a HIP device only (no CPU path): move the input with "
                               ".to('cuda') as evaluate.py does via transfer_to_device")
        if x.dim() == 1:
            x = x.unsqueeze(0)
        if x.dim() != 2:
            raise ValueError("expected input of shape [B, T]")
        if x.dtype == torch.int16:
            return x.contiguous(), _lib.IN_I16
        return x.to(torch.float32).contiguous(), _lib.IN_F32

    def num_frames(self, T):
        if not self.center:                                  # torch.stft without centring: no padding
            return 1 + (T - self.n_fft) // self.hop_size
        return 1 + T // self.hop_size

    def num_crops(self, T):
        nf = self.num_frames(T)
        return 1 if nf <= self.target_length else -(-nf // self.target_length)

    # ------------------------------------------------------------------ forward (uit.py:452-493, eval branch)
    def forward(self, x, mixup=None, return_logits=False, amax_sync=None):
        """probs[B, outputdim] = model(wave[B, T]).

        return_logits: also return the pre-sigmoid outputs [B, ncrops, outputdim].
        amax_sync: optional callable(tensor[64]) run between the two kernels on the batch-amax slots (used by the
        data-parallel wrapper to MAX-all-reduce the top_db reference across shards)."""
        if self.training:
            raise RuntimeError("uit_mobile_amd implements inference only: call model.eval() first")
        if amax_sync is not None:
            st = self.frontend(x)
            amax_sync(self.amax_tensor(st[2]))
            return self.encode(st, return_logits=return_logits)
        x, in_dtype = self._prep_input(x)
        B, T = x.shape
        L = _lib.load()
        h = self._handle_obj(x.device, _from_forward=True).ptr
        with torch.cuda.device(x.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            probs = torch.empty((B, self.outputdim), device=x.device, dtype=torch.float32)
            logits = torch.empty((B, self.num_crops(T), self.outputdim), device=x.device, dtype=torch.float32) if return_logits else None
            lp = ctypes.c_void_p(logits.data_ptr()) if return_logits else None
            _lib.check(self._fn("forward")(h, ctypes.c_void_p(x.data_ptr()), in_dtype, B, T,
                                           ctypes.c_void_p(probs.data_ptr()), lp, stream), "uitk_forward")
        return (probs, logits) if return_logits else probs

    def prepare(self, device="cuda", batch_shape=None, dtype=torch.float32, sample=None):
        """THE set-up call (round 5): everything a first forward would otherwise do on the caller's time, now -- build the parameter
        image on `device` (for 'auto' / 'f16c': weight calibration, acceptance check, a possible escalation; ~0.15 s, device-synchronous),
        and, given batch_shape = (B, T): reserve the workspace for it, measure the front-end's pass split for that shape
        (uitk_tune_frontend, on a scratch buffer of `dtype` float32 / int16) and run one warm-up forward, so that every later forward of
        that shape is enqueue-only -- and may be captured into a HIP graph.  `sample`: a representative [B, T] device batch to tune and
        warm up on instead of the scratch noise (its shape and dtype then stand for batch_shape / dtype): the split is a property of the
        device and the pass count, but the leftover-frame skip makes a pass's cost mildly data-dependent, and a caller that has its real
        data at hand gets the split measured on it.  Stands in for `.to(DEVICE).eval()` + the first batch of inference.py:39-55 /
        evaluate.py:34-65.  Returns self."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("uit_mobile_amd runs on a HIP device only (no CPU path)")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if self.training:
            raise RuntimeError("uit_mobile_amd implements inference only: call model.eval() first")
        self._warned_setup = True                         # the caller did the right thing
        if sample is not None:
            sample, _ = self._prep_input(sample)
            batch_shape = tuple(sample.shape)
        if batch_shape is not None and self.precision in _CALIBRATING and 'calib_short_check' not in self._options:
            # round 6: a model prepared for SHORT clips (fewer than the six time patches of a full crop) is judged on short clips too --
            # the held-out check then scores its clips at half length as well (option calib_short_check): with fewer tokens
            # to average over, a checkpoint can be inside the budget on 1 s clips and at the gate on 0.5 s ones
            if self.num_frames(int(batch_shape[1])) < 16 * self.patch_embed.grid_size[1]:
                self.set_option('calib_short_check', 1)
        self._handle_obj(device)
        if batch_shape is not None:
            B, T = int(batch_shape[0]), int(batch_shape[1])
            self.reserve(B, T, device)
            with torch.cuda.device(device):
                if sample is not None:
                    x = sample.to(device)
                else:
                    g = torch.Generator(device=device).manual_seed(1)
                    x = torch.randn((B, T), device=device, generator=g) * 0.1    # (noise keeps every frame live)
                    if dtype == torch.int16:
                        x = (x * 32768.0).round().clamp(-32768, 32767).to(torch.int16)
                if self.fe_autotune:
                    self.tune_frontend(x)
                with torch.no_grad():
                    self.forward(x)
                torch.cuda.synchronize(device)
        return self

    def reserve(self, B, T, device="cuda"):
        """Allocate the workspace for batches of up to B clips of T samples now (uitk_reserve) instead of inside the first
        forward call; later calls of that size are enqueue-only."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("uit_mobile_amd runs on a HIP device only (no CPU path)")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(device):
            _lib.check(self._fn("reserve")(self._handle(device), int(B), int(T)), "uitk_reserve")

    def tune_frontend(self, x):
        """Measure the front-end's pass split for batches shaped like `x` on x's device now (uitk_tune_frontend: ~140 short launches
        on x's buffer and the current stream, event waits only).  The result is shared by every model of the process on that device;
        with fe_autotune=True (default) the first forward of an unmeasured shape does this by itself."""
        if self.general:
            return                                        # (nothing to tune)
        x, in_dtype = self._prep_input(x)
        B, T = x.shape
        h = self._handle(x.device)
        with torch.cuda.device(x.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            _lib.check(_lib.load().uitk_tune_frontend(h, ctypes.c_void_p(x.data_ptr()), in_dtype, B, T, stream), "uitk_tune_frontend")

    # the two stages of forward(), for callers that put work between them (the amax all-reduce of the data-parallel
    # wrapper; a pipelined caller that runs the next batch's front-end on a second replica before this batch's encoder)
    def frontend(self, x):
        """Stage 1 (uitk_frontend): log-mel operands of x into this model's workspace + the batch-amax slots.
        Returns the state `encode` needs.  One batch per model instance may be between its two stages."""
        if self.training:
            raise RuntimeError("uit_mobile_amd implements inference only: call model.eval() first")
        x, in_dtype = self._prep_input(x)
        B, T = x.shape
        h = self._handle_obj(x.device, _from_forward=True)
        with torch.cuda.device(x.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            _lib.check(self._fn("frontend")(h.ptr, ctypes.c_void_p(x.data_ptr()), in_dtype, B, T, stream), "uitk_frontend")
        # the _Handle OBJECT rides along: it keeps the uitk_model alive until `encode` has used it, and `encode` need not look
        # the weights over a second time -- it only checks that this is still the model's live handle
        return (B, T, x.device, h)

    def encode(self, state, return_logits=False):
        """Stage 2 (uitk_encode) on the operands of the latest `frontend` call of this instance."""
        B, T, device = state[:3]
        h = state[3] if len(state) > 3 else self._handle_obj(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        live = self._handles.get(idx)
        if live is None or live[0] is not h:
            raise RuntimeError("encode(): the model's parameter image was rebuilt (refresh / calibrate / load_state_dict / .to / set_option / a "
                               "weight edit seen by another call) since the frontend() call this state came from: run frontend() again")
        h = h.ptr
        with torch.cuda.device(device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            probs = torch.empty((B, self.outputdim), device=device, dtype=torch.float32)
            logits = torch.empty((B, self.num_crops(T), self.outputdim), device=device, dtype=torch.float32) if return_logits else None
            lp = ctypes.c_void_p(logits.data_ptr()) if return_logits else None
            _lib.check(self._fn("encode")(h, B, T, ctypes.c_void_p(probs.data_ptr()), lp, stream), "uitk_encode")
        return (probs, logits) if return_logits else probs

    def amax_tensor(self, device):
        h = self._handle(device)
        ptr = self._fn("amax_ptr")(h)
        return torch.as_tensor(_DevPtr(ptr, _lib.AMAX_SLOTS), device=device)

    # ------------------------------------------------------------------ stage outputs for parity tests
    def melspec_db(self, x):
        """What self.front_end(x) returns in the reference: dB mel [B, n_mels, frames], batch-global floor applied."""
        x, in_dtype = self._prep_input(x)
        B, T = x.shape
        L = _lib.load()
        h = self._handle(x.device)
        with torch.cuda.device(x.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            out = torch.empty((B, self.n_mels, self.num_frames(T)), device=x.device, dtype=torch.float32)
            _lib.check(self._fn("mel_db")(h, ctypes.c_void_p(x.data_ptr()), in_dtype, B, T,
                                          ctypes.c_void_p(out.data_ptr()), stream), "uitk_mel_db")
        return out

    def precision_margin(self, x, gate=1e-3):
        """How much of this checkpoint's fp16 error is weight rounding?  Runs `x` through precision 'f16' (single fp16 weights) and
        'f16l6' (the parity mode: the same operands plus the low-order term of the fc1 / fc2 / proj weights) and returns the largest
        logit difference: that difference IS the weight-rounding term plain fp16 would carry -- most of its error at trained-checkpoint
        magnitudes.  'f16' is recommended (it is 1.25x faster) only while the term stays under 0.3 * gate.  No
        reference, no CPU: two forward passes on the device."""
        self._not_general("precision_margin()")
        if self.precision not in ('auto', 'f16', 'f16c', 'f16x2', 'f16l6'):
            raise ValueError("precision_margin compares the fp16 precisions 'f16' and the family's parity mode")
        # the parity mode of this family: f16l6 (BNeckAttention + ReLU), f16x2 (full-width attention / GELU, mean pooling)
        par = 'f16l6' if not (self.attention_full or self.act_gelu) else 'f16x2'
        keep, out = self.precision, {}
        try:
            for prec in ('f16', par):
                self.precision = prec
                self.refresh()
                with torch.no_grad():
                    out[prec] = self.forward(x, return_logits=True)[1].float().clone()
        finally:
            self.precision = keep
            self.refresh()
        d = float((out['f16'] - out[par]).abs().max())
        return {'max_abs_logit_diff': d, 'max_abs_logit': float(out[par].abs().max()),
                'recommended': par if d > 0.3 * gate else 'f16'}

    def debug_tokens(self, x):
        """(probs, tokens[B, ncrops, depth+1, 24, 128]): residual stream after patch-embed and after each block."""
        self._not_general("debug_tokens()")
        x, _ = self._prep_input(x)
        B, T = x.shape
        L = _lib.load()
        h = self._handle(x.device)
        nc = self.num_crops(T)
        buf = torch.zeros((B, nc, self.depth + 1, 24, self.embed_dim), device=x.device, dtype=torch.float32)
        _lib.check(L.uitk_set_debug_tokens(h, ctypes.c_void_p(buf.data_ptr())), "uitk_set_debug_tokens")
        try:
            probs = self.forward(x)
            torch.cuda.synchronize(x.device)
        finally:
            L.uitk_set_debug_tokens(h, None)
        return probs, buf


def _family(depth, **kwargs):
    model_kwargs = dict(patch_size=16, embed_dim=128, depth=depth, num_heads=2, mlp_ratio=3.0, pooling='mean',
                        init_bn=True, drop_path_rate=0.0, act_layer=nn.ReLU, attention_type='BNeckAttention')
    model_kwargs = {**model_kwargs, **kwargs}
    return UITBase(**model_kwargs)


def _h128(depth, act_layer, **kwargs):
    """The non-bottleneck factories (uit.py:514-579): full-width Attention (the UITBase default attention_type), mean pooling."""
    model_kwargs = dict(patch_size=16, embed_dim=128, depth=depth, num_heads=2, mlp_ratio=3.0, pooling='mean',
                        init_bn=True, drop_path_rate=0.0)
    if act_layer is not None:
        model_kwargs['act_layer'] = act_layer
    return UITBase(**{**model_kwargs, **kwargs})


def audio_transformer_h128_d4_m3_relu(**kwargs):    # uit.py:514-529
    return _h128(4, nn.ReLU, **kwargs)


def audio_transformer_h128_d4_m3(**kwargs):         # uit.py:532-546  (act_layer default: nn.GELU, uit.py:344)
    return _h128(4, None, **kwargs)


def audio_transformer_h128_d6_m3(**kwargs):         # uit.py:548-562
    return _h128(6, None, **kwargs)


def audio_transformer_h128_d6_m3_relu(**kwargs):    # uit.py:564-579
    return _h128(6, nn.ReLU, **kwargs)


def uit_xs(**kwargs):      # uit.py:581-597
    return _family(12, **kwargs)


def uit_xxs(**kwargs):     # uit.py:619-635
    return _family(6, **kwargs)


def uit_xxxs(**kwargs):    # uit.py:600-616
    return _family(4, **kwargs)


PRETRAINED_CHECKPOINTS = {   # uit.py:639-655
    'uit_xs': {'model': uit_xs, 'model_kwargs': dict(outputdim=537, target_length=102),
               'chkpt': 'https://zenodo.org/record/7690036/files/uit_xs_mAP3409.pt?download=1'},
    'uit_xxs': {'model': uit_xxs, 'model_kwargs': dict(outputdim=537, target_length=102),
                'chkpt': 'https://zenodo.org/record/7690036/files/uit_xxs_mAP3221.pt?download=1'},
    'uit_xxxs': {'model': uit_xxxs, 'model_kwargs': dict(outputdim=537, target_length=102),
                 'chkpt': 'https://zenodo.org/record/7690036/files/uit_xxxs_mAP3097.pt?download=1'},
}
