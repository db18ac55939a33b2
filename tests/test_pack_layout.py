"""CPU: the packed parameter image (csrc/pack.cpp via uitk_pack_weights) + the kernel's index algebra, executed by
the numpy MFMA-layout emulator (tests/mfma_emu.py), must reproduce the oracle.  No GPU, no compute call into
libuitk (uitk_pack_weights is host-only)."""
import numpy as np
import pytest
import torch

from mfma_emu import EncoderEmu
from oracle.oracle import Oracle
from synth_weights import make_state_dict, synth_pcm_i16
from uit_mobile_amd import models


def _model(name, seed, prec, **kw):
    sd = make_state_dict(name, seed)
    m = getattr(models, name)(outputdim=537, target_length=102, precision=prec, **kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.eval(), sd


@pytest.mark.parametrize("prec,tol_tok,tol_logit", [("f16", 6e-3, 1e-3), ("bf16", 5e-2, 8e-3)])
def test_emulated_kernel_matches_oracle(prec, tol_tok, tol_logit):
    m, sd = _model("uit_xxxs", 11, prec)
    emu = EncoderEmu(m.packed_image())
    pcm = synth_pcm_i16(2, 16000, seed=7)
    o = Oracle(sd)
    ref = o.forward(pcm, want_stages=True)
    mel_raw, amax = o.melspec_db(pcm)                   # floor already applied by the oracle; re-clamping is a no-op
    mel2 = np.ascontiguousarray(mel_raw.transpose(0, 2, 1))     # [2][frames][64]
    logits, stages = emu.run_wave(mel2, amax - 120.0, nt=6)
    d_tok = np.abs(stages - ref["stages"][:, 0]).max(axis=(0, 2, 3))
    assert d_tok[0] < tol_tok, d_tok
    assert d_tok.max() < 4 * tol_tok, d_tok
    assert np.abs(logits - ref["logits"][:, 0]).max() < tol_logit


def test_emulated_kernel_short_clip_masks_tokens():
    m, sd = _model("uit_xxxs", 11, "f16")
    emu = EncoderEmu(m.packed_image())
    pcm = synth_pcm_i16(2, 8000, seed=8)                # 51 frames -> nt = 3
    o = Oracle(sd)
    ref = o.forward(pcm, want_stages=True)
    mel_raw, amax = o.melspec_db(pcm)
    mel2 = np.ascontiguousarray(mel_raw.transpose(0, 2, 1))
    logits, stages = emu.run_wave(mel2, amax - 120.0, nt=3)
    # oracle tokens are f*3+t; emulator tokens are f*6+t with t<3 valid
    got = stages[:, :, :, :].reshape(2, -1, 4, 6, 128)[:, :, :, :3].reshape(2, -1, 12, 128)
    assert np.abs(got - ref["stages"][:, 0]).max() < 2e-2
    assert np.abs(logits - ref["logits"][:, 0]).max() < 1e-3
