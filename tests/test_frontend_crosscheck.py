"""CPU: the front-end restatement (oracle/uit_oracle.c, which the golden fixtures' torchaudio stand-in agrees with) against an
INDEPENDENT implementation of the same published algorithm: transformers.audio_utils (numpy, float64).

Why: torchaudio==0.13.0 is an un-vendored dependency of the reference (requirements.txt:17, call site models/uit.py:298-308)
and the reference holds no test for it, so this boundary is "parity unpinned by the reference" (SURVEY.md 8c).  The
cross-check used to live inside tests/golden/make_golden.py on one file; here it runs on every test pass over a pure tone,
noise, a silent + loud batch (the batch-global top_db floor) and odd lengths.
"""
import numpy as np
import pytest

from conftest import assert_mel_close
from oracle.oracle import Oracle
from uit_mobile_amd.synth import make_state_dict, synth_pcm_i16

audio_utils = pytest.importorskip("transformers.audio_utils")


def independent_db(wave, floor=True):
    """[B, T] float -> dB mel [B, 64, frames] by transformers.audio_utils (float64), torchaudio-0.13 parameters:
    periodic Hann 512, hop 160, center + reflect, power 2, 64 HTK mels 0..8 kHz norm=None, 10 log10(max(., 1e-10)).
    The top_db rule is applied here exactly as torchaudio 0.13 applies it to the 3-D tensor the reference passes
    (packed-channel reshape => ONE amax over the whole batch)."""
    fb = audio_utils.mel_filter_bank(257, 64, 0.0, 8000.0, 16000, norm=None, mel_scale="htk")
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(512) / 512)
    out = np.stack([audio_utils.spectrogram(np.asarray(w, np.float64), win, 512, 160, fft_length=512, power=2.0, center=True,
                                            pad_mode="reflect", mel_filters=fb, log_mel="dB", reference=1.0, min_value=1e-10,
                                            db_range=None) for w in wave])
    return np.maximum(out, out.max() - 120.0) if floor else out


@pytest.fixture(scope="module")
def oracle():
    return Oracle(make_state_dict("uit_xxxs", 0))


def test_filterbank_matches_independent_implementation():
    fb = audio_utils.mel_filter_bank(257, 64, 0.0, 8000.0, 16000, norm=None, mel_scale="htk")
    ours = np.asarray(make_state_dict("uit_xxxs", 0)["front_end.0.mel_scale.fb"])
    assert ours.shape == fb.shape == (257, 64)
    assert np.abs(ours - fb).max() < 2e-5


@pytest.mark.parametrize("case", ["noise", "tone", "silent_and_loud", "odd_16001", "odd_4099"])
def test_db_mel_matches_independent_implementation(oracle, case):
    T = {"odd_16001": 16001, "odd_4099": 4099}.get(case, 16000)
    pcm = synth_pcm_i16(4, T, seed=17)                        # clip 1 = loud tone, 2 = silent tail, 3 = very quiet
    if case == "tone":
        t = np.arange(T)
        pcm = np.round(32767 * 0.99 * np.sin(2 * np.pi * 1000.0 * t / 16000.0))[None].astype(np.int16)
    if case == "silent_and_loud":
        pcm[0] = 0                                            # digital silence next to a full-scale square wave
        pcm[1] = np.where((np.arange(T) // 8) % 2 == 0, 32767, -32768).astype(np.int16)
    wave = pcm.astype(np.float64) / 32768.0
    got, amax = oracle.melspec_db(pcm)
    want = independent_db(wave)
    assert got.shape == want.shape == (pcm.shape[0], 64, 1 + T // 160)
    assert amax == pytest.approx(independent_db(wave, floor=False).max(), abs=2e-4)
    assert_mel_close(got, want.astype(np.float32), scale=2.0)
    if case == "silent_and_loud":
        # the silent clip sits ON the floor, and the floor is the LOUD clip's peak - 120 dB (batch-global), not -100 dB
        assert np.all(got[0] == got[0, 0, 0]) and got[0, 0, 0] == pytest.approx(amax - 120.0, abs=1e-4)
        assert amax - 120.0 > -100.0 + 20.0
