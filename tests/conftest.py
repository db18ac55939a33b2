import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
GOLDEN = ROOT / "tests" / "golden"
for p in (str(ROOT), str(GOLDEN)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(GOLDEN / f"{name}.npz"))
    return load


def assert_mel_close(got, want, scale=1.0):
    """dB comparison, graded by how far a bin sits below its clip's peak.

    Bins far below the peak sit on the fp32 FFT round-off floor (e.g. the skirts of a full-scale pure tone):
    any two fp32 FFT algorithms differ there, the reference's pocketfft vs a radix-2 FFT included (measured
    oracle-vs-reference: 8e-5 dB at 50-60 dB down, 4e-3 at 80-90, 0.08 at 110).  Tolerances (dB):
    < 60 dB below peak: 5e-4;  60..90: 1e-2;  >= 90: 0.25   (x `scale`)."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape
    peak = want.reshape(want.shape[0], -1).max(1).reshape((-1,) + (1,) * (want.ndim - 1))
    depth = peak - want
    d = np.abs(got - want)
    for lo, hi, tol in ((-1.0, 60.0, 5e-4), (60.0, 90.0, 1e-2), (90.0, 1e9, 0.25)):
        m = (depth >= lo) & (depth < hi)
        if m.any():
            assert d[m].max() < tol * scale, f"dB diff {d[m].max()} for bins {lo}..{hi} dB below peak"


def kws_loop_restatement(y_pred, y, threshold, num_fillers=527):
    """Loop restatement of the reference's _output_transform_xiaoai (evaluate.py:212-229) on host tensors -- the checker
    for uit_mobile_amd.evaluate.output_transform_kws and the uitk_kws_transform kernel (test infrastructure)."""
    import torch
    y_pred = y_pred.clone()
    mask = torch.ones_like(y_pred)
    mask[:, :num_fillers] = (y_pred[:, :num_fillers] == y_pred[:, :num_fillers].max(dim=1, keepdim=True)[0]).to(torch.int32)
    y_pred = y_pred * mask
    _, yy = y.max(dim=-1)
    for i, scores in enumerate(y_pred):
        idx = scores[0:num_fillers].max(dim=-1)[1]
        if yy[i] < num_fillers:
            yy[i] = idx
        for score in scores[num_fillers:]:
            if score >= threshold:
                y_pred[i][idx] = 0.0
    return y_pred, yy


def centred(tokens):
    """Residual-stream tokens minus their per-token channel mean.  The kernel keeps its residual stream centred (pack.cpp
    removes the channel-mean of everything added to it; every consumer is a LayerNorm, which cannot see a common offset), so
    stage-level comparisons with the reference's stream are made on the centred values."""
    tokens = np.asarray(tokens)
    return tokens - tokens.mean(-1, keepdims=True)
