"""Minimal RIFF/WAVE PCM16 reader (stands in for torchaudio.load at inference.py:52; torchaudio is not a
dependency of this package).  Returns (int16 numpy array [T] for mono / [T, C] otherwise, sample_rate)."""
import struct

import numpy as np


def read_wav_pcm16(path):
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack_from("<HHIIHH", body, 0)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, channels, rate, _, _, bits = fmt
    if tag not in (1, 0xFFFE) or bits != 16:
        raise ValueError(f"{path}: only 16-bit PCM is supported (format tag {tag}, {bits} bits)")
    x = np.frombuffer(pcm, dtype="<i2", count=len(pcm) // 2)
    if channels > 1:
        x = x[: len(x) // channels * channels].reshape(-1, channels)
    return x, rate


def write_wav_pcm16(path, pcm, rate=16000):
    pcm = np.asarray(pcm, dtype="<i2")
    ch = 1 if pcm.ndim == 1 else pcm.shape[1]
    body = pcm.tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * ch * 2, ch * 2, 16))
        f.write(b"data" + struct.pack("<I", len(body)) + body)
