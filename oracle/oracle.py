"""ctypes binding of oracle/liboracle.so (the C restatement in uit_oracle.c).  TEST INFRASTRUCTURE ONLY.

`Oracle(state_dict, depth=...)` takes a reference-layout state_dict (numpy arrays or torch tensors, keys as in
SURVEY.md 8a row 13) and exposes `forward`, `melspec_db`.
"""
import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None


class OracleCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("depth", "dim", "heads", "hidden", "outputdim", "n_mels", "n_fft", "hop", "win",
                 "target_length", "patch", "eval_avg")] + [("top_db", ctypes.c_float), ("attn_full", ctypes.c_int),
                                                           ("act_gelu", ctypes.c_int), ("pool_dm", ctypes.c_int)]


def build(force=False):
    so = _HERE / "liboracle.so"
    src = _HERE / "uit_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = ctypes.CDLL(str(so))
        pp = ctypes.POINTER(ctypes.c_void_p)
        L.oracle_forward.argtypes = [ctypes.POINTER(OracleCfg), pp, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_forward.restype = ctypes.c_int
        L.oracle_melspec_db.argtypes = [ctypes.POINTER(OracleCfg), pp, ctypes.c_void_p, ctypes.c_int,
                                        ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_melspec_db.restype = ctypes.c_int
        L.oracle_num_crops.argtypes = [ctypes.POINTER(OracleCfg), ctypes.c_int64]
        L.oracle_num_crops.restype = ctypes.c_int
        L.oracle_set_threads.argtypes = [ctypes.c_int]
        L.oracle_set_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


_GLOBAL_KEYS = [
    "front_end.0.spectrogram.window", "front_end.0.mel_scale.fb",
    "init_bn.1.weight", "init_bn.1.bias", "init_bn.1.running_mean", "init_bn.1.running_var",
    "patch_embed.proj.weight", "patch_embed.proj.bias", "time_pos_embed", "freq_pos_embed",
    "norm.weight", "norm.bias", "outputlayer.0.weight", "outputlayer.0.bias",
    "outputlayer.1.weight", "outputlayer.1.bias", "cls_token", "token_pos_embed",
]
_BLOCK_KEYS = ["norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
               "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
               "mlp.fc2.weight", "mlp.fc2.bias"]


def _np(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=np.float32)


class Oracle:
    def __init__(self, state_dict, target_length=102, eval_avg="mean", hop=160, heads=2, act="relu", pooling="mean"):
        sd = {k: _np(v) for k, v in state_dict.items() if k != "init_bn.1.num_batches_tracked"}
        depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
        dim = sd["norm.weight"].shape[0]
        fb = sd["front_end.0.mel_scale.fb"]
        self.cfg = OracleCfg(depth=depth, dim=dim, heads=heads, hidden=sd["blocks.0.mlp.fc1.weight"].shape[0],
                             outputdim=sd["outputlayer.1.weight"].shape[0], n_mels=fb.shape[1],
                             n_fft=2 * (fb.shape[0] - 1), hop=hop,
                             win=sd["front_end.0.spectrogram.window"].shape[0],
                             target_length=target_length, patch=sd["patch_embed.proj.weight"].shape[-1],
                             eval_avg={"mean": 0, "max": 1}[eval_avg], top_db=120.0,
                             # BNeckAttention: qkv rows 3*dim/4; Attention: 3*dim  (models/uit.py:103 / :139)
                             attn_full=int(sd["blocks.0.attn.qkv.weight"].shape[0] == 3 * dim),
                             act_gelu={"relu": 0, "gelu": 1}[act], pool_dm={"mean": 0, "dm": 1, "token": 2}[pooling])
        self._keep = [sd[k] for k in _GLOBAL_KEYS]
        for b in range(depth):
            self._keep += [sd[f"blocks.{b}.{k}"] for k in _BLOCK_KEYS]
        self._ptrs = (ctypes.c_void_p * len(self._keep))(*[a.ctypes.data for a in self._keep])

    # ------------------------------------------------------------------
    def n_frames(self, T):
        return 1 + T // self.cfg.hop

    def num_crops(self, T):
        return lib().oracle_num_crops(ctypes.byref(self.cfg), T)

    def _pcm(self, pcm):
        if hasattr(pcm, "detach"):
            pcm = pcm.detach().cpu().numpy()
        if pcm.dtype == np.int16:
            return np.ascontiguousarray(pcm), 1
        return np.ascontiguousarray(pcm, dtype=np.float32), 0

    def melspec_db(self, pcm):
        """-> (mel_db [B, n_mels, nF] after the batch-global top_db floor, amax)"""
        pcm, i16 = self._pcm(pcm)
        B, T = pcm.shape
        out = np.empty((B, self.cfg.n_mels, self.n_frames(T)), np.float32)
        amax = ctypes.c_float()
        rc = lib().oracle_melspec_db(ctypes.byref(self.cfg), self._ptrs, pcm.ctypes.data, i16, B, T,
                                     out.ctypes.data, ctypes.byref(amax))
        if rc:
            raise RuntimeError(f"oracle_melspec_db rc={rc}")
        return out, amax.value

    def forward(self, pcm, want_stages=False):
        """-> dict(probs [B,C], logits [B,ncrops,C], mel_db [B,n_mels,nF], stages [B,ncrops,depth+1,N,dim])"""
        pcm, i16 = self._pcm(pcm)
        B, T = pcm.shape
        c = self.cfg
        nF, nc = self.n_frames(T), self.num_crops(T)
        N = (c.n_mels // c.patch) * (min(nF, c.target_length) // c.patch) + (1 if c.pool_dm == 2 else 0)
        probs = np.empty((B, c.outputdim), np.float32)
        logits = np.empty((B, nc, c.outputdim), np.float32)
        mel = np.empty((B, c.n_mels, nF), np.float32)
        stages = np.empty((B, nc, c.depth + 1, N, c.dim), np.float32) if want_stages else None
        rc = lib().oracle_forward(ctypes.byref(c), self._ptrs, pcm.ctypes.data, i16, B, T, probs.ctypes.data,
                                  logits.ctypes.data, mel.ctypes.data,
                                  stages.ctypes.data if want_stages else None)
        if rc:
            raise RuntimeError(f"oracle_forward rc={rc}")
        return dict(probs=probs, logits=logits, mel_db=mel, stages=stages)


def set_threads(n):
    return lib().oracle_set_threads(int(n))
