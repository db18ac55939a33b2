"""Device-side post-processing of the probability rows (the step after the hot path).

`topk`            = `output.topk(k)` of inference.py:57 / evaluate.py:263
`kws_transform`   = `_output_transform_xiaoai` of evaluate.py:212-229 (a per-sample Python loop on the host there)

Both run as HIP kernels of libuitk.so on the tensor's device and stream; only [B, k] (or class indices) need to cross
PCIe afterwards.  CUDA tensors only -- there is no CPU implementation behind these calls.
"""
import ctypes

import torch

from . import _lib


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _check_rows(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor: uit_mobile_amd.postprocess has no CPU path")
    if t.dim() != 2:
        raise ValueError(f"{name} must be [B, C], got {tuple(t.shape)}")


def topk(probs, k):
    """values [B,k] float32, indices [B,k] int64 = probs.topk(k) (descending; equal scores: lower class index first)."""
    squeeze = isinstance(probs, torch.Tensor) and probs.dim() == 1
    if squeeze:
        probs = probs.unsqueeze(0)
    _check_rows(probs, "probs")
    p = probs.detach().to(torch.float32).contiguous()
    B, C = p.shape
    vals = torch.empty((B, k), dtype=torch.float32, device=p.device)
    idx = torch.empty((B, k), dtype=torch.int32, device=p.device)
    with torch.cuda.device(p.device):
        _lib.check(_lib.load().uitk_topk(ctypes.c_void_p(p.data_ptr()), B, C, int(k), ctypes.c_void_p(vals.data_ptr()),
                                         ctypes.c_void_p(idx.data_ptr()), _stream(p)), "uitk_topk")
    idx = idx.to(torch.int64)
    return (vals[0], idx[0]) if squeeze else (vals, idx)


def kws_transform(y_pred, y=None, threshold=0.2, num_fillers=527):
    """(y_pred', y_index) as the reference's Accuracy@threshold output transform returns them; y_pred is not modified."""
    _check_rows(y_pred, "y_pred")
    out = y_pred.detach().to(torch.float32).clone(memory_format=torch.contiguous_format)
    B, C = out.shape
    yt, yi = None, None
    if y is not None:
        _check_rows(y, "y")
        if y.shape != out.shape or y.device != out.device:
            raise ValueError("y must have y_pred's shape and device")
        yt = y.detach().to(torch.float32).contiguous()
        yi = torch.empty((B,), dtype=torch.int64, device=out.device)
    with torch.cuda.device(out.device):
        _lib.check(_lib.load().uitk_kws_transform(
            ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(yt.data_ptr()) if yt is not None else None, B, C,
            int(num_fillers), float(threshold), ctypes.c_void_p(yi.data_ptr()) if yi is not None else None, _stream(out)),
            "uitk_kws_transform")
    return out, yi
