"""Counterpart of the hot-path callers in the reference's evaluate.py: the batch step `Evaluator._inference`
(evaluate.py:53-66) with its pad rule, `test_sample` (:247-263) and a plain batch loop standing in for the ignite
engine (:153).  Metric post-processing (sklearn / ignite, evaluate.py:94-101, 212-229) is CPU-side scoring and out
of scope; `run` returns the stacked model outputs so any scorer can consume them.
"""
import numpy as np
import torch

from .inference import build_model
from .wav import read_wav_pcm16

DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")      # run.py:27


def transfer_to_device(batch, device=DEVICE):                               # run.py:30-32
    return (x.to(device, non_blocking=True) if isinstance(x, torch.Tensor) else x for x in batch)


def pad_to_target(data, model):
    """evaluate.py:56-64: right-zero-pad so that the clip has target_length-1 hops."""
    if hasattr(model, "target_length"):
        t_len = model.target_length - 1
        input_nframes = data.shape[-1] / model.hop_size
        if input_nframes < t_len:
            diff = int((t_len - input_nframes) * model.hop_size)
            data = torch.nn.functional.pad(data, (0, diff), mode="constant")
    return data


def sequential_pad(waves, padding_value=0.0):
    """dataset.py:183-204 collate: zero right-pad variable-length clips to the batch maximum; returns (batch, lengths).
    int16 clips stay int16 (the x(1/32768) of dataset.py:44-46 is fused into the kernel's load)."""
    lengths = [int(w.shape[-1]) for w in waves]
    dtype = torch.int16 if all(torch.as_tensor(w).dtype == torch.int16 for w in waves) else torch.float32
    out = torch.full((len(waves), max(lengths)), padding_value, dtype=dtype)
    for i, w in enumerate(waves):
        w = torch.as_tensor(w)
        out[i, : lengths[i]] = w if dtype == torch.int16 else w.to(torch.float32) / (32768.0 if w.dtype == torch.int16 else 1.0)
    return out, torch.as_tensor(lengths)


def load_pretrained(model, trained_model):
    """Counterpart of utils.load_pretrained (utils.py:184-205, called at evaluate.py:37): keep only the checkpoint
    entries whose key AND shape match the model, always take the positional embeddings (the model's load_state_dict
    resizes them, uit.py:416-450), then load strictly."""
    if 'model' in trained_model:
        trained_model = trained_model['model']
    model_dict = model.state_dict()
    pretrained = {k: v for k, v in trained_model.items() if k in model_dict and model_dict[k].shape == v.shape}
    assert len(pretrained) > 0, "Couldnt load pretrained model"
    if 'time_pos_embed' in trained_model:
        pretrained['time_pos_embed'] = trained_model['time_pos_embed']
        pretrained['freq_pos_embed'] = trained_model['freq_pos_embed']
    model_dict.update(pretrained)
    model.load_state_dict(model_dict, strict=True)
    return model


def output_transform_kws(y_pred, y, threshold=0.2, num_fillers=527):
    """`_output_transform_xiaoai` (evaluate.py:212-229) without its per-sample Python loop: keep only the maximal
    AudioSet ("filler") score, map filler targets to it, and zero it whenever any keyword score reaches the threshold
    (so Accuracy@threshold picks the keyword).  y_pred [B, C] probabilities, y [B, C] one/multi-hot targets ->
    (y_pred', y_index).  Model outputs live on the GPU: CUDA tensors go through the HIP kernel
    (`postprocess.kws_transform`); host tensors (the reference's own setting for this host-side metric hook) take the
    vectorised torch form below."""
    if y_pred.is_cuda:
        from .postprocess import kws_transform
        return kws_transform(y_pred, y, threshold, num_fillers)
    fill = y_pred[:, :num_fillers]
    fmax, fidx = fill.max(dim=1, keepdim=True)
    out = y_pred.clone()
    out[:, :num_fillers] = fill * (fill == fmax).to(fill.dtype)
    tgt = y.max(dim=-1)[1]
    tgt = torch.where(tgt < num_fillers, fidx.squeeze(1), tgt)
    fire = (y_pred[:, num_fillers:] >= threshold).any(dim=1)
    rows = torch.nonzero(fire, as_tuple=True)[0]
    out[rows, fidx.squeeze(1)[rows]] = 0.0
    return out, tgt


class Evaluator:
    def __init__(self, experiment_path, device=DEVICE, precision=None):
        self.model = build_model(experiment_path, device=device, precision=precision)
        self.device = device

    def _inference(self, batch, pad=False):
        """batch = (data, targets, lengths, filenames) as produced by the reference's collate."""
        with torch.no_grad():
            data, targets, lengths, filenames = transfer_to_device(batch, self.device)
            if pad:
                data = pad_to_target(data, self.model)
            clip_out = self.model(data)
            return clip_out, targets

    def run(self, dataloader, pad=False, num_labels=None):
        """Plain batch loop (the ignite engine of evaluate.py:153 without the metrics)."""
        outs, tgts = [], []
        for batch in dataloader:
            y, t = self._inference(batch, pad=pad)
            outs.append(y[..., :num_labels] if num_labels else y)          # evaluate.py:92 slices to 527
            tgts.append(t)
        return torch.cat(outs), tgts

    def test_sample(self, sample):
        pcm, sr = read_wav_pcm16(sample)
        wav = torch.from_numpy(np.ascontiguousarray(pcm)).unsqueeze(0)
        wav = pad_to_target(wav, self.model)
        pred = self.model(wav.to(self.device))[0]
        from .postprocess import topk
        vals, idxs = topk(pred, 5)                                   # evaluate.py:263, reduced on the device
        lines = []
        for val, idx in zip(vals.tolist(), idxs.tolist()):
            lines.append(f"[{int(idx):=3}] : {float(val) * 100:.2f}")
        print("\n".join(lines))
        return pred.cpu()
