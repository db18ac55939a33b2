"""Counterpart of the reference's inference.py (call site of the hot path: inference.py:55).

    python -m uit_mobile_amd.inference clip.wav [more.wav ...] [-m uit_xs | checkpoint.pt] [-k 3]

Same argument surface (input_wav..., -m/--model, -k/--topk), same output format (label, 4-decimal score,
"Keyword: " prefix for indices > 526, inference.py:57-62).  Differences: the wav is read by uit_mobile_amd.wav
(PCM16) instead of torchaudio, the model runs on the HIP device (`cuda:0`), and public checkpoints are looked up
in the torch hub cache / a local path because the build environment has no network.
"""
import argparse
import csv
from pathlib import Path

import numpy as np
import torch

from . import models
from .models import PRETRAINED_CHECKPOINTS
from .wav import read_wav_pcm16

LABELS = Path(__file__).resolve().parent / "data" / "class_labels.csv"


def load_label_map(path=LABELS):
    with open(path, newline="") as f:
        return {int(r["index"]): r["display_name"] for r in csv.DictReader(f)}


def build_model(model_arg, device="cuda", precision=None):
    """Both checkpoint formats of the reference (inference.py:35-48): a public name -> bare state_dict from the
    hub cache, or a path to {'model': sd, 'config': {model, model_args, num_classes}}."""
    if model_arg in PRETRAINED_CHECKPOINTS:
        params = PRETRAINED_CHECKPOINTS[model_arg]
        dump = torch.hub.load_state_dict_from_url(params["chkpt"], map_location="cpu")
        model = params["model"](precision=precision, **params["model_kwargs"])
        model.load_state_dict(dump, strict=True)
    else:
        dump = torch.load(model_arg, map_location="cpu", weights_only=False)
        if "config" in dump:
            cfg = dump["config"]
            model = getattr(models, cfg["model"])(outputdim=cfg.get("num_classes", 537), precision=precision,
                                                  **cfg["model_args"])
            model.load_state_dict(dump["model"], strict=True)
        else:                                            # bare state_dict: infer the family member from the depth
            depth = 1 + max(int(k.split(".")[1]) for k in dump if k.startswith("blocks."))
            name = {12: "uit_xs", 6: "uit_xxs", 4: "uit_xxxs"}[depth]
            model = getattr(models, name)(outputdim=dump["outputlayer.1.weight"].shape[0], target_length=102,
                                          precision=precision)
            model.load_state_dict(dump, strict=True)
    # prepare(): the set-up a first forward would otherwise hide -- parameter image incl. weight calibration and its budget check (round 5)
    return model.to(device).eval().prepare(device)


def topk_lines(probs, label_maps, k):
    """The printed lines of inference.py:57-62 for one probability row.  A GPU row is reduced on the device
    (postprocess.topk: only k scores cross PCIe); a host row (tests, goldens) uses torch's host topk."""
    lines = []
    probs = torch.as_tensor(probs)
    if probs.is_cuda:
        from .postprocess import topk
        vals, idxs = topk(probs, k)
    else:
        vals, idxs = probs.topk(k)
    for prob, label in zip(vals.tolist(), idxs.tolist()):
        name = label_maps[label]
        if label > 526:
            name = f"Keyword: {name}"
        lines.append(f"{name:<30} {prob:.4f}")
    return lines


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("input_wav", type=Path, nargs="+")
    ap.add_argument("-m", "--model", type=str, nargs="?", default="uit_xs",
                    metavar=f"Public Checkpoint [{','.join(PRETRAINED_CHECKPOINTS)}] or Experiment Path")
    ap.add_argument("-k", "--topk", type=int, default=3, help="Print top-k results")
    ap.add_argument("--precision", default=None, choices=["auto", "f16c", "f16l6", "f16", "f16x2", "f32", "bf16", "fp8", "fp8c", "fp8w"],
                    help="default: auto, the parity mode of every factory (the fp16 kernels on weights with calibrated rounding, held to the "
                         "library's deviation budget, else split weights)")
    args = ap.parse_args(argv)
    label_maps = load_label_map()
    model = build_model(args.model, precision=args.precision)
    for wavpath in args.input_wav:
        pcm, sr = read_wav_pcm16(wavpath)
        assert sr == 16000, "Models are trained on 16khz, please sample your input to 16khz"
        if pcm.ndim > 1:
            pcm = pcm[:, 0]
        with torch.no_grad():
            output = model(torch.from_numpy(np.ascontiguousarray(pcm)).unsqueeze(0).cuda()).squeeze(0)
        print(f"===== {str(wavpath):^20} =====")
        for line in topk_lines(output, label_maps, args.topk):
            print(line)


if __name__ == "__main__":
    main()
