"""torch-CPU restatement of the reference hot path, op for op (TEST INFRASTRUCTURE / cpu_baseline only).

Executes the same ATen CPU operators the reference executes (torch.stft, matmul, log10, batch_norm, conv2d,
layer_norm, linear, softmax, relu) in the order of models/uit.py:452-493, without importing the reference (which
cannot travel to the GPU box) or torchaudio (not installed).  It exists so that bench.py's `cpu_baseline` can time
"the reference PyTorch CPU path" on the GPU box's host cores with the kernels the reference would really run; it is
pinned against the reference-generated fixtures in tests/test_oracle_golden.py like the C oracle.
Never imported by the product package.
"""
import torch
import torch.nn.functional as F


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(__import__("numpy").asarray(v))


class TorchPort:
    def __init__(self, state_dict, target_length=102, eval_avg="mean", hop=160, heads=2):
        self.sd = {k: _t(state_dict, k).float() for k in state_dict if k != "init_bn.1.num_batches_tracked"}
        self.depth = 1 + max(int(k.split(".")[1]) for k in self.sd if k.startswith("blocks."))
        self.tl, self.eval_avg, self.hop, self.heads = target_length, eval_avg, hop, heads

    def front_end(self, x):                                   # models/uit.py:298-308 (torchaudio 0.13 semantics)
        sd = self.sd
        win = sd["front_end.0.spectrogram.window"]
        spec = torch.stft(x, n_fft=win.numel(), hop_length=self.hop, win_length=win.numel(), window=win, center=True,
                          pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs().pow(2.0)
        mel = torch.matmul(spec.transpose(-1, -2), sd["front_end.0.mel_scale.fb"]).transpose(-1, -2)
        db = 10.0 * torch.log10(torch.clamp(mel, min=1e-10))
        return torch.max(db, db.amax() - 120.0)                # packed-channel reshape => batch-global amax

    def features_head(self, x):                               # forward_features + forward_head, uit.py:379-404
        sd = self.sd
        x = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=16)
        b, c, f, t = x.shape
        x = x + sd["time_pos_embed"][:, :, :, :t] + sd["freq_pos_embed"]
        x = x.flatten(2).transpose(1, 2)                       # b (f t) c
        H = self.heads
        for i in range(self.depth):
            p = f"blocks.{i}."
            h = F.layer_norm(x, (c,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
            B, N, _ = h.shape
            inner = sd[p + "attn.proj.weight"].shape[1]
            qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, N, 3, H, inner // H).permute(2, 0, 3, 1, 4)
            q, k, v = qkv.unbind(0)
            attn = ((q @ k.transpose(-2, -1)) * (c // H) ** -0.5).softmax(dim=-1)
            h = (attn @ v).transpose(1, 2).reshape(B, N, inner)
            x = x + F.linear(h, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
            h = F.layer_norm(x, (c,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
            h = F.relu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
            x = x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        x = F.layer_norm(x, (c,), sd["norm.weight"], sd["norm.bias"], 1e-6).mean(1)
        x = F.layer_norm(x, (c,), sd["outputlayer.0.weight"], sd["outputlayer.0.bias"], 1e-5)
        return F.linear(x, sd["outputlayer.1.weight"], sd["outputlayer.1.bias"])

    @torch.no_grad()
    def forward(self, wave, return_logits=False):
        sd = self.sd
        x = self.front_end(wave).unsqueeze(1)
        x = F.batch_norm(x.transpose(1, 2), sd["init_bn.1.running_mean"], sd["init_bn.1.running_var"],
                         sd["init_bn.1.weight"], sd["init_bn.1.bias"], False, 0.0, 1e-5).transpose(1, 2)
        if x.shape[-1] > self.tl:
            outs = []
            for f in x.split(self.tl, -1):
                if f.shape[-1] != self.tl:
                    f = x[..., -self.tl:]
                outs.append(self.features_head(f))
            logits = torch.stack(outs, 1)
            probs = logits.sigmoid()
            probs = probs.mean(1) if self.eval_avg == "mean" else probs.max(1)[0]
        else:
            logits = self.features_head(x).unsqueeze(1)
            probs = logits[:, 0].sigmoid()
        return (probs, logits) if return_logits else probs
