"""GPU: the GENERAL engine (csrc/general.hip) -- UITBase geometries outside the published factories', plain fp32 on the device --
against the fixtures the imported reference produced (tests/golden/general_*.npz, make_golden.py --general) and against the C oracle.

Tolerance: the task's 1e-3 on logits / probabilities; measured ~1e-5 (fp32 against fp32: summation order), 4e-4 on the full-scale
pure-tone clip of every batch (the fp32 FFT round-off floor: conftest.assert_mel_close)."""
import warnings

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import assert_mel_close
from general_cases import GENERAL_CASES, state_dict_args
from uit_mobile_amd import models
from uit_mobile_amd.models.uit import UITBase, UitkSetupWarning
from uit_mobile_amd.synth import make_general_state_dict, synth_pcm_i16

pytestmark = pytest.mark.gpu


def build(case, **over):
    kw = dict(case["kwargs"])
    kw["attention_type"] = {"bneck": "BNeckAttention", "full": "Attention"}[case["attention"]]
    kw["act_layer"] = {"relu": nn.ReLU, "gelu": nn.GELU}[case["act"]]
    kw.update(over)
    m = UITBase(**kw)
    sd = make_general_state_dict(**state_dict_args(case))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.cuda().eval(), sd


@pytest.mark.parametrize("tag", list(GENERAL_CASES))
def test_general_engine_against_the_reference(golden, tag):
    case, g = GENERAL_CASES[tag], golden(f"general_{tag}")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UitkSetupWarning)
        m, _ = build(case)
        assert m.general and m.precision == "f32"
        for i, (B, T) in enumerate(case["clips"]):
            pcm = g[f"c{i}_pcm"]
            for x in (torch.from_numpy(pcm).cuda(), torch.from_numpy(pcm.astype(np.float32) / 32768.0).cuda()):
                with torch.no_grad():
                    if case["kwargs"]["pooling"] == "dm":
                        probs, logits = m(x), None
                    else:
                        probs, logits = m(x, return_logits=True)
                    mel = m.melspec_db(x)
                torch.cuda.synchronize()
                if f"c{i}_mel_db" in g:
                    assert_mel_close(mel.cpu().numpy(), g[f"c{i}_mel_db"])
                dp = np.abs(probs.cpu().numpy() - g[f"c{i}_probs"]).reshape(B, -1).max(1)
                tone = np.arange(B) == 1
                assert dp.max() < 1e-3 and dp[~tone].max() < 5e-5, (tag, i, dp)
                if logits is not None:
                    dl = np.abs(logits.cpu().numpy() - g[f"c{i}_logits"]).reshape(B, -1).max(1)
                    assert dl.max() < 1e-3 and dl[~tone].max() < 2e-4, (tag, i, dl)
        assert m.effective_precision() == "f32"


def test_general_engine_against_the_oracle_on_a_larger_batch():
    """70 random clips of 1.3 s through the 128-mel / 1024-point configuration (two crops each) and the oracle; the two-stage API
    (frontend -> amax -> encode: what the data-parallel wrapper uses) gives the same bits as forward()."""
    from test_oracle_golden import general_oracle
    case = GENERAL_CASES["mels128_fft1024"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UitkSetupWarning)
        m, _ = build(case)
        pcm = synth_pcm_i16(70, 20800, seed=77)
        x = torch.from_numpy(pcm).cuda()
        with torch.no_grad():
            probs, logits = m(x, return_logits=True)
            st = m.frontend(x)
            amax = m.amax_tensor(st[2]).clone()
            probs2 = m.encode(st)
        torch.cuda.synchronize()
    ref = general_oracle(case).forward(pcm)
    assert logits.shape == ref["logits"].shape == (70, 2, 50)
    assert torch.equal(probs, probs2)
    assert abs(float(amax.max()) - float(ref["mel_db"].max())) < 1e-3
    tone = np.arange(70) % 8 == 1
    dl = np.abs(logits.cpu().numpy() - ref["logits"]).reshape(70, -1).max(1)
    dp = np.abs(probs.cpu().numpy() - ref["probs"]).reshape(70, -1).max(1)
    print(f"general engine, 70 x 1.3 s clips, 128 mels / n_fft 1024: max|dlogit| {dl.max():.2e} (pure tones) / {dl[~tone].max():.2e} (others), max|dprob| {dp.max():.2e}")
    assert dl.max() < 1e-3 and dp.max() < 1e-3 and dl[~tone].max() < 1e-4


def test_published_geometry_never_takes_the_general_engine_and_the_general_engine_says_what_it_is():
    m = models.uit_xxxs(outputdim=537, target_length=102)
    assert not m.general and m.precision == "auto"
    case = GENERAL_CASES["patch8_token"]
    m, _ = build(case)
    x = torch.from_numpy(synth_pcm_i16(2, 16000, seed=5)).cuda()
    with pytest.warns(UitkSetupWarning, match="GENERAL fp32 engine"):
        with torch.no_grad():
            m(x)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                     # said once per model
        with torch.no_grad():
            m(x)
    assert m.calibration_report() is None
    for call in (lambda: m.set_option("enc_waves", 4), lambda: m.calibrate(), lambda: m.self_check(), lambda: m.debug_tokens(x), lambda: m.packed_image()):
        with pytest.raises(NotImplementedError, match="general"):
            call()
    with pytest.raises(ValueError, match="shorter than one patch|longer than n_fft"):
        with torch.no_grad():
            m(x[:, :100])


def test_general_engine_is_stream_ordered_after_prepare():
    """prepare() reserves the workspace; later forwards of that shape are enqueue-only: a HIP graph captures and replays one."""
    case = GENERAL_CASES["tl256_h4"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UitkSetupWarning)
        m, _ = build(case)
        m.prepare("cuda", (3, 30000))
        x = torch.from_numpy(synth_pcm_i16(3, 30000, seed=9).astype(np.float32) / 32768.0).cuda()
        with torch.no_grad():
            want = m(x).clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s), torch.no_grad():
                out = m(x)
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)


def test_fuzz_over_general_geometries_against_the_oracle():
    """tools/fuzz_general.py, time-boxed: random embed_dim / heads / mlp_ratio / depth / attention / activation / pooling / patch size / n_mels /
    n_fft (powers of two and not) / win / hop / center / target_length / eval_avg / f_min / f_max, batch and clip length (short, full,
    several crops) -- every case inside 1e-3 of the C oracle (profiles/round6/fuzz_general.log: 120 cases, worst 4.7e-4 on a pure tone)."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
    import fuzz_general
    rows, bad = fuzz_general.run(cases=40, seed=7, budget_s=45, verbose=False)
    assert len(rows) >= 15 and not bad, "\n".join(bad)


def test_general_engine_from_plain_c(tmp_path):
    """The general engine is C ABI too (include/uitk.h: uitk_general_*): a plain-C99 consumer (tests/native/cabi_general_consumer.c: gcc, no
    C++, no Python, no torch) builds the 8 x 8-patch / 96-wide / 3-head / cls-token model from raw float arrays -- after seeing uitk_create
    refuse that geometry by name -- and writes, bit for bit, what the Python wrapper returns for the same checkpoint and clips."""
    import shutil
    import subprocess
    from pathlib import Path
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    ROOT = Path(__file__).resolve().parents[1]
    case = GENERAL_CASES["patch8_token"]
    k = case["kwargs"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UitkSetupWarning)
        m, sd = build(case)
        pcm = synth_pcm_i16(3, 17000, seed=12)
        with torch.no_grad():
            x = torch.from_numpy(pcm).cuda()
            want_p, want_mel = m(x).cpu().numpy(), m.melspec_db(x).cpu().numpy()
    order = ["front_end.0.spectrogram.window", "front_end.0.mel_scale.fb", "init_bn.1.weight", "init_bn.1.bias", "init_bn.1.running_mean",
             "init_bn.1.running_var", "patch_embed.proj.weight", "patch_embed.proj.bias", "time_pos_embed", "freq_pos_embed", "norm.weight",
             "norm.bias", "outputlayer.0.weight", "outputlayer.0.bias", "outputlayer.1.weight", "outputlayer.1.bias", "cls_token", "token_pos_embed"]
    for b in range(k["depth"]):
        order += [f"blocks.{b}.{n}" for n in ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
                                              "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")]
    hdr = [k["depth"], k["embed_dim"], k["num_heads"], int(k["embed_dim"] * k["mlp_ratio"]), k["outputdim"], k["target_length"], 64, 512, 160, 512,
           k["patch_size"], 1, 1, 1, 2, pcm.shape[0], pcm.shape[1]]
    blob = tmp_path / "general.bin"
    with open(blob, "wb") as f:
        f.write(np.array(hdr, np.int32).tobytes())
        for name in order:
            f.write(np.ascontiguousarray(np.asarray(sd[name]), dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(pcm).tobytes())
    exe, lib = tmp_path / "cabi_general_consumer", ROOT / "uit_mobile_amd"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{ROOT / 'include'}",
                    str(ROOT / "tests" / "native" / "cabi_general_consumer.c"), "-o", str(exe), f"-L{lib}", "-luitk", "-L/opt/rocm/lib", "-lamdhip64",
                    f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe), str(blob), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    print(r.stdout.strip(), r.stderr.strip())
    assert r.returncode == 0 and "patch_size must be 16" in r.stdout or "embed_dim must be 128" in r.stdout, r.stdout + r.stderr
    out = np.fromfile(tmp_path / "out.bin", np.float32)
    n = want_p.size
    assert np.array_equal(out[:n].reshape(want_p.shape), want_p) and np.array_equal(out[n:].reshape(want_mel.shape), want_mel)


def test_general_engine_and_the_f32_rung_agree_on_the_published_geometry():
    """Two independent fp32 implementations of the same model on the device: the `f32` rung of the MFMA library (the product front-end
    + the folded network of calibrate.hip) and the general engine (its own front-end, the reference's order of operations, nothing
    folded) -- made to take the PUBLISHED geometry here, which the wrapper never does by itself -- and the C oracle: all three within
    2e-4 on logits on noise clips (the full-scale tone: the FFT floor)."""
    from oracle.oracle import Oracle
    from uit_mobile_amd.synth import make_trained_like_state_dict
    sd = make_trained_like_state_dict("uit_xxs", 204)
    pcm = synth_pcm_i16(24, 16000, seed=31)
    outs = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for engine in ("f32", "general"):
            m = models.uit_xxs(outputdim=537, target_length=102, precision="f32")
            m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
            m.general = engine == "general"
            m = m.cuda().eval()
            with torch.no_grad():
                outs[engine] = m(torch.from_numpy(pcm).cuda(), return_logits=True)[1].cpu().numpy()
            assert m.effective_precision() == "f32" and (m._handle_obj(torch.device("cuda", 0)).general == (engine == "general"))
    ref = Oracle(sd).forward(pcm)["logits"]
    tone = np.arange(24) % 8 == 1
    d = {k: np.abs(a - b).reshape(24, -1).max(1) for k, (a, b) in {"general-oracle": (outs["general"], ref), "f32-oracle": (outs["f32"], ref),
                                                                    "general-f32": (outs["general"], outs["f32"])}.items()}
    print("published geometry, uit_xxs trained-like 204, 24 clips: " + ", ".join(f"{k} {v[~tone].max():.1e} (tones {v[tone].max():.1e})" for k, v in d.items()))
    for k, v in d.items():
        assert v[~tone].max() < 2e-4 and v.max() < 1e-3, (k, v)


def test_general_engine_refuses_to_grow_its_workspace_inside_a_capture():
    case = GENERAL_CASES["patch8_token"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UitkSetupWarning)
        m, _ = build(case)
        x = torch.from_numpy(synth_pcm_i16(2, 16000, seed=3)).cuda()
        m._handle_obj(x.device)                                # the handle exists, the workspace does not
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            gr = torch.cuda.CUDAGraph()
            with pytest.raises(RuntimeError, match="workspace has to grow"):
                with torch.cuda.graph(gr, stream=s), torch.no_grad():
                    m(x)
        torch.cuda.synchronize()
        with torch.no_grad():
            assert torch.isfinite(m(x)).all()                  # outside a capture the same call allocates and runs


def _general_ranks_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    from uit_mobile_amd.sharded import ShardedUiT
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m, _ = build(GENERAL_CASES["patch8_token"])
            pcm = _two_rank_pcm()
            rows = pcm.shape[0] // world
            outs = []
            for gather in ("rccl", "peer"):
                sh = ShardedUiT(m, gather=gather)
                out = sh(torch.from_numpy(pcm[rank * rows:(rank + 1) * rows]).cuda())
                torch.cuda.synchronize()
                outs.append(out.cpu().numpy())
        q.put((rank, outs))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _two_rank_pcm():
    pcm = synth_pcm_i16(8, 16000, seed=21)
    pcm[[1, 5]] = pcm[[5, 1]]              # the loud tone (clip 1) into rank 1's shard, the clip with a silent tail (2) stays in rank 0's:
    return pcm                             # rank 0's dB floor is decided on rank 1


def test_general_engine_under_the_data_parallel_wrapper():
    """SURVEY 8e for a general geometry: two ranks sharing the GPU (gloo; the collectives staged through the host), batch shards, the MAX
    all-reduce of the general engine's amax slots between its two stages, the gather ('rccl' path and the copy-engine PeerGather) --
    every rank holds the whole batch's output, bit for bit the single-rank result, although rank 0's top_db floor is set by a clip
    that only rank 1 sees."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + os.getpid() % 400
    procs = [ctx.Process(target=_general_ranks_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m, _ = build(GENERAL_CASES["patch8_token"])
        pcm = _two_rank_pcm()
        with torch.no_grad():
            single = m(torch.from_numpy(pcm).cuda()).cpu().numpy()
            alone = m(torch.from_numpy(pcm[:4]).cuda()).cpu().numpy()      # rank 0's shard WITHOUT the other shard's amax
    assert np.abs(alone - single[:4]).max() > 1e-4                          # (the floor does matter in this batch)
    for r in range(2):
        for out in res[r]:
            assert out.shape == single.shape and np.array_equal(out, single), r


def test_the_wide_layer_tile_gives_the_same_bits():
    """Wide models (K >= 512, O >= 256) run their dense layers on a 128 x 128 tile once a call has 8192 tokens (calibrate.hip); every
    output is the same sum of the same products in the same order as on the 64 x 64 tile: the constructor's default UITBase on one 10 s
    clip (249 tokens: the small tile) and on the same clip 34 times over (8466 tokens: the large one) -- bit for bit, and the fixture of
    the imported reference still holds."""
    case = GENERAL_CASES["default_uitbase"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UitkSetupWarning)
        m, _ = build(case)
        pcm = synth_pcm_i16(1, 160000, seed=19)
        one = torch.from_numpy(pcm).cuda()
        many = torch.from_numpy(np.repeat(pcm, 34, 0)).cuda()
        with torch.no_grad():
            a = m(one, return_logits=True)[1]
            b = m(many, return_logits=True)[1]
            # ... and 400 times over: 8 MB of work buffers per clip-crop of this model, so the engine walks the batch in TWO chunks (384 + 16
            # rows of its 3 GiB workspace) -- the chunk loop's row offsets (crops, logits, probabilities)
            c_probs, c = m(torch.from_numpy(np.repeat(pcm, 400, 0)).cuda(), return_logits=True)
            a_probs = m(one)
    assert torch.equal(a.expand_as(b), b)
    assert torch.equal(a.expand_as(c), c) and torch.equal(a_probs.expand_as(c_probs), c_probs)
