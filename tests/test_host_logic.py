"""CPU: host-side mirror of the reference interface -- state_dict keys/shapes, checkpoint formats, pos-embed resize,
crop planning, pad rule, WAV reader, label lookup, C-ABI exports, loud failure without a device."""
import ctypes
from pathlib import Path

import numpy as np
import pytest
import torch

from uit_mobile_amd.synth import make_state_dict
from uit_mobile_amd import _lib, models
from uit_mobile_amd.evaluate import pad_to_target, sequential_pad
from uit_mobile_amd.inference import load_label_map, topk_lines
from uit_mobile_amd.wav import read_wav_pcm16, write_wav_pcm16

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    header = (ROOT / "include" / "uitk.h").read_text()
    import re
    declared = set(re.findall(r"\b(uitk_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None
    assert L.uitk_version().startswith(b"uitk")


def test_create_fails_loudly_without_device():
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    m = models.uit_xxxs(outputdim=537, target_length=102).eval()
    cfg = m._config()
    w, keep = m._host_weights()
    out = ctypes.c_void_p()
    rc = _lib.load().uitk_create(ctypes.byref(cfg), ctypes.byref(w), ctypes.byref(out))
    assert rc == _lib.ERR_NO_DEVICE and "no CPU path" in _lib.last_error()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 16000))
    with pytest.raises(RuntimeError):                  # the two-stage API and reserve have no CPU path either
        m.frontend(torch.zeros(1, 16000))
    with pytest.raises(RuntimeError):
        m.reserve(4, 16000, "cpu")
    from uit_mobile_amd.postprocess import kws_transform, topk
    with pytest.raises(RuntimeError):
        topk(torch.rand(2, 537), 3)
    with pytest.raises(RuntimeError):
        kws_transform(torch.rand(2, 537))


@pytest.mark.parametrize("name,depth", [("uit_xs", 12), ("uit_xxs", 6), ("uit_xxxs", 4)])
def test_state_dict_matches_reference_layout(name, depth):
    m = getattr(models, name)(outputdim=537, target_length=102)
    sd = make_state_dict(name, 0)                  # loaded with strict=True into the REAL reference in make_golden.py
    mine = m.state_dict()
    assert set(mine) == set(sd)
    for k in sd:
        assert tuple(mine[k].shape) == tuple(np.asarray(sd[k]).shape), k
    assert m.depth == depth and m.target_length == 102 and m.hop_size == 160 and m.n_mels == 64
    assert m.patch_embed.grid_size == (4, 6) and m.outputdim == 537
    assert set(models.PRETRAINED_CHECKPOINTS) == {"uit_xs", "uit_xxs", "uit_xxxs"}
    assert models.PRETRAINED_CHECKPOINTS[name]["model_kwargs"] == dict(outputdim=537, target_length=102)


@pytest.mark.parametrize("name,depth,gelu", [("audio_transformer_h128_d4_m3", 4, True), ("audio_transformer_h128_d4_m3_relu", 4, False),
                                             ("audio_transformer_h128_d6_m3", 6, True), ("audio_transformer_h128_d6_m3_relu", 6, False)])
def test_other_factories_match_reference_layout(name, depth, gelu):
    """the non-bottleneck factories (models/uit.py:514-579): full-width Attention (qkv 128->384, proj 128->128),
    GELU unless `_relu`, mean pooling.  The same synthetic checkpoints are loaded strict=True into the REAL reference by
    make_golden.py --variants, which pins the key set and shapes."""
    m = getattr(models, name)(outputdim=537, target_length=102)
    sd = make_state_dict(name, 0, attention="full")
    mine = m.state_dict()
    assert set(mine) == set(sd)
    for k in sd:
        assert tuple(mine[k].shape) == tuple(np.asarray(sd[k]).shape), k
    assert tuple(mine["blocks.0.attn.qkv.weight"].shape) == (384, 128) and tuple(mine["blocks.0.attn.proj.weight"].shape) == (128, 128)
    assert m.depth == depth and m.attention_full and m.act_gelu == gelu and m.pooling == "mean"
    assert m.precision == "auto"                    # the default of every factory (round 5): 'f16c' held to the library's budget, else split weights
    m16 = getattr(models, name)(outputdim=537, target_length=102, precision="f16").eval()
    assert (8 + 40 * depth) * 16384 > len(m16.packed_image()) > (8 + 20 * depth) * 16384       # the 20-stage single-weight layout
    assert len(m.eval().packed_image()) == len(m16.packed_image())                             # auto / f16c: the same layout, other roundings
    img = getattr(models, name)(outputdim=537, target_length=102, precision="f16x2").eval().packed_image()      # split weights (round 3)
    assert len(img) > (8 + 40 * depth) * 16384      # host-only packing works without a GPU
    with pytest.raises(NotImplementedError):        # built for the uit_* family only
        getattr(models, name)(outputdim=537, target_length=102, precision="fp8")
    assert getattr(models, name)(outputdim=537, target_length=102, pooling="token").precision == "auto"       # any pooling
    assert len(getattr(models, name)(outputdim=537, target_length=102, pooling="dm", precision="f16x2").eval().packed_image()) == len(img)


def test_precision_kwarg_surface():
    """precision is this package's own kwarg: 'f16' (default, parity), 'f16x2' (split weights), 'bf16' / 'fp8' (measured modes);
    anything else is a ValueError like the reference's unknown-eval_avg; non-mean pooling is f16 only."""
    for prec, stages in (("f16", 14), ("bf16", 14), ("fp8", 8), ("f16x2", 28)):
        m = models.uit_xxxs(outputdim=537, target_length=102, precision=prec).eval()
        assert m.precision == prec
        assert len(m.packed_image()) > (8 + stages * 4 + 4) * 16384
    with pytest.raises(ValueError):
        models.uit_xxxs(outputdim=537, target_length=102, precision="int8")
    assert models.uit_xxxs(outputdim=537, target_length=102, precision="f16x2", pooling="token").eval().packed_image()      # (round 3: built)
    with pytest.raises(NotImplementedError):
        models.uit_xxxs(outputdim=537, target_length=102, precision="fp8", pooling="token")


def test_strict_load_and_pos_embed_resize():
    m = models.uit_xxxs(outputdim=537, target_length=102)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xxxs", 1).items()}
    bad = dict(sd); bad.pop("norm.weight")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad, strict=True)
    long = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xxxs", 1, target_length=160).items()}
    assert long["time_pos_embed"].shape[-1] == 10
    m.load_state_dict(long, strict=True)           # truncated to 6 (models/uit.py:432-434)
    assert torch.equal(m.time_pos_embed, long["time_pos_embed"][..., :6])
    short = dict(sd); short["time_pos_embed"] = sd["time_pos_embed"][..., :3]
    m.load_state_dict(short, strict=True)          # bilinear interpolation 3 -> 6 (models/uit.py:436-440)
    ref = torch.nn.functional.interpolate(short["time_pos_embed"], size=(1, 6), align_corners=False, mode="bilinear")
    assert torch.allclose(m.time_pos_embed, ref)


def test_unsupported_variants_raise():
    tl = dict(target_length=102)                                # (the published geometry: the MFMA kernels; UITBase's own default, 1012 frames, is a general one)
    with pytest.raises(NotImplementedError):
        models.uit_xs(patch_size=16, patch_stride=8, **tl)      # overlapping patches: no engine
    assert models.uit_xs(patch_size=8, patch_stride=8, **tl).general        # (round 6: other geometries run on the general fp32 engine)
    assert models.uit_xs(pooling="dm", **tl).pooling == "dm"    # POOL = 1 instantiation
    assert models.UITBase(embed_dim=128, depth=4, num_heads=2, mlp_ratio=3.0, **tl).pooling == "token"   # the reference's defaults: POOL = 2
    with pytest.raises(NotImplementedError):
        models.uit_xs(pooling="token", precision="bf16", **tl)
    with pytest.raises(NotImplementedError):
        models.uit_xs(pooling="dm", precision="bf16", **tl)
    with pytest.raises(NotImplementedError):
        models.uit_xs(act_layer=torch.nn.SiLU, **tl)
    for pooling in ("mean", "dm", "token"):                     # the fp32 rung: any family, any pooling; its image is the f16 one (front-end tables)
        mf = models.uit_xxxs(outputdim=537, target_length=102, pooling=pooling, precision="f32")
        assert len(mf.packed_image()) == len(models.uit_xxxs(outputdim=537, target_length=102, pooling=pooling, precision="f16").packed_image())
    for prec in ("f16a", "f16ax"):                              # the split-activation precisions are pooling 'token''s ladder only
        with pytest.raises(NotImplementedError):
            models.uit_xs(precision=prec, **tl)
        assert models.uit_xs(pooling="token", precision=prec, **tl).precision == prec
    m = models.uit_xs(embed_dim=256, **tl)          # geometry outside the MFMA kernels' family: the general engine, which has no packed image ...
    assert m.general
    with pytest.raises(NotImplementedError):
        m.packed_image()
    cfg = m._config()                               # ... and the MFMA side of the C ABI still rejects it by name
    w, keep = m._host_weights()
    need = ctypes.c_size_t()
    cfg.precision = _lib.PREC_F16
    assert _lib.load().uitk_pack_weights(ctypes.byref(cfg), ctypes.byref(w), None, 0, ctypes.byref(need)) == _lib.ERR_UNSUPPORTED
    assert "embed_dim must be 128" in _lib.last_error()


def test_causal_kwarg_is_swallowed_as_in_the_reference():
    """`Attention(causal=True)` exists in the reference (models/uit.py:131, 164-169) but nothing can set it: Block builds its attention
    without it (:224-228) and UITBase drops unknown kwargs (:276).  A factory called with causal=True is therefore the ordinary model there,
    and here ("what's missing" 6)."""
    a, b = models.audio_transformer_h128_d4_m3(causal=True), models.audio_transformer_h128_d4_m3()
    assert a.attention_full and sorted(a.state_dict()) == sorted(b.state_dict())
    ca, cb = a._config(), b._config()
    assert all(getattr(ca, f) == getattr(cb, f) for f, _ in ca._fields_)


def test_crop_planning_matches_reference_rule():
    m = models.uit_xs(outputdim=537, target_length=102)
    assert [m.num_frames(t) for t in (16000, 16160, 16384, 160000)] == [101, 102, 103, 1001]
    assert [m.num_crops(t) for t in (16000, 16160, 16384, 50000, 160000)] == [1, 1, 2, 4, 10]


def test_pad_rule_and_collate():
    m = models.uit_xs(outputdim=537, target_length=102)
    x = torch.zeros(2, 16000)
    assert pad_to_target(x, m).shape[-1] == 16160          # evaluate.py:56-64
    assert pad_to_target(torch.zeros(2, 16160), m).shape[-1] == 16160
    b, lengths = sequential_pad([torch.ones(5, dtype=torch.int16), torch.ones(3, dtype=torch.int16)])
    assert b.dtype == torch.int16 and b.tolist() == [[1, 1, 1, 1, 1], [1, 1, 1, 0, 0]] and lengths.tolist() == [5, 3]


def test_wav_reader_roundtrip(tmp_path):
    pcm = (np.arange(1000) * 31 % 65536 - 32768).astype(np.int16)
    write_wav_pcm16(tmp_path / "a.wav", pcm)
    got, sr = read_wav_pcm16(tmp_path / "a.wav")
    assert sr == 16000 and np.array_equal(got, pcm)
    (tmp_path / "bad.wav").write_bytes(b"nope")
    with pytest.raises(ValueError):
        read_wav_pcm16(tmp_path / "bad.wav")


def test_label_lookup_and_topk_format():
    lm = load_label_map()
    assert len(lm) == 537 and lm[0] == "Speech" and lm[532] == "on"
    p = torch.zeros(537); p[532] = 0.9999; p[0] = 1.0; p[500] = 0.0001
    lines = topk_lines(p, lm, 3)
    assert lines[0].startswith("Speech") and lines[0].endswith("1.0000")
    assert lines[1].startswith("Keyword: on") and lines[1].endswith("0.9999")


from conftest import kws_loop_restatement as _kws_loop_restatement  # noqa: E402


def test_kws_output_transform_matches_reference_loop():
    from uit_mobile_amd.evaluate import output_transform_kws
    g = torch.Generator().manual_seed(0)
    y_pred = torch.rand(64, 537, generator=g)
    y_pred[::3, 527:] *= 0.1                                  # rows without a firing keyword
    y = torch.zeros(64, 537)
    y[torch.arange(64), torch.randint(0, 537, (64,), generator=g)] = 1.0
    for thr in (0.2, 0.5):
        a, ya = output_transform_kws(y_pred, y, thr)
        b, yb = _kws_loop_restatement(y_pred, y, thr)
        assert torch.equal(a, b) and torch.equal(ya, yb)


def test_load_pretrained_filters_by_shape_and_resizes_pos_embed():
    from uit_mobile_amd.evaluate import load_pretrained
    m = models.uit_xxxs(outputdim=527, target_length=102)          # evaluate.py default num_classes = 527
    ck = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xxxs", 5, outputdim=537, target_length=160).items()}
    before = m.outputlayer[1].weight.clone()
    load_pretrained(m, {"model": ck})
    assert torch.equal(m.outputlayer[1].weight, before)            # [537,128] does not fit [527,128]: skipped
    assert torch.equal(m.blocks[2].mlp.fc1.weight, ck["blocks.2.mlp.fc1.weight"])
    assert torch.equal(m.time_pos_embed, ck["time_pos_embed"][..., :6])   # 10 -> 6 entries
    with pytest.raises(AssertionError):
        load_pretrained(m, {"unrelated": torch.zeros(1)})


def test_uploaded_image_follows_the_weights(monkeypatch):
    """Round 4: every call compares a CHEAP stamp of the weights (sum of the tensors' version counters + two storage
    pointers, one pass over a cached tensor list) with the one the uploaded image was packed from -- in-place edits under no_grad,
    nn.init.*, a sub-module's load_state_dict are seen; load_state_dict, .to()/.float() (= _apply) and refresh() drop the handles
    outright.  check_weights=False (a constructor kwarg since round 5; benchmark loops) returns the cached handle without looking."""
    from uit_mobile_amd.models import uit as U
    m = models.uit_xxxs(outputdim=537, target_length=102).eval()

    class H:                                            # stand-in for _Handle
        ptr = 1234

    dev = torch.device("cuda", 0)
    assert m.check_weights and not hasattr(U, "_CHECK_WEIGHTS")     # the default; no environment variable is read any more
    m._handles[0] = (H(), m._stamp())
    assert m._handle(dev) == 1234                       # unchanged weights: the cached handle
    for edit in (lambda: m.norm.weight.mul_(2.0), lambda: torch.nn.init.zeros_(m.blocks[1].mlp.fc2.bias),
                 lambda: m.outputlayer.load_state_dict(m.outputlayer.state_dict()),
                 lambda: torch.nn.utils.vector_to_parameters(torch.nn.utils.parameters_to_vector(m.parameters()), m.parameters())):
        m._handles[0] = (H(), m._stamp())
        with torch.no_grad():
            edit()
        assert m._handles[0][1] != m._stamp()           # the stamp moved: _handle would re-pack (and re-calibrate)
    m._handles[0] = (H(), m._stamp())
    m.float()                                                           # _apply hook
    assert m._handles == {}
    m._handles[0] = (H(), ())
    m.load_state_dict(m.state_dict())
    assert m._handles == {}
    m._handles[0] = (H(), ())
    m.refresh()
    assert m._handles == {}
    m._handles[0] = (H(), ())                           # a stale stamp ...
    m.check_weights = False
    calls = []
    monkeypatch.setattr(m, "_stamp", lambda: calls.append(1) or ())
    assert m._handle(dev) == 1234 and calls == []       # ... is not even looked at with check_weights=False
    assert models.uit_xxxs(outputdim=537, target_length=102, check_weights=False).check_weights is False
    with pytest.raises(ValueError):                     # option names are validated without a live handle (no leak, no poisoned state)
        m.set_option("no_such_option", 1)
    assert "no_such_option" not in m._options


def test_parity_precision_is_part_of_the_interface():
    """precision='f16l6' (f16 operands + MX-fp6 low-order weight term) packs without a GPU; it is a uit_* family mode."""
    m = models.uit_xxxs(outputdim=537, target_length=102, precision="f16l6").eval()
    img = m.packed_image()
    assert len(img) > len(models.uit_xxxs(outputdim=537, target_length=102, precision="f16").eval().packed_image())
    with pytest.raises(NotImplementedError):
        models.audio_transformer_h128_d4_m3(outputdim=537, target_length=102, precision="f16l6")
    with pytest.raises(ValueError):
        models.uit_xxxs(precision="f16l7")


def test_other_kwargs_keep_the_reference_state_dict_layout():
    """Round 3: qkv_bias=False, init_values (LayerScale) and init_bn=False are accepted; the state_dict then has exactly the keys the
    REFERENCE's model has for these kwargs (make_golden.py --kwargs loads the same checkpoint into it with strict=True): no
    attn.qkv.bias, no init_bn.*, blocks.N.ls{1,2}.gamma.  center=False changes the frame count rule.  All pack without a GPU."""
    from uit_mobile_amd.synth import make_kwargs_variant_state_dict
    m = models.uit_xxxs(outputdim=537, target_length=102, qkv_bias=False, init_values=0.1, init_bn=False)
    sd = make_kwargs_variant_state_dict("uit_xxxs", 33)
    assert set(m.state_dict()) == set(sd)
    assert float(m.blocks[0].ls1.gamma.detach()[0]) == pytest.approx(0.1)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    assert len(m.eval().packed_image()) > 0
    c = models.uit_xxxs(outputdim=537, target_length=102, center=False)
    assert c.num_frames(16000) == 97 and c.num_crops(16000) == 1 and c.num_crops(17000) == 2
    assert models.uit_xxxs().num_frames(16000) == 101
    assert len(c.eval().packed_image()) > 0


def test_pinned_batch_feeder_orders_batches_and_pads_like_the_collate():
    """PinnedBatchFeeder (round 3): several producer threads fill a ring of int16 batches; the consumer receives
    them in batch order, every clip zero right-padded (dataset.py:183-204) even when its buffer held a longer clip before; a producer
    exception surfaces in the consumer."""
    from uit_mobile_amd.streaming import PinnedBatchFeeder

    def fill(i, out):
        for r in range(out.shape[0]):
            PinnedBatchFeeder.collate(out[r], np.full(100 + r - 5 * (i % 3), i * 10 + r, np.int16))
    f = PinnedBatchFeeder(fill, 11, 4, 128, producers=3, pin=False)
    got = [b.clone().numpy() for b in f]
    assert len(got) == 11
    for i, g in enumerate(got):
        for r in range(4):
            n = 100 + r - 5 * (i % 3)
            assert (g[r, :n] == i * 10 + r).all() and (g[r, n:] == 0).all()

    def bad(i, out):
        if i == 2:
            raise OSError("unreadable clip")
        out[:] = i
    with pytest.raises(OSError):
        list(PinnedBatchFeeder(bad, 5, 2, 16, producers=2, pin=False))


def test_clip_store_feeds_collated_int16_batches(tmp_path):
    """SURVEY 8f-1 input side (dataset.py:16-56, 183-204): clips of different lengths and dtypes from a store -> int16 batches, truncated /
    zero right-padded rows, a short last batch, the clips' own lengths and names beside them -- through the multi-producer feeder
    (unpinned here: no device)."""
    from uit_mobile_amd.dataset import ClipStore, to_int16
    rng = np.random.default_rng(5)
    clips = []
    for i in range(11):
        n = int(rng.integers(50, 400))
        c = rng.integers(-32768, 32767, n).astype(np.int16)
        clips.append(c if i % 3 else (c / 32768.0).astype(np.float32))        # the reference's float form of an int16 clip: exact round trip
        np.save(tmp_path / f"clip_{i:02d}.npy", clips[-1])
    assert np.array_equal(to_int16((np.array([-32768, -1, 0, 1, 32767]) / 32768.0).astype(np.float32)), [-32768, -1, 0, 1, 32767])
    for store in (ClipStore.from_npy_dir(tmp_path), ClipStore.from_arrays(clips, [f"clip_{i:02d}" for i in range(11)])):
        B, T = 4, 256
        feeder = store.feeder(B, T, producers=3, pin=False)
        index = store.batch_index(B)
        assert [(f, c) for f, c, _ in index] == [(0, 4), (4, 4), (8, 3)] and index[2][2] == ["clip_08", "clip_09", "clip_10"]
        got = [b.numpy().copy() for b in feeder]
        assert len(got) == 3
        for (first, count, names), batch in zip(index, got):
            assert batch.shape == (B, T) and batch.dtype == np.int16
            for j in range(count):
                want = to_int16(clips[first + j])[:T]
                assert np.array_equal(batch[j, :len(want)], want) and not batch[j, len(want):].any()
            assert not batch[count:].any()
            assert list(store.lengths(first)) == [len(clips[first + j]) for j in range(count)]
    with pytest.raises(ImportError):
        ClipStore.from_hdf5_table([("a", "b.h5")])          # h5py is not part of this image: a clear error, not a silent fallback


def test_shorter_window_is_zero_padded_to_n_fft_at_pack_time():
    """win_size < n_fft (models/uit.py:290): the packed front-end window table holds the state_dict's window centred in 512 zeros
    (what torch.stft does with win_length < n_fft); win_size > 512 and other n_fft are not the MFMA front-end's."""
    import struct
    m = models.uit_xxxs(outputdim=537, target_length=102, win_size=400, precision="f16").eval()
    assert tuple(m.state_dict()["front_end.0.spectrogram.window"].shape) == (400,)
    img = m.packed_image()
    from mfma_emu import parse_header
    h = parse_header(img)
    tab = np.frombuffer(img, np.float32, 512, h["off_window"])
    want = np.zeros(512, np.float32)
    want[56:456] = m.state_dict()["front_end.0.spectrogram.window"].numpy()
    assert np.array_equal(tab, want)
    # (round 6) a window longer than 512 or another n_fft is outside the MFMA front-end's geometry: an explicit MFMA precision is refused by
    # name, the default runs on the general engine (which validates win_size <= n_fft when its handle is created, as torch.stft does)
    with pytest.raises(NotImplementedError, match="MFMA-kernel mode"):
        models.uit_xxxs(outputdim=537, target_length=102, win_size=640, precision="f16")
    with pytest.raises(NotImplementedError, match="MFMA-kernel mode"):
        models.uit_xxxs(outputdim=537, target_length=102, n_fft=1024, win_size=1024, precision="f16")
    assert models.uit_xxxs(outputdim=537, target_length=102, n_fft=1024, win_size=1024).general


def test_deviation_stats_keep_a_nan_in_the_middle():
    """the running maximum of the held-out comparison (`if (!(d <= mx)) mx = d`) kept a NaN only until the next finite element
    replaced it: d = [1, NaN, 0.5] came out as 0.5, and the acceptance check, the absolute budget and uitk_self_check trusted that number.
    uitk_deviation_stats is that comparison (host only): one non-finite deviation makes BOTH results NaN, whatever follows."""
    L = _lib.load()

    def stats(got, ref):
        got = np.ascontiguousarray(got, np.float32)
        ref = np.ascontiguousarray(ref, np.float64)
        out = (ctypes.c_double * 2)()
        assert L.uitk_deviation_stats(got.ctypes.data, ref.ctypes.data, got.size, out) == _lib.UITK_OK
        return out[0], out[1]

    mx, rms = stats([1.0, 2.0, 2.5], [0.0, 2.0, 2.0])
    assert mx == 1.0 and abs(rms - np.sqrt((1.0 + 0.25) / 3)) < 1e-12
    for bad in (np.nan, np.inf, -np.inf):
        for pos in (0, 1, 2):                                    # first, MIDDLE (the case that slipped through), last
            got = [1.0, 1.0, 0.5]
            got[pos] = bad
            mx, rms = stats(got, [0.0, 0.0, 0.0])
            assert np.isnan(mx) and np.isnan(rms), (bad, pos, mx, rms)
            assert not (mx <= 6e-4)                              # what the budget comparison sees
    mx, rms = stats([1.0], [np.nan])                             # a non-finite REFERENCE is no better
    assert np.isnan(mx)
    assert L.uitk_deviation_stats(None, None, 0, None) == _lib.ERR_INVALID_ARG


class _FakeLadderLib:
    """Stand-in for libuitk.so in the parity ladder of models.uit (no GPU): every handle is a dict of what the library would report."""

    def uitk_get_calib_check(self, ptr, rep):
        for i, v in enumerate(ptr.state["check"]):
            rep[i] = v
        return 0

    def uitk_get_option(self, ptr, name, ref):
        ref._obj.value = {b"parity_escalate_to": ptr.state["esc"], b"calib_budget_e6": 600}[name]
        return 0

    def uitk_self_check(self, ptr, w, stream, out2):
        out2[0], out2[1] = ptr.state["self_check"]
        return 0


class _FakePtr:
    def __init__(self, state):
        self.state = state

    def __bool__(self):                                          # _Handle.__del__ must not hand this to uitk_destroy
        return False


def _ladder(monkeypatch, precision, pooling, first, rungs, **kw):
    """run UITBase._hold_to_budget on fake handles: `first` = state of the calibrated handle, `rungs` = {precision name: state}"""
    import uit_mobile_amd.models.uit as U
    m = models.uit_xxs(outputdim=537, target_length=102, pooling=pooling, precision=precision, **kw).eval()
    built = []

    def create(L, w, prec):
        h = U._Handle(_FakePtr(dict(rungs[prec])))
        h.mode = {"precision": prec, "requested": precision, "escalated": False, "reason": None, "check": None}
        built.append(prec)
        return h

    monkeypatch.setattr(m, "_create", create)
    monkeypatch.setattr(U.UITBase, "_calibrate_handle", staticmethod(lambda L, handle, w, x, stream: None))
    h0 = U._Handle(_FakePtr(dict(first)))
    start = {"auto": "f16a" if pooling == "token" else "f16c"}.get(precision, precision)
    h0.mode = {"precision": start, "requested": precision, "escalated": False, "reason": None, "check": None}
    out = m._hold_to_budget(_FakeLadderLib(), ctypes.c_int(0), h0, start, None, None, 0)
    return out, built, h0


def test_ladder_never_installs_a_rung_that_is_not_finite(monkeypatch):
    """a rung whose self check is NaN failed `sc[0] >= cur_dev` and was TAKEN in place of a finite build.  Now a non-finite
    rung is worse than anything finite: skipped, reported, the ladder goes on; with the fp32 rung switched off the finite build stays and
    is reported outside its budget."""
    from uit_mobile_amd.models.uit import UitkParityWarning
    P = _lib
    nan = float("nan")
    over = {"check": (1.5e-3, 9e-4, 1e-4, 1e-4, 1.0), "esc": P.PREC_F16AX}               # calibrated f16a: 9e-4 > 6e-4
    rungs = {"f16ax": {"esc": P.PREC_F32, "self_check": (nan, nan), "check": (0,) * 5},
             "f32": {"esc": -1, "self_check": (2e-6, 1e-6), "check": (0,) * 5}}
    with pytest.warns(UitkParityWarning, match="f32"):
        out, built, h0 = _ladder(monkeypatch, "auto", "token", over, rungs)
    assert built == ["f16ax", "f32"] and out.mode["precision"] == "f32" and out.mode["within_budget"] is True
    assert "not finite" in out.mode["reason"] and out.mode["ladder"] == ["f16a", "f32"]
    with pytest.warns(UitkParityWarning, match="kept"):
        out, built, h0 = _ladder(monkeypatch, "auto", "token", over, rungs, fp32_fallback=False)
    assert out is h0 and out.mode["within_budget"] is False and "not finite" in out.mode["reason"]
    # the installed image itself not finite on the held-out clips (sticky NaN from the library): any finite rung is better
    first = {"check": (nan, nan, nan, nan, 0.0), "esc": P.PREC_F16AX}
    rungs = {"f16ax": {"esc": -1, "self_check": (3e-4, 1e-4), "check": (0,) * 5}}
    with pytest.warns(UitkParityWarning, match="f16ax"):
        out, built, _ = _ladder(monkeypatch, "auto", "token", first, rungs)
    assert out.mode["precision"] == "f16ax" and out.mode["within_budget"] is True


def test_explicit_precisions_over_budget_warn(monkeypatch):
    """explicit 'f16ax' (and 'fp8c' / 'fp8w' after a refusal) used to return from the ladder without a word."""
    from uit_mobile_amd.models.uit import UitkParityWarning
    import warnings
    over = {"check": (2e-3, 1.1e-3, 1e-4, 1e-4, 1.0), "esc": _lib.PREC_F32}
    with pytest.warns(UitkParityWarning, match="does not escalate"):
        out, built, h0 = _ladder(monkeypatch, "f16ax", "token", over, {})
    assert out is h0 and built == [] and out.mode["within_budget"] is False and "over the budget" in out.mode["reason"]
    refused = {"check": (4e-2, 5e-2, 1e-2, 1e-2, 0.0), "esc": -1}                         # fp8c: measured mode, calibration refused
    with pytest.warns(UitkParityWarning, match="nearest-rounded"):
        out, _, _ = _ladder(monkeypatch, "fp8c", "mean", refused, {})
    assert out.mode["precision"] == "fp8"
    inside = {"check": (2e-4, 1e-4, 1e-5, 1e-5, 1.0), "esc": -1}
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        out, _, _ = _ladder(monkeypatch, "f16ax", "token", inside, {})
    assert out.mode["within_budget"] is True


# ---- round 6: the general engine's host side (no GPU)
def test_general_geometry_builds_the_reference_key_set_and_names_its_engine():
    """UITBase with a geometry outside the published factories' constructs (it used to be refused), holds the state_dict the reference's
    constructor gives those kwargs (the fixtures' checkpoints load strictly: they were loaded strictly into the reference's own UITBase by
    make_golden.py --general), runs in 'f32' on the general engine, and refuses the MFMA-kernel precisions by name."""
    import numpy as np
    import torch
    import torch.nn as nn
    from general_cases import GENERAL_CASES, state_dict_args
    from uit_mobile_amd.models.uit import UITBase
    from uit_mobile_amd.synth import make_general_state_dict
    from general_cases import SMALL_CASES
    for tag in SMALL_CASES:
        case = GENERAL_CASES[tag]
        kw = dict(case["kwargs"], attention_type={"bneck": "BNeckAttention", "full": "Attention"}[case["attention"]],
                  act_layer={"relu": nn.ReLU, "gelu": nn.GELU}[case["act"]])
        m = UITBase(**kw)
        assert m.general and m.precision == "f32", tag
        sd = make_general_state_dict(**state_dict_args(case))
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        assert m.patch_embed.grid_size == (kw.get("n_mels", 64) // kw["patch_size"], kw["target_length"] // kw["patch_size"])
        with pytest.raises(NotImplementedError, match="MFMA-kernel mode"):
            UITBase(**kw, precision="f16c")
        with pytest.raises(RuntimeError, match="HIP device only"):
            m.eval()(torch.zeros(1, 16000))
    assert not models.uit_xs(target_length=102).general
    assert models.uit_xs().general                       # the constructor's own default target_length (1012 frames: 63 time patches) is a general geometry
    with pytest.raises(NotImplementedError, match="patch_stride"):
        UITBase(patch_size=16, patch_stride=8, embed_dim=128, depth=1, num_heads=2, target_length=102)


def test_general_engine_validates_its_config_before_it_looks_for_a_device():
    """uitk_general_create (no GPU here): a config it cannot run is refused with the limit it breaks; one it can run gets as far as the
    device query (UITK_ERR_NO_DEVICE on this box, or a handle on a GPU box: destroyed again)."""
    L = _lib.load()
    m = models.UITBase(embed_dim=96, depth=1, num_heads=3, mlp_ratio=2.0, patch_size=8, patch_stride=8, target_length=102, outputdim=10)
    w, keep = m._host_weights()

    def create(**over):
        cfg = m._config()
        for k, v in over.items():
            setattr(cfg, k, v)
        out = ctypes.c_void_p()
        rc = L.uitk_general_create(ctypes.byref(cfg), ctypes.byref(w), ctypes.byref(out))
        if rc == _lib.UITK_OK:
            L.uitk_general_destroy(out)
        return rc, _lib.last_error()

    for over, why in ((dict(precision=_lib.PREC_F16), "precision must be UITK_PREC_F32"), (dict(n_fft=511), "n_fft must be even"), (dict(n_fft=3000), "no power of two must be <= 2048"),
                      (dict(win_size=600), "win_size must be 1..n_fft"), (dict(patch_size=6), "multiple of 4"), (dict(embed_dim=100), "multiple of 16"),
                      (dict(mlp_hidden=100), "multiple of 16"), (dict(num_heads=5), "divisible by num_heads"), (dict(pooling=3), "pooling must be"),
                      (dict(n_mels=4), "smaller than one patch"), (dict(target_length=7), "smaller than one patch"), (dict(eval_avg=7), "unknown eval_avg")):
        rc, msg = create(**over)
        assert rc == _lib.ERR_UNSUPPORTED and why in msg, (over, rc, msg)
    rc, msg = create()
    assert rc in (_lib.UITK_OK, _lib.ERR_NO_DEVICE), (rc, msg)
    assert L.uitk_general_create(None, None, None) == _lib.ERR_INVALID_ARG
    assert L.uitk_general_num_frames(None, 16000) == -1 and L.uitk_general_destroy(None) == _lib.UITK_OK
