"""CPU: the packed parameter image (csrc/pack.cpp via uitk_pack_weights) + the kernel's index algebra, executed by
the numpy MFMA-layout emulator (tests/mfma_emu.py), must reproduce the oracle.  No GPU, no compute call into
libuitk (uitk_pack_weights is host-only)."""
import numpy as np
import pytest
import torch

from conftest import centred
from mfma_emu import EncoderEmu, parse_header
from oracle.oracle import Oracle
from uit_mobile_amd.synth import make_state_dict, synth_pcm_i16
from uit_mobile_amd import models


def _model(name, seed, prec, **kw):
    sd = make_state_dict(name, seed)
    m = getattr(models, name)(outputdim=537, target_length=102, precision=prec, **kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.eval(), sd


@pytest.mark.parametrize("prec,tol_tok,tol_logit", [("f16", 6e-3, 1e-3), ("bf16", 5e-2, 8e-3)])
def test_emulated_kernel_matches_oracle(prec, tol_tok, tol_logit):
    m, sd = _model("uit_xxxs", 11, prec)
    emu = EncoderEmu(m.packed_image())
    pcm = synth_pcm_i16(2, 16000, seed=7)
    o = Oracle(sd)
    ref = o.forward(pcm, want_stages=True)
    mel_raw, amax = o.melspec_db(pcm)                   # floor already applied by the oracle; re-clamping is a no-op
    mel2 = np.ascontiguousarray(mel_raw.transpose(0, 2, 1))     # [2][frames][64]
    logits, stages = emu.run_wave(mel2, amax - 120.0, nt=6)
    d_tok = np.abs(centred(stages) - centred(ref["stages"][:, 0])).max(axis=(0, 2, 3))
    assert np.abs(stages.mean(-1)).max() < 1e-3          # the kernel's stream is centred (rounding-level means)
    assert d_tok[0] < tol_tok, d_tok
    assert d_tok.max() < 4 * tol_tok, d_tok
    assert np.abs(logits - ref["logits"][:, 0]).max() < tol_logit


def test_emulated_kernel_short_clip_masks_tokens():
    m, sd = _model("uit_xxxs", 11, "f16")
    emu = EncoderEmu(m.packed_image())
    pcm = synth_pcm_i16(2, 8000, seed=8)                # 51 frames -> nt = 3
    o = Oracle(sd)
    ref = o.forward(pcm, want_stages=True)
    mel_raw, amax = o.melspec_db(pcm)
    mel2 = np.ascontiguousarray(mel_raw.transpose(0, 2, 1))
    logits, stages = emu.run_wave(mel2, amax - 120.0, nt=3)
    # oracle tokens are f*3+t; emulator tokens are f*6+t with t<3 valid
    got = stages[:, :, :, :].reshape(2, -1, 4, 6, 128)[:, :, :, :3].reshape(2, -1, 12, 128)
    assert np.abs(centred(got) - centred(ref["stages"][:, 0])).max() < 2e-2
    assert np.abs(logits - ref["logits"][:, 0]).max() < 1e-3


@pytest.mark.parametrize("kind", ["htk", "dense", "fmin2000", "fmin4000", "fmin6000", "fmax4000"])
def test_front_end_filterbank_table_reconstructs_fb(kind):
    """pack.cpp's banded B-operand table for v_mfma_f32_4x4x1 ([64 lane first-bins][S/4][64 lanes][4 steps], x 1/4):
    walking every 4-mel group's window must reproduce the state_dict's mel_scale.fb exactly, windows start on
    multiples of 4 bins and end inside the 272-column power row; HTK fits the on-chip fast path (S <= 52)."""
    import struct
    from mfma_emu import HDR_FMT, HDR_FIELDS
    kw = {"fmin2000": dict(f_min=2000), "fmin4000": dict(f_min=4000), "fmin6000": dict(f_min=6000),
          "fmax4000": dict(f_max=4000)}.get(kind, {})
    m, sd = _model("uit_xxxs", 12, "f16", **kw)
    fb = np.array(sd["front_end.0.mel_scale.fb"], dtype=np.float32)
    if kw:
        # f_min / f_max are reference kwargs (models/uit.py:291-292): narrow bands that start high in frequency; the
        # window of such a group is clamped down to stay inside the power row and must still cover its whole band
        fb = getattr(models, "uit_xxxs")(outputdim=537, target_length=102, **kw).state_dict()["front_end.0.mel_scale.fb"].numpy().copy()
        assert not np.array_equal(fb, np.array(sd["front_end.0.mel_scale.fb"]))
        sd = dict(sd); sd["front_end.0.mel_scale.fb"] = fb
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    if kind == "dense":
        fb = (fb + 0.001).astype(np.float32)
        sd = dict(sd); sd["front_end.0.mel_scale.fb"] = fb
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    img = m.packed_image()
    h = dict(zip(HDR_FIELDS, struct.unpack_from(HDR_FMT, img, 0)))
    S, off = h["mel_steps"], h["off_mel_b"]
    klo = np.frombuffer(img, np.int32, 64, off)
    tab = np.frombuffer(img, np.float32, S * 64, off + 256).reshape(S // 4, 64, 4)
    assert S % 4 == 0 and np.all(klo % 4 == 0) and np.all(klo >= 0) and np.all(klo + max(S, 52) <= 272)
    assert np.all(klo.reshape(16, 4) == klo.reshape(16, 4)[:, :1])          # one window per 4-mel group
    if S <= 52:
        # LDS bank spreading: the four mel groups served in one 16-lane slice of a ds_read_b128 start on distinct residues mod 16
        for sg in ([0, 3, 5, 6], [1, 2, 4, 7], [8, 11, 13, 14], [9, 10, 12, 15]):
            assert len({int(klo[4 * g]) % 16 for g in sg}) == 4, (kind, [int(klo[4 * g]) for g in sg])
    assert (S <= 52) == (kind != "dense")
    rec = np.zeros_like(fb)
    for st in range(S):
        k = klo + st
        ok = k < 257
        rec[k[ok], np.arange(64)[ok]] = 4.0 * tab[st >> 2, :, st & 3][ok]
        assert np.all(tab[st >> 2, :, st & 3][~ok] == 0.0)
    assert np.array_equal(rec, fb)


def test_fp8_weight_stream_reconstructs_the_folded_weights():
    """precision='fp8' (config #5): the fragment8 layout of pack.cpp (uitk_dev.h) -- two lane-linear 1 KiB halves, byte j of
    lane (c, g) = W[16 tile + c][16 (j >> 2) + 4 g + (j & 3)], MX scale byte of hardware block beta supplied by lane (row, beta),
    beta = 2 (j >> 4) + (g >> 1) -- decoded on the CPU must give back the folded fc1 weights to e4m3 precision (2^-4 of the
    block maximum), with every block's maximum inside the format's top binade (scales chosen as tight as the format allows)."""
    import struct
    from mfma_emu import HDR_FMT, HDR_FIELDS
    m, sd = _model("uit_xxxs", 12, "fp8")
    img = m.packed_image()
    h = dict(zip(HDR_FIELDS, struct.unpack_from(HDR_FMT, img, 0)))
    assert h["precision"] == 2
    STAGE, FRAG8, PE_ST, BLK8 = 16384, 2048, 8, 8
    assert h["off_bias"] - h["off_wstream"] >= (PE_ST + 4 * BLK8 + 4) * STAGE
    lut = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).to(torch.float32).numpy()
    w2 = np.asarray(sd["blocks.0.norm2.weight"], np.float64)
    W1 = np.asarray(sd["blocks.0.mlp.fc1.weight"], np.float64) * w2[None]              # folded (pack.cpp Wfc1)
    s1 = h["off_wstream"] + (PE_ST + 1) * STAGE                                          # stage 1 of block 0: proj + MLP scales
    for q in range(3):
        st = h["off_wstream"] + (PE_ST + 2 + 2 * q) * STAGE
        for i in range(8):
            t = 8 * q + i
            raw = np.frombuffer(img, np.uint8, FRAG8, st + i * FRAG8).reshape(2, 64, 16)
            sc = np.frombuffer(img, np.uint8, 256, s1 + 8192 + (t >> 2) * 256).reshape(64, 4)[:, t & 3].astype(np.int32) - 127
            rec = np.zeros((16, 128))
            for lane in range(64):
                c, g = lane & 15, lane >> 4
                for j in range(32):
                    beta = 2 * (j >> 4) + (g >> 1)
                    e = sc[beta * 16 + c]                                              # lane (row c, g' = beta) holds the scale
                    rec[c, 16 * (j >> 2) + 4 * g + (j & 3)] = lut[raw[j >> 4, lane, j & 15]] * 2.0 ** e
            want = W1[16 * t:16 * t + 16]
            k = np.arange(128)
            blk = 2 * (k >> 6) + ((k >> 3) & 1)
            for b in range(4):
                mx = np.abs(want[:, blk == b]).max(1, keepdims=True)
                assert np.all(np.abs(rec[:, blk == b] - want[:, blk == b]) <= mx * 2.0 ** -4 + 1e-12)
                e = sc[b * 16 + np.arange(16)][:, None]
                assert np.all(mx / 2.0 ** e <= 448.0) and np.all(mx / 2.0 ** e > 224.0)


def test_f16x2_weight_stream_holds_hi_plus_lo_of_every_block_weight():
    """precision='f16x2': 28 stages per block (uitk_dev.h), every stage holding the hi AND the lo fragments of its rows.  Decoding the
    universal fragment order (lane (c, g), element j <-> W[16 ft + c][32 s + 16 (j >> 2) + 4 g + (j & 3)]) must give hi + lo = the
    folded fp64 weight to 2^-21 of its magnitude (two fp16 terms), and hi must be an fp16 rounding of it (within one ulp: fc2's hi is
    rounded structurally)."""
    import struct
    from mfma_emu import HDR_FMT, HDR_FIELDS
    m, sd = _model("uit_xxxs", 12, "f16x2")
    img = m.packed_image()
    h = dict(zip(HDR_FIELDS, struct.unpack_from(HDR_FMT, img, 0)))
    assert h["precision"] == 3
    STAGE, FRAG, PE_ST, BLKX = 16384, 1024, 8, 28
    assert h["off_bias"] - h["off_wstream"] >= (PE_ST + 4 * BLKX + 4) * STAGE

    def frag(stage, f):                                   # -> [16 rows][32 k] of one fragment
        raw = np.frombuffer(img, np.float16, 512, h["off_wstream"] + stage * STAGE + f * FRAG).reshape(64, 8).astype(np.float64)
        out = np.zeros((16, 32))
        for lane in range(64):
            c, g = lane & 15, lane >> 4
            for j in range(8):
                out[c, 16 * (j >> 2) + 4 * g + (j & 3)] = raw[lane, j]
        return out

    g = lambda k: np.asarray(sd[k], np.float64)
    scale = (128 // 2) ** -0.5 * 1.4426950408889634
    for b in (0, 3):
        p = f"blocks.{b}."
        Wq = g(p + "attn.qkv.weight") * g(p + "norm1.weight")[None]
        Wq[:32] *= scale
        Wp = g(p + "attn.proj.weight"); Wp = Wp - Wp.mean(0, keepdims=True)
        W1 = g(p + "mlp.fc1.weight") * g(p + "norm2.weight")[None]
        W2 = g(p + "mlp.fc2.weight"); W2 = W2 - W2.mean(0, keepdims=True)
        st = PE_ST + b * BLKX

        def check(want, stage, f_hi, f_lo):
            hi, lo = frag(stage, f_hi), frag(stage, f_lo)
            # lo is often an fp16 subnormal (|W| ~ 4e-3 -> lo ~ 1e-6): absolute resolution 2^-24, still 60x below hi's rounding step
            assert np.all(np.abs(hi + lo - want) <= np.abs(want) * 2.0 ** -21 + 2.0 ** -25), (stage, f_hi)
            assert np.all(np.abs(hi - want) <= np.abs(want) * 2.0 ** -10 + 2.0 ** -24)          # hi: an fp16 neighbour of W

        for t in range(6):                                 # qkv row tiles: stages 0..2, half t & 1: hi s0..3, lo s0..3
            for s in range(4):
                check(Wq[16 * t:16 * t + 16, 32 * s:32 * s + 32], st + t // 2, (t & 1) * 8 + s, (t & 1) * 8 + 4 + s)
        for ft in range(8):                                # proj: K = 32, one k-step
            check(Wp[16 * ft:16 * ft + 16, :32], st + 3, ft, 8 + ft)
        for ch in (0, 5, 11):
            for i in range(2):
                for s in range(4):
                    check(W1[32 * ch + 16 * i:32 * ch + 16 * i + 16, 32 * s:32 * s + 32], st + 4 + 2 * ch, i * 8 + s, i * 8 + 4 + s)
            for ft in range(8):
                check(W2[16 * ft:16 * ft + 16, 32 * ch:32 * ch + 32], st + 5 + 2 * ch, ft, 8 + ft)


def test_f16x2_full_attention_stream_holds_hi_plus_lo_of_every_block_weight():
    """precision='f16x2' on the full-width-attention factories (round 3): 40 stages per block (uitk_dev.h, BLOCK_STAGES_X2_ATT).  A q / k / v
    row group of a head takes two stages of two d-tiles, tile slot dl = fragments dl*8 + s (hi), dl*8 + 4 + s (lo); proj four stages
    of two output tiles; the MLP as in the bottleneck layout.  hi + lo must give the folded fp64 weight to 2^-21."""
    import struct
    from mfma_emu import HDR_FMT, HDR_FIELDS
    from uit_mobile_amd.synth import make_trained_like_state_dict
    name = "audio_transformer_h128_d4_m3"
    sd = make_trained_like_state_dict(name, 7, attention="full")
    m = getattr(models, name)(outputdim=537, target_length=102, precision="f16x2")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    img = m.eval().packed_image()
    h = dict(zip(HDR_FIELDS, struct.unpack_from(HDR_FMT, img, 0)))
    assert h["precision"] == 3
    STAGE, FRAG, PE_ST, BLKX = 16384, 1024, 8, 40
    assert h["off_bias"] - h["off_wstream"] >= (PE_ST + 4 * BLKX + 4) * STAGE

    def frag(stage, f):
        raw = np.frombuffer(img, np.float16, 512, h["off_wstream"] + stage * STAGE + f * FRAG).reshape(64, 8).astype(np.float64)
        out = np.zeros((16, 32))
        for lane in range(64):
            c, g = lane & 15, lane >> 4
            for j in range(8):
                out[c, 16 * (j >> 2) + 4 * g + (j & 3)] = raw[lane, j]
        return out

    g = lambda k: np.asarray(sd[k], np.float64)
    scale = (128 // 2) ** -0.5 * 1.4426950408889634
    for b in (0, 3):
        p = f"blocks.{b}."
        Wq = g(p + "attn.qkv.weight") * g(p + "norm1.weight")[None]
        Wq[:128] *= scale
        Wp = g(p + "attn.proj.weight"); Wp = Wp - Wp.mean(0, keepdims=True)
        W1 = g(p + "mlp.fc1.weight") * g(p + "norm2.weight")[None]
        W2 = g(p + "mlp.fc2.weight"); W2 = W2 - W2.mean(0, keepdims=True)
        st = PE_ST + b * BLKX

        def check(want, stage, f_hi, f_lo):
            hi, lo = frag(stage, f_hi), frag(stage, f_lo)
            assert np.all(np.abs(hi + lo - want) <= np.abs(want) * 2.0 ** -21 + 2.0 ** -25), (stage, f_hi)
            assert np.all(np.abs(hi - want) <= np.abs(want) * 2.0 ** -10 + 2.0 ** -24)

        for hd in range(2):
            for which in range(3):
                for dt in range(4):
                    r0 = which * 128 + hd * 64 + 16 * dt
                    for s in range(4):
                        check(Wq[r0:r0 + 16, 32 * s:32 * s + 32], st + 2 * (3 * hd + which) + dt // 2, (dt & 1) * 8 + s, (dt & 1) * 8 + 4 + s)
        for ft in range(8):
            for s in range(4):
                check(Wp[16 * ft:16 * ft + 16, 32 * s:32 * s + 32], st + 12 + ft // 2, (ft & 1) * 8 + s, (ft & 1) * 8 + 4 + s)
        for ch in (0, 6, 11):
            for i in range(2):
                for s in range(4):
                    check(W1[32 * ch + 16 * i:32 * ch + 16 * i + 16, 32 * s:32 * s + 32], st + 16 + 2 * ch, i * 8 + s, i * 8 + 4 + s)
            for ft in range(8):
                check(W2[16 * ft:16 * ft + 16, 32 * ch:32 * ch + 32], st + 17 + 2 * ch, ft, 8 + ft)


def test_structured_rounding_preserves_group_sums_of_the_conv_weights():
    """pack.cpp round_structured: the packed patch-embed weights are 16-bit neighbours of the exact (column-centred) conv
    weights -- never more than one step away -- and inside every group of 16 frame-columns of a mel row the SUMMED rounding
    error is several times smaller than round-to-nearest leaves it (the error cannot see a spectrum that is constant along the
    patch's frames), at nearly unchanged per-element error."""
    import struct
    from mfma_emu import HDR_FMT, HDR_FIELDS
    m, sd = _model("uit_xxxs", 12, "f16")
    img = m.packed_image()
    h = dict(zip(HDR_FIELDS, struct.unpack_from(HDR_FMT, img, 0)))
    W = np.asarray(sd["patch_embed.proj.weight"], np.float64).reshape(128, 256)
    W = W - W.mean(0, keepdims=True)                                       # residual centring
    Q = np.zeros_like(W)
    for i in range(4):                                                     # the four high-order stages
        for f in range(16):
            raw = np.frombuffer(img, np.float16, 512, h["off_wstream"] + (i * 16 + f) * 1024).reshape(64, 8).astype(np.float64)
            ft, s = 2 * i + f // 8, f % 8
            for lane in range(64):
                g, c = lane >> 4, lane & 15
                for j in range(8):
                    Q[16 * ft + c, (2 * s + (g >> 1)) * 16 + 8 * (g & 1) + j] = raw[lane, j]
    rtn = W.astype(np.float16).astype(np.float64)
    step = np.abs(np.spacing(W.astype(np.float16))).astype(np.float64)
    assert np.all(np.abs(Q - W) <= step * 2.0001)                          # a neighbour (the step doubles across a binade boundary)
    assert np.mean(np.abs(Q - W) > step * 1.0001) < 1e-3
    assert np.mean(Q != rtn) < 0.25                                        # a minority of roundings (the nearest ties) was flipped
    gs = lambda A: (A - W).reshape(128, 16, 16).sum(-1)                    # summed error per (row, mel row)
    assert np.sqrt((gs(Q) ** 2).mean()) < 0.25 * np.sqrt((gs(rtn) ** 2).mean())
    assert np.sqrt(((Q - W) ** 2).mean()) < 1.1 * np.sqrt(((rtn - W) ** 2).mean())


@pytest.mark.parametrize("seed", [205, 207])
def test_emulated_f16l6_low_order_term_cuts_the_weight_rounding_error(seed):
    """precision f16l6 (round 3): the packed fragment6 stream (e2m3 fields, row scales in the bias table, variable-size stages), the
    fp6 / fp4 activation copies and the K = 128 MFMA lane map (tools/probe/mx6_layout.hip), executed by the emulator on a
    trained-like checkpoint, must be CLOSER to the oracle than the same emulation of precision f16 -- i.e. the low-order term is
    applied to the right rows / columns / scales (a misplaced correction adds error instead of removing it)."""
    from uit_mobile_amd.synth import make_trained_like_state_dict
    sd = make_trained_like_state_dict("uit_xxxs", seed)
    pcm = synth_pcm_i16(2, 16000, seed=7)
    o = Oracle(sd)
    ref = o.forward(pcm, want_stages=True)
    mel_raw, amax = o.melspec_db(pcm)
    mel2 = np.ascontiguousarray(mel_raw.transpose(0, 2, 1))
    err = {}
    for prec in ("f16", "f16l6"):
        m = models.uit_xxxs(outputdim=537, target_length=102, precision=prec)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        logits, stages = EncoderEmu(m.eval().packed_image()).run_wave(mel2, amax - 120.0, nt=6)
        err[prec] = float(np.abs(logits - ref["logits"][:, 0]).max())
        assert np.abs(stages.mean(-1)).max() < 1e-3
    print(seed, err)
    assert err["f16l6"] < 0.6 * err["f16"] and err["f16l6"] < 4e-4


@pytest.mark.parametrize("factory,kw", [("uit_xxxs", dict(precision="f16")), ("uit_xxxs", dict(precision="bf16")), ("uit_xxxs", dict(precision="fp8")),
                                        ("uit_xxxs", dict(precision="f16x2")), ("uit_xxxs", dict(precision="f16l6")),
                                        ("uit_xxxs", dict(precision="f16", pooling="dm")), ("uit_xxxs", dict(precision="f16l6", pooling="token")),
                                        ("audio_transformer_h128_d4_m3", dict(precision="f16")), ("audio_transformer_h128_d4_m3", dict(precision="f16x2")),
                                        ("audio_transformer_h128_d4_m3_relu", dict(precision="f16", pooling="token")), ("audio_transformer_h128_d4_m3_relu", dict(precision="bf16"))])
def test_everything_added_to_the_residual_stream_is_centred_at_pack_time(factory, kw):
    """The kernel's LayerNorm is mean-free (x * rsqrt(E[x^2] + eps), encoder.hip ln_stats): correct only while pack.cpp removes the
    channel mean of EVERYTHING that is ever added to the residual stream.  Pack-time invariant, checked on the packed image for every
    precision / attention / pooling family: the rows of the conv-bias + positional table (incl. the cls row), the proj and
    fc2 bias vectors of every block -- exactly (fp32 tables) -- and, for the 16-bit single-weight layouts, the column sums of the
    ROUNDED proj / fc2 / patch-embed operands over their 128 output channels (zero up to the operand rounding: |sum| <= 128 half-ulps)."""
    import struct
    from mfma_emu import HDR_FMT, HDR_FIELDS, to_f32
    from uit_mobile_amd.synth import make_trained_like_state_dict
    full = factory.startswith("audio_transformer")
    sd = make_trained_like_state_dict(factory, 3, attention="full" if full else "bneck")
    m = getattr(models, factory)(outputdim=537, target_length=102, **kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    img = m.eval().packed_image()
    h = dict(zip(HDR_FIELDS, struct.unpack_from(HDR_FMT, img, 0)))
    peb = np.frombuffer(img, np.float32, 25 * 128, h["off_pe_bias"]).reshape(25, 128)
    assert np.abs(peb.astype(np.float64).mean(1)).max() < 1e-6 and np.abs(peb).max() > 1.0
    b_proj, b_fc2 = (384, 896) if full else (96, 608)
    for blk in range(h["depth"]):
        bias = np.frombuffer(img, np.float32, 1024, h["off_bias"] + blk * 4096)
        for off in (b_proj, b_fc2):
            v = bias[off:off + 128].astype(np.float64)
            assert abs(v.mean()) < 1e-6 and np.abs(v).max() > 0.05, (blk, off)
    prec = kw["precision"]
    if full or prec not in ("f16", "bf16"):
        return
    op = 0 if prec == "f16" else 1
    frag = lambda stage, f: to_f32(np.frombuffer(img, np.uint16, 512, h["off_wstream"] + stage * 16384 + f * 1024).reshape(64, 8), op).astype(np.float64)
    half_ulp = 2.0 ** (-11 if op == 0 else -8)
    # patch embed: fragment (stage i, f) = rows 16 (2i + f/8) .., k-step f % 8; a column = one (lane group, element) slot summed over 16 rows x 8 tiles
    col = np.zeros((8, 4, 8))
    mx = 0.0
    for i in range(4):
        for f in range(16):
            a = frag(i, f).reshape(4, 16, 8)                      # [g][row c][j]
            col[f % 8] += a.sum(1)
            mx = max(mx, np.abs(a).max())
    assert np.abs(col).max() < 128 * half_ulp * mx, (np.abs(col).max(), mx)
    for blk in range(h["depth"]):
        st = 8 + blk * 14
        colp, mxp = np.zeros((4, 8)), 0.0
        for ft in range(8):                                       # proj: K = 32 = one k-step; fragment 8 + ft of stage VP
            a = frag(st + 1, 8 + ft).reshape(4, 16, 8)
            colp += a.sum(1); mxp = max(mxp, np.abs(a).max())
        assert np.abs(colp).max() < 128 * half_ulp * mxp
        for ch in range(12):                                      # fc2 k-step ch: fragments 8 + ft of MLP stage ch
            colf, mxf = np.zeros((4, 8)), 0.0
            for ft in range(8):
                a = frag(st + 2 + ch, 8 + ft).reshape(4, 16, 8)
                colf += a.sum(1); mxf = max(mxf, np.abs(a).max())
            assert np.abs(colf).max() < 128 * half_ulp * mxf


@pytest.mark.parametrize("seed", [205, 207])
def test_emulated_calibrated_rounding_cuts_the_weight_rounding_error(seed):
    """Calibrated rounding (round 4; csrc/pack.cpp: round_gptq, uitk_pack_weights_calibrated): single fp16 weights rounded against the
    operand statistics of a calibration batch.  The statistics come from the numpy restatement (tests/calib_ref.py) on ONE set of clips;
    the packed image, executed by the emulator on OTHER clips of a trained-like checkpoint, must be much closer to the oracle than the
    round-to-nearest image -- with the same kernel dataflow (nothing but the rounding direction of the weights differs) -- and every
    calibrated weight must still be one of the two fp16 neighbours of the exact folded value."""
    from calib_ref import calib_stats
    from uit_mobile_amd.synth import make_trained_like_state_dict
    sd = make_trained_like_state_dict("uit_xxxs", seed)
    stats = calib_stats(sd, synth_pcm_i16(16, 16000, seed=4242), depth=4)
    pcm = synth_pcm_i16(2, 16000, seed=7)                                  # held out
    o = Oracle(sd)
    ref = o.forward(pcm, want_stages=True)
    mel_raw, amax = o.melspec_db(pcm)
    mel2 = np.ascontiguousarray(mel_raw.transpose(0, 2, 1))
    m = models.uit_xxxs(outputdim=537, target_length=102, precision="f16c")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    imgs = {"nearest": m.eval().packed_image(), "calibrated": m.packed_image(calib_stats=stats)}
    assert len(imgs["nearest"]) == len(imgs["calibrated"])
    err = {}
    for kind, img in imgs.items():
        logits, stages = EncoderEmu(img).run_wave(mel2, amax - 120.0, nt=6)
        err[kind] = float(np.abs(logits - ref["logits"][:, 0]).max())
        assert np.abs(stages.mean(-1)).max() < 2e-3
    print(seed, err)
    assert err["calibrated"] < 0.6 * err["nearest"] and err["calibrated"] < 4e-4
    # the two weight streams hold the same matrices up to rounding-sized moves: the error feedback shifts a weight by a fraction of the
    # rounding steps of its row's LARGE weights (many units in the last place of a small weight, never a visible change of the matrix)
    h = EncoderEmu(imgs["nearest"]).h
    n16 = (h["off_bias"] - h["off_wstream"]) // 2
    a = np.frombuffer(imgs["nearest"], np.float16, n16, h["off_wstream"]).astype(np.float64)
    b = np.frombuffer(imgs["calibrated"], np.float16, n16, h["off_wstream"]).astype(np.float64)
    assert 0.02 < np.mean(a != b) < 0.9
    assert np.linalg.norm(a - b) < 1.5e-3 * np.linalg.norm(a) and np.abs(a - b).max() < 2e-3 * np.abs(a).max()


def test_calibrated_packing_rejects_bad_statistics_and_other_precisions():
    import ctypes
    from uit_mobile_amd import _lib
    m, sd = _model("uit_xxxs", 3, "f16c")
    with pytest.raises(ValueError):
        m.packed_image(calib_stats=np.zeros(10))
    L = _lib.load()
    cfg = m._config()
    n = L.uitk_calib_stats_doubles(ctypes.byref(cfg))
    assert n == 256 * 256 + 4 * (2 * 128 * 128 + 32 * 32 + 384 * 384) + 4 * 128        # second moments + the LayerNorm-1 means (round 5)
    # all-zero statistics (no information): the image is plain round-to-nearest of every matrix, and still a valid image
    img0 = m.packed_image(calib_stats=np.zeros(n))
    assert EncoderEmu(img0).h["precision"] == 0
    # round 5: the split-weight precisions take statistics too -- their weights are not re-rounded (same stream), the statistics install
    # the operand centring of the attention (k - E[k], v - E[v] -> biases): all-zero statistics leave the image as it was
    m2, _ = _model("uit_xxxs", 3, "f16l6")
    img6 = m2.packed_image(calib_stats=np.zeros(L.uitk_calib_stats_doubles(ctypes.byref(m2._config()))))
    assert len(img6) == len(m2.packed_image())


def test_fp8_grid_weights_with_and_without_gptq_in_the_emulated_f16_kernel():
    """Round 5: precision 'fp8w' = qkv / fc1 / fc2 weights on the MX e4m3 grid (32 consecutive k per power-of-two
    scale), expanded to fp16 fragments -> the f16 kernel.  (a) Every weight of an MLP stage decodes to an e4m3 value times a power of two
    (<= 4 significant bits).  (b) With the operand statistics the grid rounding is GPTQ (pack.cpp: round_fp8_grid): on a held-out clip
    pair the emulated kernel's logit error drops against nearest-on-the-grid.  (c) The fp8 KERNEL's stream ('fp8c') takes the same
    statistics: its image differs from nearest in the fp8 bytes, on the grid the exact weights' MX scales define.  (d) Statistics with a
    NaN or negative diagonal are refused instead of reaching std::stable_sort."""
    from calib_ref import calib_stats
    from uit_mobile_amd.synth import make_trained_like_state_dict
    sd = make_trained_like_state_dict("uit_xxxs", 203)
    stats = calib_stats(sd, synth_pcm_i16(16, 16000, seed=4242), depth=4)
    pcm = synth_pcm_i16(2, 16000, seed=7)
    o = Oracle(sd)
    ref = o.forward(pcm)
    mel_raw, amax = o.melspec_db(pcm)
    mel2 = np.ascontiguousarray(mel_raw.transpose(0, 2, 1))
    m = models.uit_xxxs(outputdim=537, target_length=102, precision="fp8w")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    imgs = {"nearest": m.eval().packed_image(), "gptq": m.packed_image(calib_stats=stats)}
    err = {}
    for kind, img in imgs.items():
        emu = EncoderEmu(img)
        assert emu.h["precision"] == 5
        logits, _ = emu.run_wave(mel2, amax - 120.0, nt=6)
        err[kind] = (float(np.abs(logits - ref["logits"][:, 0]).max()), float(np.sqrt(np.mean((logits - ref["logits"][:, 0]) ** 2))))
        # (a) MLP stage of block 0 (stage 8 + 2): fc1 + fc2 fragments -- every value has at most 4 significant bits
        w = np.concatenate([emu.frag(8 + 2, f).ravel() for f in range(16)]).astype(np.float64)
        mant, _ = np.frexp(w[w != 0])
        assert np.all(np.abs(mant) * 16 == np.round(np.abs(mant) * 16)), kind
    print("fp8w emulated, (max, rms) logit error vs oracle:", err)
    assert err["gptq"][1] < 0.8 * err["nearest"][1] and err["nearest"][0] < 0.3
    # (c)
    m8 = models.uit_xxxs(outputdim=537, target_length=102, precision="fp8c")
    m8.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    a, b = m8.eval().packed_image(), m8.packed_image(calib_stats=stats)
    h = EncoderEmu.__new__(EncoderEmu); h = parse_header(a)
    assert h["precision"] == 2 and len(a) == len(b)
    s1 = h["off_wstream"] + (8 + 1) * 16384                                  # block 0, stage 1: proj + the MLP's MX scale bytes
    sa = np.frombuffer(a, np.uint8, 12 * 256, s1 + 8192).astype(int); sb = np.frombuffer(b, np.uint8, 12 * 256, s1 + 8192).astype(int)
    # the grid is that of the exact weights' scales; a block whose largest element the error feedback moved below half its range is
    # re-expressed one binade lower (the same values, exactly)
    assert np.all((sb == sa) | (sb == sa - 1)) and np.mean(sb == sa) > 0.9
    st2 = h["off_wstream"] + (8 + 2) * 16384
    da = np.frombuffer(a, np.uint8, 16384, st2); db = np.frombuffer(b, np.uint8, 16384, st2)
    assert 0.02 < np.mean(da != db) < 0.9                                    # rounding-sized moves on the grid
    # (d)
    bad = stats.copy(); bad[0] = np.nan
    with pytest.raises(ValueError):
        m.packed_image(calib_stats=bad)
    bad = stats.copy(); bad[256 * 256 + 5 * 129] = -1.0                       # (a diagonal entry of block 0's LayerNorm-1 moment)
    with pytest.raises(ValueError):
        m.packed_image(calib_stats=bad)


@pytest.mark.parametrize("factory", ["uit_xxxs", "audio_transformer_h128_d4_m3"])
def test_f16ax_runs_on_the_parameter_image_of_f16x2(factory):
    """Precision 'f16ax' (round 5: the top of pooling 'token''s ladder) = the split-weight stream of 'f16x2' under another kernel (split
    activation operands): the packed images must be byte-identical behind the header's precision field, with and without statistics
    (the centring folds follow the kernel precision the image is for: both builds carry the runtime LayerNorm-1 centring)."""
    import ctypes
    from uit_mobile_amd import _lib
    from uit_mobile_amd.synth import make_trained_like_state_dict
    att = "bneck" if factory.startswith("uit") else "full"
    sd = make_trained_like_state_dict(factory, 5, attention=att)
    imgs = {}
    for prec in ("f16x2", "f16ax"):
        m = getattr(models, factory)(outputdim=537, target_length=102, pooling="token", precision=prec)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        n = _lib.load().uitk_calib_stats_doubles(ctypes.byref(m._config()))
        rng = np.random.default_rng(3)
        st = np.zeros(n)
        st[-m.depth * 128:] = 0.3 * rng.standard_normal(m.depth * 128)                     # LayerNorm-1 means: the centring folds move
        imgs[prec] = (m.packed_image(), m.packed_image(calib_stats=st))
    for a, b in zip(imgs["f16x2"], imgs["f16ax"]):
        ha, hb = EncoderEmu(a).h, EncoderEmu(b).h
        assert (ha["precision"], hb["precision"]) == (3, 7) and len(a) == len(b)
        off = 8                                                                               # magic + precision
        assert a[off:] == b[off:]
    assert imgs["f16ax"][0] != imgs["f16ax"][1]                                             # the statistics did something
