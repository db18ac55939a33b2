"""CPU: the front-end kernel's pass / slot / store-address algebra (csrc/frontend.hip), restated in numpy, must place every
frame of every clip exactly where the reference's crop rule (models/uit.py:468-481) says the encoder will look for it:
patch-major operands [clip-crop][token pf*6+pt][mel 16][frame 16].  Frames beyond the patches only feed the amax."""
import numpy as np
import pytest


def kernel_store_map(B, nF, tl, ncrops, nt):
    """Mirror of mel_frontend_kernel's pass geometry and PATCH stores.  Returns {dest element index: (clip, frame, mel)}
    and the set of (clip, frame) pairs whose dB value enters the amax."""
    full, rem = nF >> 4, nF & 15
    per_tail = 16 // rem if rem else 1
    nfull = B * full
    npass = nfull + ((B + per_tail - 1) // per_tail if rem else 0)
    dest, seen = {}, set()
    for p in range(npass):
        is_tail = p >= nfull
        for wave in range(4):
            for r in range(4):                       # row r of the wave's 4x64 result = slot 4*wave + r
                slot = 4 * wave + r
                if not is_tail:
                    b, f = p // full, 16 * (p % full) + slot
                    k0 = p % full
                else:
                    cb = (p - nfull) * per_tail
                    cl, fr = slot // rem, slot % rem
                    if cl >= per_tail or cb + cl >= B:
                        continue                      # dead slot
                    b, f, k0 = cb + cl, 16 * full + fr, full
                seen.add((b, f))
                for mel in range(64):
                    pf, mp = mel >> 4, mel & 15
                    if ncrops == 1:
                        if not is_tail and k0 < nt:
                            d = ((b * 24 + pf * 6 + k0) * 16 + mp) * 16 + 4 * wave + r
                            assert d not in dest
                            dest[d] = (b, f, mel)
                    else:
                        cr0, t0 = f // tl, f % tl
                        if cr0 < ncrops - 1 and t0 < 96:
                            d = (((b * ncrops + cr0) * 24 + pf * 6 + (t0 >> 4)) * 16 + mp) * 16 + (t0 & 15)
                            assert d not in dest
                            dest[d] = (b, f, mel)
                        t1 = f - (nF - tl)
                        if 0 <= t1 < 96:
                            d = (((b * ncrops + ncrops - 1) * 24 + pf * 6 + (t1 >> 4)) * 16 + mp) * 16 + (t1 & 15)
                            assert d not in dest
                            dest[d] = (b, f, mel)
    return dest, seen


def kernel_walk(B, nF, tl, ncrops, nt, G, static_pct=88):
    """Mirror of the round-2 pass loop.  Passes: idx = column * B + clip for the full passes (column-major), then the tail passes.
    Workgroup wg takes idx = wg + r G for its first Rs rounds and the rest is drawn one pass at a time from a counter (here: handed
    out round-robin, any order must do).  In round r wave w transforms frame quad (w + r) & 3 of a full pass; full-pass stores are
    uniform byte base ((b*24 + k)*512 + 8 wq) + lane-invariant byte offset.  Returns (dest map in ELEMENT indices, amax set,
    visited passes)."""
    full, rem = nF >> 4, nF & 15
    per_tail = 16 // rem if rem else 1
    ntail = (B + per_tail - 1) // per_tail if rem else 0
    nfull, npass = B * full, B * full + ntail
    Rs = max(1, npass * static_pct // 100 // G)
    pstatic = Rs * G
    dest, seen, visited = {}, set(), []

    def store_elems(b, f, mel, val):
        pf, mp = mel >> 4, mel & 15
        if ncrops == 1:
            if f < 16 * nt:
                d = (((b * 24 + pf * 6 + (f >> 4)) * 16 + mp) * 16) + (f & 15)
                assert d not in dest
                dest[d] = val
            return
        cr0, t0 = f // tl, f % tl
        if cr0 < ncrops - 1 and t0 < 96:
            d = (((b * ncrops + cr0) * 24 + pf * 6 + (t0 >> 4)) * 16 + mp) * 16 + (t0 & 15)
            assert d not in dest
            dest[d] = val
        t1 = f - (nF - tl)
        if 0 <= t1 < 96:
            d = (((b * ncrops + ncrops - 1) * 24 + pf * 6 + (t1 >> 4)) * 16 + mp) * 16 + (t1 & 15)
            assert d not in dest
            dest[d] = val

    def do_pass(idx, rnd):
        visited.append(idx)
        if idx < nfull:
            ck, cb = idx // B, idx % B
            for wave in range(4):
                wq = (wave + rnd) & 3
                for lane in range(64):
                    st_lane = (((lane >> 4) * 6 * 16 + (lane & 15)) * 16) * 2                    # bytes
                    for r in range(4):
                        f = 16 * ck + 4 * wq + r
                        if lane == 0:
                            seen.add((cb, f))
                        if ncrops == 1:
                            if ck < nt:
                                byte = (cb * 24 + ck) * 512 + 8 * wq + st_lane + 2 * r
                                assert byte // 2 not in dest
                                dest[byte // 2] = (cb, f, lane)
                        else:
                            store_elems(cb, f, lane, (cb, f, lane))
        else:
            t = idx - nfull
            for wave in range(4):
                for r in range(4):
                    slot = 4 * wave + r
                    cl, fr = slot // rem, slot % rem
                    b = t * per_tail + cl
                    if cl >= per_tail or b >= B:
                        continue
                    f = 16 * full + fr
                    seen.add((b, f))
                    for mel in range(64):
                        store_elems(b, f, mel, (b, f, mel))

    rounds = {wg: 0 for wg in range(G)}
    for wg in range(G):                                                    # static rounds
        for r in range(Rs):
            idx = wg + r * G
            if idx < npass:
                do_pass(idx, r)
                rounds[wg] = r + 1
    for k, idx in enumerate(range(pstatic, npass)):                        # drawn passes: whoever asks first
        wg = (7 * k) % G
        do_pass(idx, rounds[wg])
        rounds[wg] += 1
    return dest, seen, visited


def reference_expectation(B, nF, tl):
    """What the encoder must find: per the reference, crops = split(tl) with a short tail replaced by the last tl frames;
    each crop's first (min(frames, tl) // 16) * 16 frames form the patches."""
    ncrops = 1 if nF <= tl else -(-nF // tl)
    starts = [0] if ncrops == 1 else [c * tl for c in range(ncrops - 1)] + [nF - tl]
    nt = min((tl if ncrops > 1 else nF) // 16, 6)
    want = {}
    for b in range(B):
        for cr, s0 in enumerate(starts):
            for pt in range(nt):
                for ti in range(16):
                    f = s0 + 16 * pt + ti
                    for mel in range(64):
                        d = (((b * ncrops + cr) * 24 + (mel >> 4) * 6 + pt) * 16 + (mel & 15)) * 16 + ti
                        want[d] = (b, f, mel)
    return ncrops, nt, want


@pytest.mark.parametrize("B,T,hop,tl", [(5, 16000, 160, 102), (3, 16384, 160, 102), (2, 160000, 160, 102), (4, 8000, 160, 102),
                                        (7, 16000, 128, 102), (3, 4099, 160, 102), (1, 33000, 160, 102), (6, 15360, 160, 96)])
def test_every_patch_element_comes_from_the_right_frame(B, T, hop, tl):
    nF = 1 + T // hop
    ncrops, nt, want = reference_expectation(B, nF, tl)
    got, seen = kernel_store_map(B, nF, tl, ncrops, nt)
    assert seen == {(b, f) for b in range(B) for f in range(nF)}        # every frame of every clip feeds the amax, once
    assert got == want                                                   # and every operand is written once, from the right source


@pytest.mark.parametrize("G", [1, 5, 512])
@pytest.mark.parametrize("B,T,hop,tl", [(5, 16000, 160, 102), (3, 16384, 160, 102), (2, 160000, 160, 102), (4, 8000, 160, 102),
                                        (7, 16000, 128, 102), (3, 4099, 160, 102), (6, 15360, 160, 96), (9, 2400, 160, 102)])
def test_persistent_walk_visits_every_pass_once_and_stores_like_the_reference(B, T, hop, tl, G):
    """The specialised full / tail loops of round 2 (uniform base + lane-invariant offset; (cb, ck) stepping; tail_index)."""
    nF = 1 + T // hop
    ncrops, nt, want = reference_expectation(B, nF, tl)
    got, seen, visited = kernel_walk(B, nF, tl, ncrops, nt, G)
    full, rem = nF >> 4, nF & 15
    npass = B * full + ((B + 16 // rem - 1) // (16 // rem) if rem else 0)
    assert sorted(visited) == list(range(npass))
    assert seen == {(b, f) for b in range(B) for f in range(nF)}
    assert got == want


@pytest.mark.parametrize("npass,G,pm", [(51883, 512, 590), (51883, 512, 500), (1000, 512, 590), (513, 512, 590), (7, 2, 590), (9, 4, 999)])
def test_asymmetric_static_split_covers_every_pass_once(npass, G, pm):
    """a.asym_pm: the first-dispatched half of the grid (blockIdx < G/2) walks its own pool [0, nA) with stride G/2, the other half
    the pool [nA, npass): every pass exactly once, for any share."""
    S, nA = G // 2, npass * pm // 1000
    seen = []
    for wg in range(G):
        idx, end = (wg, nA) if wg < S else (nA + wg - S, npass)
        while idx < end:
            seen.append(idx)
            idx += S
    assert sorted(seen) == list(range(npass))


def test_front_end_tables_and_split_algebra_reproduce_the_oracle_mel():
    """The kernel's arithmetic, restated with numpy on the PACKED tables (window, W256 twiddles as [p][k1], W512 as
    (cos, sin), banded filterbank x 1/4): 256-point complex FFT of the even/odd-packed frame as 16x16 Cooley-Tukey,
    the real-FFT split exactly as frontend.hip writes it (4|X|^2), block-banded mel, 10 log10 -> must equal the oracle's
    (= the reference's) dB mel."""
    import struct
    import torch
    from mfma_emu import HDR_FMT, HDR_FIELDS
    from oracle.oracle import Oracle
    from uit_mobile_amd.synth import make_state_dict, synth_pcm_i16
    from uit_mobile_amd import models
    sd = make_state_dict("uit_xxxs", 21)
    m = models.uit_xxxs(outputdim=537, target_length=102)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    img = m.eval().packed_image()
    h = dict(zip(HDR_FIELDS, struct.unpack_from(HDR_FMT, img, 0)))
    window = np.frombuffer(img, np.float32, 512, h["off_window"]).astype(np.float64)
    tw256 = np.frombuffer(img, np.float32, 512, h["off_tw256"]).astype(np.float64).reshape(16, 16, 2)   # [p][k1] (cos, -sin)
    tw512 = np.frombuffer(img, np.float32, 256, h["off_tw512"]).astype(np.float64).reshape(128, 2)      # (cos, sin)
    S = h["mel_steps"]
    klo = np.frombuffer(img, np.int32, 64, h["off_mel_b"])
    btab = np.frombuffer(img, np.float32, S * 64, h["off_mel_b"] + 256).astype(np.float64).reshape(S // 4, 64, 4)
    pcm = synth_pcm_i16(2, 16000, seed=5)
    x = pcm.astype(np.float64) / 32768.0
    T = x.shape[1]
    o = Oracle(sd)
    nF = o.n_frames(T)
    want, amax = o.melspec_db(pcm)                                     # [B][64][nF], floor applied
    got = np.empty((2, 64, nF))
    for b in range(2):
        for f in range(nF):
            idx = f * 160 - 256 + np.arange(512)
            idx = np.where(idx < 0, -idx, idx)
            idx = np.where(idx >= T, 2 * (T - 1) - idx, idx)           # reflect, edge not repeated
            fr = x[b, idx] * window
            z = (fr[0::2] + 1j * fr[1::2]).reshape(16, 16)             # z[m1][p], m = 16 m1 + p
            y = np.fft.fft(z, axis=0).T                                # [p][k1]: 16-point FFT over m1
            y = y * (tw256[..., 0] + 1j * tw256[..., 1])               # W256^(p k1)
            Z = np.fft.fft(y, axis=0)                                  # [k2][k1]: over p -> Z[k1 + 16 k2]
            Z = Z.reshape(256)                                         # index 16 k2 + k1
            P = np.zeros(272)
            for k in range(0, 128):
                za, zb = Z[k], Z[(256 - k) & 255]
                Ar, Ai = za.real + zb.real, za.imag - zb.imag
                Br, Bi = za.real - zb.real, za.imag + zb.imag
                c, s = tw512[k]
                Tr, Ti = c * Bi - s * Br, -s * Bi - c * Br
                P[k] = (Ar + Tr) ** 2 + (Ai + Ti) ** 2                 # 4 |X[k]|^2
                P[256 - k] = (Ar - Tr) ** 2 + (Ti - Ai) ** 2
            P[128] = 4.0 * abs(Z[128]) ** 2
            mel = np.zeros(64)
            for st in range(S):
                mel += P[klo + st] * btab[st >> 2, :, st & 3]
            got[b, :, f] = 10.0 * np.log10(np.maximum(mel, 1e-10))
    got = np.maximum(got, amax - 120.0)
    from conftest import assert_mel_close
    assert_mel_close(got.astype(np.float32), want, scale=1.0)
