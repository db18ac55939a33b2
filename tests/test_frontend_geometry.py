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
