"""numpy emulation of the encoder kernel's DATAFLOW on the packed parameter image (test helper, CPU only).

Purpose: pin the host-side packing (csrc/pack.cpp: folds + MFMA fragment order) and the kernel's index algebra
without a GPU.  It executes the same steps as csrc/encoder.hip on "wave registers" shaped [64 lanes, n] using the
documented gfx950 MFMA lane maps (cdna_hip_programming.md section 3):
    16x16x32: A[row = l&15][k = 8*(l>>4)+j], B[k = 8*(l>>4)+j][col = l&15], C/D[row = 4*(l>>4)+r][col = l&15]
    16x16x16: the same with k = 4*(l>>4)+j.
This is NOT a product path and is never compared against itself: tests compare it with the C oracle.
"""
import struct

import numpy as np

LANES = np.arange(64)
C_ = LANES & 15
G_ = LANES >> 4

HDR_FMT = "<I4i12i3iQ12Q"
HDR_FIELDS = (["magic", "precision", "depth", "outputdim", "mel_steps"] + [f"mel_tile_{n}{j}" for n in ("klo", "steps", "base")
              for j in range(4)] + ["_p0", "_p1", "_p2", "total_bytes",
              "off_wstream", "off_bias", "off_bn_sc", "off_bn_sh", "off_pe_bias", "off_fin", "off_head_w",
              "off_head_b", "off_window", "off_tw256", "off_tw512", "off_mel_b"])
FRAG, STAGE = 1024, 16384
PE_HI_STAGES, PE_STAGES, BLOCK_STAGES, BIAS_FLOATS = 4, 8, 14, 1024     # PE: 4 stages of weights + 4 of their low-order terms
BIAS_QKV, BIAS_PROJ, BIAS_FC1, BIAS_FC2 = 0, 96, 224, 608
# precision f16l6 (uitk_dev.h): stage sizes of the variable-size stream, fragment6, row-scale table
FRAG6, L6_QK, L6_VP, L6_M, L6_MB, L6_SCALES = 1536, 16384, 16384 + 8 * 1024, 16384 + 2 * 1536, 16384 + 10 * 1536, 736
L6_BLOCK = L6_QK + L6_VP + 9 * L6_M + 3 * L6_MB


def l6_mlp_offset(ch):
    return L6_QK + L6_VP + ch * L6_M + (ch // 4) * (L6_MB - L6_M)


def q_e2m3(v):
    """OCP fp6 e2m3 grid (round to nearest even, saturating at 7.5)."""
    a = np.abs(v).astype(np.float64)
    e = np.clip(np.floor(np.log2(np.maximum(a, 1e-30))), 0, 2)
    step = np.where(a < 1.0, 0.125, np.exp2(e - 3))
    return (np.sign(v) * np.minimum(np.round(a / step) * step, 7.5)).astype(np.float32)


def q_e2m1(v):
    """OCP fp4 e2m1 grid: 0 .5 1 1.5 2 3 4 6."""
    a = np.abs(v).astype(np.float64)
    e = np.clip(np.floor(np.log2(np.maximum(a, 1e-30))), 0, 2)
    step = np.where(a < 1.0, 0.5, np.exp2(e - 1))
    return (np.sign(v) * np.minimum(np.round(a / step) * step, 6.0)).astype(np.float32)


def mfma128(A, B, Cc):
    """v_mfma_scale_f32_16x16x128_f8f6f4 with fp6 / fp4 operands (tools/probe/mx6_layout.hip): A, B [64 lanes][32 fields], already scaled;
    D[m][n] = sum_g sum_i A[(m, g)][i] B[(n, g)][i]."""
    Am = np.zeros((16, 128), np.float64)
    Bm = np.zeros((128, 16), np.float64)
    k = 32 * G_[:, None] + np.arange(32)[None]
    Am[C_[:, None], k] = A
    Bm[k, C_[:, None]] = B
    D = (Am @ Bm).astype(np.float32)
    return Cc + D[4 * G_[:, None] + np.arange(4)[None], C_[:, None]]


def parse_header(img):
    vals = struct.unpack_from(HDR_FMT, img, 0)
    return dict(zip(HDR_FIELDS, vals))


def to_f32(u16, prec):
    u16 = np.asarray(u16, dtype=np.uint16)
    if prec == 0:
        return u16.view(np.float16).astype(np.float32)
    return (u16.astype(np.uint32) << 16).view(np.float32)


def round_op(x, prec):
    x = np.asarray(x, dtype=np.float32)
    if prec == 0:
        return x.astype(np.float16).astype(np.float32)
    u = x.view(np.uint32)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def mfma32(A, B, Cc):
    Am = np.zeros((16, 32), np.float32)
    Bm = np.zeros((32, 16), np.float32)
    k = 8 * G_[:, None] + np.arange(8)[None]
    Am[C_[:, None], k] = A
    Bm[k, C_[:, None]] = B
    D = (Am.astype(np.float64) @ Bm.astype(np.float64)).astype(np.float32)
    return Cc + D[4 * G_[:, None] + np.arange(4)[None], C_[:, None]]


def mfma16(A, B, Cc):
    Am = np.zeros((16, 16), np.float32)
    Bm = np.zeros((16, 16), np.float32)
    k = 4 * G_[:, None] + np.arange(4)[None]
    Am[C_[:, None], k] = A
    Bm[k, C_[:, None]] = B
    D = (Am.astype(np.float64) @ Bm.astype(np.float64)).astype(np.float32)
    return Cc + D[4 * G_[:, None] + np.arange(4)[None], C_[:, None]]


class EncoderEmu:
    def __init__(self, img):
        self.img = img
        self.h = parse_header(img)
        assert self.h["magic"] == 0x4B544955
        self.l6 = self.h["precision"] == 4                                # f16 operands + MX-fp6 low-order weight term
        self.prec = 0 if self.l6 or self.h["precision"] == 5 else self.h["precision"]      # 5 = fp8w: fp8-grid weights in fp16 fragments (the f16 kernel)
        self.depth = self.h["depth"]
        self.C = self.h["outputdim"]

    def _f32(self, off, n):
        return np.frombuffer(self.img, np.float32, n, off)

    def frag(self, stage, f):
        off = self.h["off_wstream"] + stage * STAGE + f * FRAG
        return to_f32(np.frombuffer(self.img, np.uint16, 512, off).reshape(64, 8), self.prec)

    def l6_frag(self, blk, stage_off, f):
        off = self.h["off_wstream"] + PE_STAGES * STAGE + blk * L6_BLOCK + stage_off + f * FRAG
        return to_f32(np.frombuffer(self.img, np.uint16, 512, off).reshape(64, 8), 0)

    def l6_frag6(self, blk, byte_off, frag_index):
        """fragment6 at byte_off inside block blk -> [64 lanes][32 fields] values, row scale (bias-table byte frag_index) applied"""
        off = self.h["off_wstream"] + PE_STAGES * STAGE + blk * L6_BLOCK + byte_off
        lo = np.frombuffer(self.img, np.uint32, 256, off).reshape(64, 4)
        hi = np.frombuffer(self.img, np.uint32, 128, off + 1024).reshape(64, 2)
        w = np.concatenate([lo, hi], 1).astype(np.uint64)
        bits = np.zeros((64, 32), np.uint32)
        for i in range(32):
            b = 6 * i
            v = w[:, b >> 5] >> np.uint64(b & 31)
            if (b & 31) > 26:
                v = v | (w[:, (b >> 5) + 1] << np.uint64(32 - (b & 31)))
            bits[:, i] = (v & np.uint64(63)).astype(np.uint32)
        e, m = (bits >> 3) & 3, bits & 7
        val = np.where(e == 0, m / 8.0, np.exp2(e.astype(np.float64) - 1) * (1 + m / 8.0)) * np.where(bits & 32, -1.0, 1.0)
        sc = np.frombuffer(self.img, np.uint8, 16 * 4, self.h["off_bias"] + (blk * BIAS_FLOATS + L6_SCALES + (frag_index >> 2) * 16) * 4).reshape(16, 4)
        return (val * np.exp2(sc[C_, frag_index & 3].astype(np.float64) - 127)[:, None]).astype(np.float64)

    @staticmethod
    def x6(X, ct):
        """fp6 copy of a token tile's four k-step operands (v_cvt_scalef32_pk32_fp6_f16, scale 2): field i = 8 s + j"""
        return 2.0 * q_e2m3(np.concatenate([X[s][ct] for s in range(4)], 1) / 2.0)

    def head_frag(self, ot, s, term=0):
        off = self.h["off_head_w"] + ((ot * 4 + s) * 2 + term) * FRAG
        return to_f32(np.frombuffer(self.img, np.uint16, 512, off).reshape(64, 8), self.prec)

    def pack8(self, lo, hi):
        return round_op(np.concatenate([lo, hi], axis=1), self.prec)

    def pack4(self, x):
        return round_op(x, self.prec)

    def layernorm_to_operand(self, R, eps):
        """LayerNorm as the kernel computes it: the stream is centred by pack.cpp, so x_hat = x * rsqrt(E[x^2] + eps)
        (no mean term; csrc/encoder.hip ln_stats)."""
        X = [[None] * 3 for _ in range(4)]
        for ct in range(3):
            ex2 = sum((R[ft][ct] ** 2).sum(1) for ft in range(8)).reshape(4, 16).sum(0)[C_] / 128.0
            rstd = 1.0 / np.sqrt(ex2 + eps)
            for s in range(4):
                X[s][ct] = self.pack8(R[2 * s][ct] * rstd[:, None], R[2 * s + 1][ct] * rstd[:, None])
        return X

    def run_wave(self, mel2, floor_db, nt=6):
        """mel2: two crops [2][frames>=16*nt][64] dB (no floor).  -> logits [2][C], stages [2][depth+1][24][128]"""
        h, prec = self.h, self.prec
        bn_sc, bn_sh = self._f32(h["off_bn_sc"], 64), self._f32(h["off_bn_sh"], 64)
        pe_bias = self._f32(h["off_pe_bias"], 24 * 128).reshape(24, 128)
        fin = self._f32(h["off_fin"], 4 * 128).reshape(4, 128)
        col = [16 * ct + C_ for ct in range(3)]
        clip = [(cc >= 24).astype(int) for cc in col]
        tk = [col[ct] - 24 * clip[ct] for ct in range(3)]
        pf = [t // 6 for t in tk]
        pt = [t % 6 for t in tk]
        j8 = np.arange(8)[None]
        # patch-embed operand
        XB = [[None] * 3 for _ in range(8)]
        for ct in range(3):
            for s in range(8):
                # k-slot (s,g,j) <-> mel-in-patch fi = 2s + (g>>1), frame-in-patch ti = 8(g&1) + j
                fbin = 16 * pf[ct] + 2 * s + (G_ >> 1)
                t = 16 * pt[ct][:, None] + 8 * (G_ & 1)[:, None] + j8
                t = np.minimum(t, mel2.shape[1] - 1)
                x = mel2[clip[ct][:, None], t, fbin[:, None]]
                x = np.maximum(x, floor_db) * bn_sc[fbin][:, None] + bn_sh[fbin][:, None]
                XB[s][ct] = round_op(x, prec) * (pt[ct] < nt)[:, None]
        r4 = np.arange(4)[None]
        R = [[pe_bias[tk[ct][:, None], 16 * ft + 4 * G_[:, None] + r4].copy() for ct in range(3)] for ft in range(8)]
        for i in range(PE_HI_STAGES):
            for f in range(16):
                A = self.frag(i, f)             # high-order term only: the low-order stages are used with option split_pe and by f16x2
                for ct in range(3):
                    R[2 * i + f // 8][ct] = mfma32(A, XB[f % 8][ct], R[2 * i + f // 8][ct])
        stages = np.zeros((2, self.depth + 1, 24, 128), np.float32)

        def dump(st):
            for ct in range(3):
                for ft in range(8):
                    stages[clip[ct][:, None], st, tk[ct][:, None], 16 * ft + 4 * G_[:, None] + r4] = R[ft][ct]
        dump(0)
        TJ, TI = [0, 1, 0, 1, 2, 1, 2], [0, 0, 1, 1, 1, 2, 2]
        tvalid, pvalid = [], []
        for i in range(7):
            j = 16 * TJ[i] + 4 * G_[:, None] + r4
            q = 16 * TI[i] + C_[:, None]
            same = (j >= 24) == (q >= 24)
            tvalid.append(same[:, 0])                                  # uniform over r (kernel: tvalid[i])
            pvalid.append(same & (((j % 24) % 6) < nt))                # kernel: pmask[i]
        for blk in range(self.depth):
            tb = PE_STAGES + blk * BLOCK_STAGES
            bias = self._f32(h["off_bias"] + blk * BIAS_FLOATS * 4, BIAS_FLOATS)
            X = self.layernorm_to_operand(R, 1e-6)
            l6 = self.l6
            fr = (lambda st, f: self.l6_frag(blk, 0 if st == 0 else L6_QK if st == 1 else l6_mlp_offset(st - 2), f)) if l6 else (lambda st, f: self.frag(tb + st, f))
            QK = [[np.repeat(bias[BIAS_QKV + 16 * ft + 4 * G_[:, None] + r4], 1, 0).copy() for _ in range(3)] for ft in range(4)]
            for f in range(16):
                A = fr(0, f)
                for ct in range(3):
                    QK[f // 4][ct] = mfma32(A, X[f % 4][ct], QK[f // 4][ct])
            Qh = [[self.pack4(QK[hh][ct]) for ct in range(3)] for hh in range(2)]
            Kh = [[self.pack4(QK[2 + hh][ct]) for ct in range(3)] for hh in range(2)]
            V = [[np.repeat(bias[BIAS_QKV + 64 + 16 * hh + C_][:, None], 4, 1) for hh in range(2)] for _ in range(3)]
            for f in range(8):
                Bw = fr(1, f)
                for ct in range(3):
                    V[ct][f // 4] = mfma32(X[f % 4][ct], Bw, V[ct][f // 4])
            Vh = [[self.pack4(V[ct][hh]) for hh in range(2)] for ct in range(3)]
            O = [[None] * 3 for _ in range(2)]
            for hh in range(2):
                S, tm = [], []
                for i in range(7):
                    s_ = mfma16(Kh[hh][TJ[i]], Qh[hh][TI[i]], np.zeros((64, 4), np.float32))    # log2-domain scores
                    S.append(s_)
                    tm.append(np.where(tvalid[i], s_.max(1), -np.inf))
                mx = [np.full(64, -np.inf, np.float32) for _ in range(3)]
                for i in range(7):
                    mx[TI[i]] = np.maximum(mx[TI[i]], tm[i])
                mx = [m_.reshape(4, 16).max(0)[C_] for m_ in mx]
                acc = [np.zeros((64, 4), np.float32) for _ in range(3)]
                Ls = [np.zeros((64, 4), np.float32) for _ in range(3)]
                ones = np.ones((64, 4), np.float32)
                for i in range(7):
                    with np.errstate(over="ignore"):
                        p = np.exp2(S[i] - mx[TI[i]][:, None]).astype(np.float32)
                    Pm = np.where(pvalid[i], self.pack4(p), 0.0).astype(np.float32)
                    acc[TI[i]] = mfma16(Vh[TJ[i]][hh], Pm, acc[TI[i]])
                    Ls[TI[i]] = mfma16(ones, Pm, Ls[TI[i]])
                for it in range(3):
                    O[hh][it] = acc[it] / Ls[it][:, :1]
            OX = [self.pack8(O[0][ct], O[1][ct]) for ct in range(3)]
            for ft in range(8):
                A = fr(1, 8 + ft)
                bp = bias[BIAS_PROJ + 16 * ft + 4 * G_[:, None] + r4]
                for ct in range(3):
                    R[ft][ct] = mfma32(A, OX[ct], R[ft][ct] + bp)
                    if l6:
                        R[ft][ct] = mfma32(fr(1, 16 + ft), OX[ct], R[ft][ct])       # proj: full fp16 low-order fragment
            X = self.layernorm_to_operand(R, 1e-6)
            HX4 = [[None] * 4 for _ in range(3)]
            for ch in range(12):
                Hh = [[bias[BIAS_FC1 + 32 * ch + 16 * j + 4 * G_[:, None] + r4].copy() for _ in range(3)] for j in range(2)]
                for f in range(8):
                    A = fr(2 + ch, f)
                    for ct in range(3):
                        Hh[f // 4][ct] = mfma32(A, X[f % 4][ct], Hh[f // 4][ct])
                if l6:
                    for i in range(2):
                        W6 = self.l6_frag6(blk, l6_mlp_offset(ch) + STAGE + i * FRAG6, 2 * ch + i)
                        for ct in range(3):
                            Hh[i][ct] = mfma128(W6, self.x6(X, ct), Hh[i][ct])
                HX = [self.pack8(np.maximum(Hh[0][ct], 0), np.maximum(Hh[1][ct], 0)) for ct in range(3)]
                for ft in range(8):
                    A = fr(2 + ch, 8 + ft)
                    for ct in range(3):
                        R[ft][ct] = mfma32(A, HX[ct], R[ft][ct])
                if l6:
                    for ct in range(3):
                        HX4[ct][ch & 3] = 2.0 * q_e2m1(HX[ct] / 2.0)                   # v_cvt_scalef32_pk_fp4_f16, scale 2
                    if (ch & 3) == 3:
                        for ft in range(8):
                            W6 = self.l6_frag6(blk, l6_mlp_offset(ch) + L6_M + ft * FRAG6, 24 + 8 * (ch // 4) + ft)
                            for ct in range(3):
                                R[ft][ct] = mfma128(W6, np.concatenate(HX4[ct], 1), R[ft][ct])
            for ft in range(8):
                b2 = bias[BIAS_FC2 + 16 * ft + 4 * G_[:, None] + r4]
                for ct in range(3):
                    R[ft][ct] = R[ft][ct] + b2
            dump(blk + 1)
        # final norm + masked mean pool (per clip) -- done directly on the dumped layout semantics
        xn = []
        for ct in range(3):
            ex2 = sum((R[ft][ct] ** 2).sum(1) for ft in range(8)).reshape(4, 16).sum(0)[C_] / 128.0
            rstd = np.where(pt[ct] < nt, 1.0 / np.sqrt(ex2 + 1e-6), 0.0)
            xn.append([R[ft][ct] * rstd[:, None] for ft in range(8)])
        lo = (C_ < 8)[:, None]
        pooled = []
        for ft in range(8):
            Aa = np.where(lo, xn[0][ft] + xn[1][ft], xn[0][ft])
            Bb = np.where(lo, xn[2][ft], xn[2][ft] + xn[1][ft])
            keep, send = np.where(lo, Aa, Bb), np.where(lo, Bb, Aa)
            v = keep + send[LANES ^ 8]
            for o in (4, 2, 1):
                v = v + v[LANES ^ o]
            pooled.append(v)
        inv_ntok = 1.0 / (4 * nt)
        z = [pooled[ft] * inv_ntok * fin[0][16 * ft + 4 * G_[:, None] + r4] + fin[1][16 * ft + 4 * G_[:, None] + r4] for ft in range(8)]
        tot = sum(zz.sum(1) for zz in z).reshape(4, 16).sum(0)[C_]
        mean = tot / 128.0
        var = sum(((zz - mean[:, None]) ** 2).sum(1) for zz in z).reshape(4, 16).sum(0)[C_] / 128.0
        rstd = 1.0 / np.sqrt(var + 1e-5)
        zbuf = np.zeros((16, 128), np.float32)
        zbuf_lo = np.zeros((16, 128), np.float32)
        for ft in range(8):
            zn = (z[ft] - mean[:, None]) * rstd[:, None] * fin[2][16 * ft + 4 * G_[:, None] + r4] + fin[3][16 * ft + 4 * G_[:, None] + r4]
            zh = self.pack4(zn)
            zl = self.pack4(zn - zh)
            for ln in LANES[(C_ & 7) == 0]:
                zbuf[int(C_[ln] >= 8), 16 * ft + 4 * G_[ln] + np.arange(4)] = zh[ln]
                zbuf_lo[int(C_[ln] >= 8), 16 * ft + 4 * G_[ln] + np.arange(4)] = zl[ln]
        # head on split operands: (Wh + Wl).(zh + zl) ~ Wh.zh + Wh.zl + Wl.zh; clip slot = column
        ntiles = (self.C + 15) // 16
        head_b = self._f32(h["off_head_b"], ntiles * 16)
        logits = np.zeros((2, self.C), np.float32)
        Bz, Bzl = [], []
        for s in range(4):
            for buf, dst in ((zbuf, Bz), (zbuf_lo, Bzl)):
                lo4 = buf[C_[:, None], 32 * s + 4 * G_[:, None] + r4]
                hi4 = buf[C_[:, None], 32 * s + 16 + 4 * G_[:, None] + r4]
                dst.append(np.concatenate([lo4, hi4], 1))
        for ot in range(ntiles):
            acc = head_b[16 * ot + 4 * G_[:, None] + r4].copy()
            for s in range(4):
                acc = mfma32(self.head_frag(ot, s, 1), Bz[s], acc)
                acc = mfma32(self.head_frag(ot, s, 0), Bzl[s], acc)
                acc = mfma32(self.head_frag(ot, s, 0), Bz[s], acc)
            for ln in LANES[C_ < 2]:
                for r in range(4):
                    cls = 16 * ot + 4 * G_[ln] + r
                    if cls < self.C:
                        logits[C_[ln], cls] = acc[ln, r]
        return logits, stages
