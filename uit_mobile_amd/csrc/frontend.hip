// frontend.hip -- fused log-mel front-end for gfx950 (MI355X).
//
// Replaces self.front_end(x) of the reference (models/uit.py:298-308, call :455): torchaudio 0.13
// MelSpectrogram(sr 16k, n_fft = win = 512, hop 160, center/reflect, power 2, 64 HTK mels 0..8 kHz) followed by
// AmplitudeToDB('power', top_db=120).  One kernel does reflect-pad indexing + window + 512-point real FFT +
// |X|^2 + mel filterbank + 10*log10(max(.,1e-10)) and the batch amax; the top_db clamp itself
// (max(x, amax-120)) needs the amax of the WHOLE call, so it is applied by the consumer (encoder prologue).
//
// Mapping (wave64; a workgroup = 4 waves walks "passes" = (clip, 16-frame column); each wave owns 4 of the 16 frames
// from the sample loads to the stored operands -- the waves of a workgroup never exchange data and there is no
// workgroup barrier in the pass loop; a clip's leftover frames are bundled into "tail" passes whose fully dead
// quads skip the FFT):
//   FFT phase, per wave, 4 frames x 16 lanes:
//     * the 512 real samples of a frame are packed as 256 complex z[m] = x[2m] + i x[2m+1]; the 256-point complex
//       FFT is a 16x16 Cooley-Tukey: lane p holds z[16*m1 + p] (m1 = 0..15), does a 16-point FFT in registers,
//       multiplies by W256^(p*k1), transposes through a conflict-free padded LDS tile (row stride 17 complex),
//       does the second 16-point FFT, and the real-FFT split X[k] = ((Z[k]+conj Z[256-k]) - i w^k (Z[k]-conj
//       Z[256-k]))/2 pairs bins k and 256-k through LDS.  Complex values are float2 => v_pk_* math, b64 LDS ops.
//     * window and both twiddle sets are per-lane VGPR constants (FE_LDS_TABLES moves any of them to LDS; with
//       FE_OCC=3 that buys a third workgroup per CU -- measured equal, the kernel is VALU-issue bound).
//     * 4|X|^2 of a frame overwrites the start of its own (dead) Z row; the 1/4 lives in the filterbank operand.
//   then the NEXT pass's sample loads are issued (the FFT registers are free), and behind them the mel phase:
//     * mel[4 frames][64] = P[4][257] . fb[257][64] on the matrix pipe with the EXACT-fp32 v_mfma_f32_4x4x1_16b_f32
//       (bitwise an fp32 fma chain): its 16 independent 4x4 blocks are the sixteen 4-mel groups, each walking its
//       OWN bin band of the triangular HTK filterbank (52 steps; a 16-mel tile of the 16x16x4 form needs up to 134).
//       Operands come four steps per ds_read_b128 (A: the lane's power row at its group's band; B: the banded table,
//       13 KiB of LDS), six groups ahead of their MFMAs.  Result column = mel = lane.
//     * 10*log10(max(.,1e-10)) -> init_bn -> 16-bit operand, stored PATCH-MAJOR [clip-crop][24 tokens][16 mels][16 frames]
//       (8 bytes per lane: four frames of one mel) = exactly the k-slot order of the encoder's patch-embed MFMA
//       operand; the crop logic of models/uit.py:468-481 is resolved here (a frame lands in its regular crop
//       and/or the re-cropped tail); wave-reduced amax -> one atomic per wave at kernel end.
//   HBM traffic: the PCM is read once (frames overlap 3.2x, absorbed by L1/L2), output 256 B per frame.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "uitk_dev.h"

namespace uitk {

typedef float c32 __attribute__((ext_vector_type(2)));       // complex: (re, im), 8-byte aligned
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ c32 cmul(c32 a, c32 w) { return c32{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
__device__ __forceinline__ c32 mul_mi(c32 a) { return c32{a.y, -a.x}; }   // a * (-i)

// forward 4-point DFT (W4 = -i) in place
__device__ __forceinline__ void dft4(c32& a, c32& b, c32& c, c32& d) {
    const c32 s0 = a + c, s1 = a - c, s2 = b + d, s3 = mul_mi(b - d);
    a = s0 + s2; c = s0 - s2;
    b = s1 + s3; d = s1 - s3;
}

// forward 16-point DFT in place, natural order in.  OUTPUT ORDER: Y[k] is left in v[FPOS(k)] (no reorder copy).
// n = 4 n1 + n2, k = k1 + 4 k2:  Y = DFT4_{n2}( W16^{n2 k1} * DFT4_{n1}(v) ), result index 4*k1 + k2.
#define FPOS(k) (4 * ((k) & 3) + ((k) >> 2))
__device__ __forceinline__ void fft16(c32 (&v)[16]) {
    const float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);   // v[4*k1 + n2] = t[k1][n2]
    v[4 + 1] = cmul(v[4 + 1], c32{C1, -S1});   // W^1
    v[4 + 2] = cmul(v[4 + 2], c32{H, -H});     // W^2
    v[4 + 3] = cmul(v[4 + 3], c32{S1, -C1});   // W^3
    v[8 + 1] = cmul(v[8 + 1], c32{H, -H});     // W^2
    v[8 + 2] = mul_mi(v[8 + 2]);               // W^4 = -i
    v[8 + 3] = cmul(v[8 + 3], c32{-H, -H});    // W^6
    v[12 + 1] = cmul(v[12 + 1], c32{S1, -C1}); // W^3
    v[12 + 2] = cmul(v[12 + 2], c32{-H, -H});  // W^6
    v[12 + 3] = cmul(v[12 + 3], c32{-C1, S1}); // W^9
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4(v[4 * k1 + 0], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}

#ifndef FE_LDS_TABLES
#define FE_LDS_TABLES 0      // mask of constant tables kept in LDS instead of per-lane VGPRs (1 window, 2 W256, 4 W512); 0 measured fastest
#endif
constexpr int FE_WAVES = 4;
#ifndef FE_OCC
#define FE_OCC 2             // workgroups per CU the register allocation aims at
#endif
constexpr int FE_ROW = 17;                               // padded row stride (complex) of the transpose tile
constexpr int FE_ZROW = 272;                             // per-frame stride (complex) of the natural-order Z tile
constexpr int FE_WAVE_LDS = 4 * FE_ZROW * 8;             // 8704 B per wave: E[4][16][17] (8704 B) / Z[4][272]
constexpr int FE_PSTRIDE = 2 * FE_ZROW + 16;             // a frame's power row (MEL_PROW floats) overlays the start of its own (dead) Z
                                                         // row, skewed by 16 floats per frame: row stride 560 = 16 mod 32 (the four rows
                                                         // of a wave hit disjoint banks), rows 16-byte aligned (b128 operand reads)
static_assert(3 * 16 + MEL_PROW <= 2 * FE_ZROW && MEL_PROW % 4 == 0, "power row must fit inside its Z row");
constexpr int FE_B_LDS = MEL_BMAX * 64 * 4;              // filterbank B operands [13][64 lanes][4 steps]: 13 KiB
constexpr int FE_TAB_LDS = FE_LDS_TABLES ? (256 + 256 + 128) * 8 : 0;   // window pairs, W256^(p*k1) as [k1][p], (cos,sin)(2 pi k/512)
constexpr int FE_LDS = FE_WAVES * FE_WAVE_LDS + FE_B_LDS + FE_TAB_LDS;    // 47 KiB: three workgroups per CU

template <bool I16>
__device__ __forceinline__ float load_sample(const void* pcm, int64_t row_off, int T, int s) {
    if (s < 0) s = -s;                                   // reflect, edge sample not repeated (torch 'reflect')
    if (s >= T) s = 2 * (T - 1) - s;
    if (I16) return (float)((const int16_t*)pcm)[row_off + s] * (1.0f / 32768.0f);
    return ((const float*)pcm)[row_off + s];
}

// 16-bit operand conversions (round-to-nearest-even, identical to the encoder's v_cvt_pk_*)
__device__ __forceinline__ unsigned short to_op16(float x, int bf16) {
    if (bf16) { __bf16 h = (__bf16)x; return __builtin_bit_cast(unsigned short, h); }
    x = __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);    // degenerate init_bn statistics must not overflow fp16 to inf
    _Float16 h = (_Float16)x; return __builtin_bit_cast(unsigned short, h);
}

__device__ __forceinline__ unsigned long long fe_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define FSTAMP(slot) do { if constexpr (TIMING) { const unsigned long long _t = fe_stamp(); tacc[slot] += _t - tlast; tlast = _t; } } while (0)

template <bool I16, bool PATCH, bool TIMING>
__global__ __launch_bounds__(FE_WAVES * 64, FE_OCC) void mel_frontend_kernel(FrontendArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, q = lane >> 4, p = lane & 15;
    if (blockIdx.x == 0 && tid < AMAX_SLOTS) ((unsigned*)a.amax_rearm)[tid] = 0xFF800000u;   // arm the NEXT launch's slots with -inf
    c32* E = (c32*)(smem + wave * FE_WAVE_LDS);          // [4][16][17] transpose tile, then Z[4][272]
    float* Blds = (float*)(smem + FE_WAVES * FE_WAVE_LDS);
    // the power rows overlay the Z rows; their pad columns (bins 257..271, multiplied by zero filterbank operands) only have
    // to be finite: stale FFT values are, uninitialised LDS need not be
    for (int i = tid; i < FE_WAVES * FE_WAVE_LDS / 4; i += FE_WAVES * 64) ((float*)smem)[i] = 0.f;
    // filterbank operands: the whole banded table if it fits (HTK: 52 steps), zero-padded to MEL_BMAX steps
    for (int i = tid; i < MEL_BMAX * 64; i += FE_WAVES * 64) Blds[i] = (i >> 8) * 4 < a.mel_steps ? a.mel_b[64 + i] : 0.f;
    // per-lane constants: window pairs, W256^(p*k1), (cos, sin)(2 pi k / 512).  Each table lives either in VGPRs (default) or
    // in LDS (FE_LDS_TABLES mask: 1 window, 2 W256, 4 W512; b64 reads, conflict-free across p, broadcast across the wave's
    // four frames) -- the LDS forms free 32 / 32 / 16 VGPRs for a higher occupancy target (FE_OCC)
    c32* Wwin = (c32*)(smem + FE_WAVES * FE_WAVE_LDS + FE_B_LDS);       // window[(2m, 2m+1)], m = 16 m1 + p
    c32* Wtw = Wwin + 256;                                               // W256^(p*k1) at [k1][p]
    c32* Wpw = Wtw + 256;                                                // (cos, sin)(2 pi k / 512), k < 128
    if (FE_LDS_TABLES) {
        const int i = tid;                                               // 256 threads
        if (FE_LDS_TABLES & 1) Wwin[i] = *(const c32*)(a.window + 2 * i);
        if (FE_LDS_TABLES & 2) { const float2 t = a.tw256[(i & 15) * 16 + (i >> 4)]; Wtw[i] = c32{t.x, t.y}; }   // source is [p][k1]
        if ((FE_LDS_TABLES & 4) && i < 128) { const float2 u = a.tw512[i]; Wpw[i] = c32{u.x, u.y}; }
    }
    __syncthreads();
    c32 win_[16], tw_[16], pw_[8];                                       // per-lane VGPR constants
    if (!(FE_LDS_TABLES & 1)) {
#pragma unroll
        for (int m1 = 0; m1 < 16; ++m1) win_[m1] = *(const c32*)(a.window + 32 * m1 + 2 * p);
    }
    if (!(FE_LDS_TABLES & 2)) {
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) { const float2 t = a.tw256[p * 16 + k1]; tw_[k1] = c32{t.x, t.y}; }
    }
    if (!(FE_LDS_TABLES & 4)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float2 t = a.tw512[p + 16 * j]; pw_[j] = c32{t.x, t.y}; }
    }
#define WIN(m1) ((FE_LDS_TABLES & 1) ? Wwin[16 * (m1) + p] : win_[m1])
#define TW(k1) ((FE_LDS_TABLES & 2) ? Wtw[16 * (k1) + p] : tw_[k1])
#define PW(j) ((FE_LDS_TABLES & 4) ? Wpw[p + 16 * (j)] : pw_[j])

    // "full" passes are (clip b, column k): frames 16k..16k+15; the nF % 16 leftover frames of a clip are bundled,
    // per_tail clips at a time, into "tail" passes (for 1 s clips: 6 full passes per clip + one tail pass per 3 clips)
    const int full = a.nF >> 4, rem = a.nF & 15;
    const int per_tail = rem ? 16 / rem : 1;
    const int64_t nfull = a.B * (int64_t)full;
    const int64_t npass = nfull + (rem ? (a.B + per_tail - 1) / per_tail : 0);
    // per-lane (clip offset, frame) of the 16 slots of a tail pass: slot = 4*wave+q in the FFT phase, 4q+r in the mel phase
    const int tslot = 4 * wave + q;
    const int t_cl = rem ? tslot / rem : 0, t_fr = rem ? tslot - t_cl * rem : 0;
    int m_cl[4], m_fr[4];                                 // mel phase: rows r of a wave's result = its own slots 4*wave + r
#pragma unroll
    for (int r = 0; r < 4; ++r) { m_cl[r] = rem ? (4 * wave + r) / rem : 0; m_fr[r] = rem ? (4 * wave + r) - m_cl[r] * rem : 0; }
    const int T = (int)a.T;
    const bool pair_ok = (T & 1) == 0;                   // pair loads need every row to start pair-aligned
    float lmax = -INFINITY;
    int it = 0;
    // mel phase lane map (v_mfma_f32_4x4x1, 16 blocks): lane = 4*group + x; as A operand x = frame row of this wave's
    // quad, as B operand / result x = mel within the group, i.e. result column = mel `lane`
    const float bn_sc = PATCH ? a.bn_sc[lane] : 0.f, bn_sh = PATCH ? a.bn_sh[lane] : 0.f;   // init_bn scale/shift of mel `lane`
    // raw (unwindowed) sample pairs of this lane's frame for a pass: z[m1] = (x[32 m1 + 2p], x[32 m1 + 2p + 1])
    // (clip, column) of a full pass are tracked incrementally by the pass loop: a 64-bit division costs ~130 scalar
    // instructions per use
    auto load_pass = [&](int64_t pass, int pb_, int pk_, c32 (&raw)[16]) {
        int b, f;
        if (pass < nfull) {
            b = pb_;
            f = 16 * pk_ + 4 * wave + q;
        } else {
            const int64_t cb = (pass - nfull) * per_tail;
            if (4 * wave >= per_tail * rem) return;                       // fully dead quad of a tail pass (uniform)
            b = (int)cb + t_cl; f = 16 * full + t_fr;
            if (t_cl >= per_tail || b >= a.B) { b = (int)cb; f = 16 * full; }    // dead slot: any valid frame
        }
        const int64_t row_off = (int64_t)b * a.T;
        const int s0 = f * a.hop - 256 + 2 * p;
        if (pair_ok && s0 >= 0 && s0 + 32 * 15 + 1 < T) {                 // interior frame: no reflection, pair loads
            if (I16) {
                const int16_t* src = (const int16_t*)a.pcm + row_off + s0;
#pragma unroll
                for (int m1 = 0; m1 < 16; ++m1) {
                    const short2 s2 = *(const short2*)(src + 32 * m1);
                    raw[m1] = c32{(float)s2.x, (float)s2.y} * (1.0f / 32768.0f);
                }
            } else {
                const float* src = (const float*)a.pcm + row_off + s0;
#pragma unroll
                for (int m1 = 0; m1 < 16; ++m1) raw[m1] = *(const c32*)(src + 32 * m1);
            }
        } else {
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1)
                raw[m1] = c32{load_sample<I16>(a.pcm, row_off, T, s0 + 32 * m1),
                              load_sample<I16>(a.pcm, row_off, T, s0 + 32 * m1 + 1)};
        }
    };
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    if constexpr (TIMING) tlast = fe_stamp();

    const int nst = a.mel_steps;
    const int klo_g = ((const int*)a.mel_b)[lane];       // first bin (multiple of 4) of this lane's 4-mel group
    typedef float f4a __attribute__((ext_vector_type(4)));

    // ---- mel phase of one pass (matrix pipe): mel[frame][m] = sum_k P[frame][k] fb[k][m] for the wave's OWN four frames,
    // as 16 independent 4x4 blocks (one per 4-mel group, each walking its own bin band) of v_mfma_f32_4x4x1: no power
    // tile shared between waves, no workgroup barrier.  Operands arrive four steps at a time: A = ds_read_b128 of the
    // lane's power row at its group's band, B = ds_read_b128 of the lane's filterbank column.  One straight-line block of
    // MEL_BMAX steps (steps past the band multiply zero B operands by finite pad columns), so the reads of later steps fly
    // under the MFMAs of earlier ones; the next pass's sample loads are already in flight behind it.
    auto mel_phase = [&](int64_t passp, int pb_, int pk_) {
        const bool is_tail = passp >= nfull;             // uniform
        const int b = is_tail ? (int)((passp - nfull) * per_tail) : pb_;   // (first) clip of the pass
        const int k0 = is_tail ? full : pk_;
        const int fbase = 16 * k0 + 4 * wave;            // first frame of this wave's quad in a full pass
        const float* Ar = (const float*)E + (lane & 3) * FE_PSTRIDE + klo_g;     // A[row = frame lane&3][k = step] of block lane>>2
        const float* Bl = Blds + lane * 4;
        f4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (nst <= MEL_BMAX) {
            // operand reads run MEL_AHEAD groups (of four steps) ahead of their MFMAs, so that the LDS latency is covered by
            // the MFMAs of the groups in between; the sched_barriers pin that order (the scheduler would otherwise sink
            // every read to just before its use and expose the latency once per group)
            constexpr int NG = MEL_BMAX / 4, MEL_AHEAD = 6;
            f4a A4[NG], B4[NG];
#pragma unroll
            for (int c = 0; c < MEL_AHEAD; ++c) { A4[c] = *(const f4a*)(Ar + 4 * c); B4[c] = *(const f4a*)(Bl + 256 * c); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NG; ++c) {
                if (c + MEL_AHEAD < NG) { A4[c + MEL_AHEAD] = *(const f4a*)(Ar + 4 * (c + MEL_AHEAD)); B4[c + MEL_AHEAD] = *(const f4a*)(Bl + 256 * (c + MEL_AHEAD)); }
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(A4[c][0], B4[c][0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(A4[c][1], B4[c][1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(A4[c][2], B4[c][2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(A4[c][3], B4[c][3], acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {                                         // bands wider than MEL_BMAX bins (not HTK; a dense filterbank walks 260)
            const float* Bg = a.mel_b + 64 + lane * 4;
            for (int c = 0; c < nst / 4; ++c) {
                const f4a A4 = *(const f4a*)(Ar + 4 * c), B4 = *(const f4a*)(Bg + 256 * c);
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(A4[0], B4[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(A4[1], B4[1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(A4[2], B4[2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(A4[3], B4[3], acc1, 0, 0, 0);
            }
        }
        const f4v acc = acc0 + acc1;                     // D[frame slot 4*wave + r][mel lane]
        FSTAMP(5);                                       // mel MFMAs
        float db[4];
        int fb[4], ff[4];                                // clip / frame of this lane's 4 rows; fb < 0 = dead slot
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            db[r] = 10.0f * __log10f(fmaxf(acc[r], 1e-10f));
            if (!is_tail) { fb[r] = b; ff[r] = fbase + r; }
            else { fb[r] = (m_cl[r] < per_tail && b + m_cl[r] < a.B) ? b + m_cl[r] : -1; ff[r] = 16 * full + m_fr[r]; }
            if (fb[r] >= 0) lmax = fmaxf(lmax, db[r]);                                // every frame counts for the top_db amax
        }
        if (!PATCH) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (fb[r] >= 0) a.mel[((int64_t)fb[r] * a.nF + ff[r]) * 64 + lane] = db[r];
        } else {
            // 16-bit operands of the encoder's patch embed: init_bn applied here (y = sc*dB + sh, uit.py:310-313);
            // the top_db clamp commutes with this monotone map and with rounding, so the encoder applies it on y.
            // Layout [clip-crop][token pf*6+pt][mel-in-patch][frame-in-patch].
            const int pf = lane >> 4, mp = lane & 15;
            unsigned short y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = to_op16(db[r] * bn_sc + bn_sh, a.op_bf16);
            unsigned short* out = (unsigned short*)a.mel;
            if (a.ncrops == 1) {
                // lane = (mel, frames 4*wave .. 4*wave+3 of time patch k0): 8 contiguous bytes per lane
                if (!is_tail && k0 < a.nt) {
                    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
                    *(us4*)(out + (((int64_t)b * 24 + pf * 6 + k0) * 16 + mp) * 16 + 4 * wave) = us4{y[0], y[1], y[2], y[3]};
                }
            } else {
                // long clips: crops start every target_length frames, not patch aligned -> per-element stores; a frame
                // belongs to its regular crop and/or the re-cropped tail (models/uit.py:474-480)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = ff[r];
                    const int64_t bb = fb[r];
                    if (bb < 0) continue;
                    const int cr0 = f / a.tl, t0 = f - cr0 * a.tl;
                    if (cr0 < a.ncrops - 1 && t0 < 96)
                        out[(((bb * a.ncrops + cr0) * 24 + pf * 6 + (t0 >> 4)) * 16 + mp) * 16 + (t0 & 15)] = y[r];
                    const int t1 = f - (a.nF - a.tl);
                    if (t1 >= 0 && t1 < 96)
                        out[(((bb * a.ncrops + a.ncrops - 1) * 24 + pf * 6 + (t1 >> 4)) * 16 + mp) * 16 + (t1 & 15)] = y[r];
                }
            }
        }
        FSTAMP(7);                                       // log + bn + stores
    };

    float* Pq = (float*)E + q * FE_PSTRIDE;              // this frame's power row (wave-private, over its own Z row)
    c32 v[16];
    const unsigned fdiv = full > 0 ? (unsigned)full : 1u;
    const int gb = (int)(gridDim.x / fdiv), gk = (int)(gridDim.x % fdiv);             // the pass stride as (clips, columns)
    int cb = (int)(blockIdx.x / fdiv), ck = (int)(blockIdx.x % fdiv);                 // (clip, column) of the current pass
    if ((int64_t)blockIdx.x < npass) load_pass(blockIdx.x, cb, ck, v);
    for (int64_t pass = blockIdx.x; pass < npass; pass += gridDim.x, ++it) {
        const bool is_tail = pass >= nfull;              // uniform
        const bool quad_live = !is_tail || 4 * wave < per_tail * rem;        // uniform per wave: skip fully dead quads
        FSTAMP(0);                                       // pass bookkeeping
        // -------------------------------------------------------------------------------- FFT phase
        if (quad_live) {
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) v[m1] = v[m1] * WIN(m1);
            fft16(v);                                    // over m1 -> k1 (at FPOS), for m2 = p
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {            // twiddle + transpose write E[q][k1][m2 = p]
                const c32 y = k1 == 0 ? v[FPOS(0)] : cmul(v[FPOS(k1)], TW(k1));
                E[(q * 16 + k1) * FE_ROW + p] = y;
            }
            FSTAMP(1);                                   // window + FFT-1 + twiddle + transpose write
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int m2 = 0; m2 < 16; ++m2) v[m2] = E[(q * 16 + p) * FE_ROW + m2];   // lane k1 = p gathers its row
            fft16(v);                                    // over m2 -> Z[k1 + 16 k2], k1 = p, k2 at FPOS
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            c32* Z = E;                                  // natural order Z[q][k], frame stride 272 (bank offset 32)
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) Z[q * FE_ZROW + 16 * k2 + p] = v[FPOS(k2)];
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            FSTAMP(2);                                   // transpose read + FFT-2 + Z write
            c32 za[8], zb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = p + 16 * j;
                za[j] = Z[q * FE_ZROW + k];
                zb[j] = Z[q * FE_ZROW + ((256 - k) & 255)];
            }
            const c32 z128 = Z[q * FE_ZROW + 128];
            __builtin_amdgcn_wave_barrier();             // every Z value is in registers: the power row may overwrite the Z row
            asm volatile("" ::: "memory");
            // power spectrum of bins k and 256-k into the workgroup tile P[frame 4*wave+q][bin]
            float* prow = Pq;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = p + 16 * j;
                const float Ar = za[j].x + zb[j].x, Ai = za[j].y - zb[j].y;
                const float Br = za[j].x - zb[j].x, Bi = za[j].y + zb[j].y;
                const c32 cs = PW(j);
                const float c = cs.x, s = cs.y;
                const float Tr = c * Bi - s * Br, Ti = -s * Bi - c * Br;
                const float xr = Ar + Tr, xi = Ai + Ti, yr = Ar - Tr, yi = Ti - Ai;
                prow[k] = xr * xr + xi * xi;                     // 4 |X[k]|^2: the 1/4 lives in the filterbank operand (pack.cpp)
                prow[256 - k] = yr * yr + yi * yi;               // k = 0 writes bin 256 (Nyquist)
            }
            if (p == 0) prow[128] = 4.0f * (z128.x * z128.x + z128.y * z128.y);
        }
        FSTAMP(3);                                       // Z read + split + power write
        __builtin_amdgcn_wave_barrier();                 // the power rows are wave-private: LDS ops of one wave complete in
        asm volatile("" ::: "memory");                   // order, so the mel phase below needs no workgroup barrier
        // the sample registers are free again: issue the NEXT pass's loads now, their HBM latency hides under this pass's
        // mel MFMAs, log and stores
        int nb = cb + gb, nk = ck + gk;
        if (nk >= full) { nk -= full; ++nb; }
        if (pass + gridDim.x < npass) load_pass(pass + gridDim.x, nb, nk, v);
        FSTAMP(4);                                       // next pass's load issue
        mel_phase(pass, cb, ck);
        cb = nb; ck = nk;
    }
    FSTAMP(6);                                                            // (accumulates log + stores of all passes? no: tail only)
    if constexpr (TIMING) {
        if (a.timing != nullptr && lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a.timing[((size_t)blockIdx.x * FE_WAVES + wave) * 8 + i] = tacc[i];
        }
    }
    // batch amax: wave reduce, then one atomic per wave into one of UITK_AMAX_SLOTS slots.  Slots are armed with -inf
    // (0xFF800000), so they can be MAX-reduced as plain floats (no NaN).  Float max by integer atomics on the bits: a value with
    // the sign bit clear wins by signed max (every negative float, -inf included, is a negative int), a value with the sign
    // bit set (-0.0 included) by unsigned min (larger magnitude = larger unsigned; any non-negative value is smaller still).
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if (lane == 0 && lmax > -INFINITY) {
        float* slot = a.amax + (blockIdx.x & (AMAX_SLOTS - 1));
        if ((__float_as_uint(lmax) >> 31) == 0u) atomicMax((int*)slot, __float_as_int(lmax));
        else atomicMin((unsigned*)slot, __float_as_uint(lmax));
    }
}

// mel[B][nF][64] (+ floor) -> out[B][64][nF]  (the reference's 'b f t' layout; parity tests only)
__global__ void mel_export_kernel(const float* __restrict__ mel, const float* __restrict__ amax, float top_db,
                                  int64_t B, int nF, float* __restrict__ out) {
    float mx = -INFINITY;
    for (int i = 0; i < AMAX_SLOTS; ++i) mx = fmaxf(mx, amax[i]);
    const float flo = mx - top_db;
    int64_t n = B * (int64_t)nF * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t b = i / ((int64_t)nF * 64);
        int r = (int)(i - b * nF * 64);
        int m = r / nF, f = r - m * nF;
        out[i] = fmaxf(mel[(b * nF + f) * 64 + m], flo);
    }
}

template <bool I16, bool PATCH, bool TIMING = false>
static hipError_t launch_fe(const FrontendArgs& a, int grid, hipStream_t stream) {
    // the attribute is per device (one model handle per device may live in one process): one bit per device, set once
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute((const void*)mel_frontend_kernel<I16, PATCH, TIMING>, hipFuncAttributeMaxDynamicSharedMemorySize, FE_LDS);
        if (e != hipSuccess) return e;
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((mel_frontend_kernel<I16, PATCH, TIMING>), dim3(grid), dim3(FE_WAVES * 64), FE_LDS, stream, a);
    return hipGetLastError();
}

hipError_t launch_frontend(const FrontendArgs& a, int in_i16, int patch_layout, int n_cu, int wg_per_cu, hipStream_t stream) {
    const int rem = a.nF & 15;
    const int64_t npass = a.B * (int64_t)(a.nF >> 4) + (rem ? (a.B + 16 / rem - 1) / (16 / rem) : 0);
    const int64_t resident = (int64_t)n_cu * (wg_per_cu > 0 ? wg_per_cu : FE_OCC);       // persistent workgroups
    int grid = (int)(npass < resident ? npass : resident);
    if (grid < 1) grid = 1;
    if (a.timing != nullptr && patch_layout && !in_i16) return launch_fe<false, true, true>(a, grid, stream);
    if (patch_layout) return in_i16 ? launch_fe<true, true>(a, grid, stream) : launch_fe<false, true>(a, grid, stream);
    return in_i16 ? launch_fe<true, false>(a, grid, stream) : launch_fe<false, false>(a, grid, stream);
}

hipError_t launch_mel_export(const float* mel, const float* amax, float top_db, int64_t B, int nF, float* out,
                             hipStream_t stream) {
    hipLaunchKernelGGL(mel_export_kernel, dim3(512), dim3(256), 0, stream, mel, amax, top_db, B, nF, out);
    return hipGetLastError();
}

}  // namespace uitk
