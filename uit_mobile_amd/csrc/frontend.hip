// frontend.hip -- fused log-mel front-end for gfx950 (MI355X).
//
// Replaces self.front_end(x) of the reference (models/uit.py:298-308, call :455): torchaudio 0.13
// MelSpectrogram(sr 16k, n_fft = win = 512, hop 160, center/reflect, power 2, 64 HTK mels 0..8 kHz) followed by
// AmplitudeToDB('power', top_db=120).  One kernel does reflect-pad indexing + window + 512-point real FFT +
// |X|^2 + mel filterbank + 10*log10(max(.,1e-10)) and the batch amax; the top_db clamp itself
// (max(x, amax-120)) needs the amax of the WHOLE call, so it is applied by the consumer (encoder prologue).
//
// Mapping (wave64; a workgroup = 4 waves walks "passes" = (clip, 16-frame column); each wave owns 4 of the 16 frames
// -- a frame quad that rotates from pass to pass -- from the sample loads to the stored operands: the waves of a
// workgroup never exchange data and there is no workgroup barrier in the pass loop; a clip's leftover frames are
// bundled into "tail" passes whose fully dead quads skip the FFT; passes are split statically over the persistent grid,
// the workgroups dispatched first taking the larger share (see the pass loop)):
//   FFT phase, per wave, 4 frames x 16 lanes:
//     * the 512 real samples of a frame are packed as 256 complex z[m] = x[2m] + i x[2m+1]; the 256-point complex
//       FFT is a 16x16 Cooley-Tukey: lane p holds z[16*m1 + p] (m1 = 0..15), does a 16-point FFT in registers,
//       multiplies by W256^(p*k1), transposes through a conflict-free padded LDS tile (row stride 17 complex),
//       does the second 16-point FFT, and the real-FFT split X[k] = ((Z[k]+conj Z[256-k]) - i w^k (Z[k]-conj
//       Z[256-k]))/2 pairs bins k and 256-k through LDS.  Complex values are float2 => v_pk_* math, b64 LDS ops.
//     * window and both twiddle sets are per-lane VGPR constants (kept in LDS instead they buy a third workgroup per CU:
//       measured +3 %, the kernel is VALU-issue bound -- DESIGN.md 4.1).
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
#include <type_traits>
#include "uitk_dev.h"

namespace uitk {

typedef float c32 __attribute__((ext_vector_type(2)));       // complex: (re, im), 8-byte aligned
typedef float f4v __attribute__((ext_vector_type(4)));

// Packed complex arithmetic.  A c32 lives in a 64-bit VGPR pair; v_pk_{mul,fma,add}_f32 take per-source half selects
// (op_sel / op_sel_hi) and per-half negation (neg_lo / neg_hi), so a complex multiply is TWO instructions and a
// multiplication by -i or a conjugation rides on the add that consumes it.  The compiler does not form these modifiers for
// f32 pairs (it emits 4 scalar mul/fma per complex multiply and v_mov + v_xor per swizzle: 406 VALU instructions in the FFT
// phase against 300 this way), hence inline asm (same arithmetic as the plain C++ forms, bit for bit: products and sums are
// formed in the same order, fma where the scalar form contracts to fma).
__device__ __forceinline__ c32 cmul(c32 a, c32 w) {          // a * w = (a.x w.x - a.y w.y, a.x w.y + a.y w.x)
    // one asm block: the compiler's hazard recogniser pads every inline-asm VGPR write that the next VALU instruction reads
    // with an s_nop (it cannot see that the writer has no dst_sel); a full 64-bit v_pk_*_f32 result needs none
    c32 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"                                               // (a.x w.x, a.x w.y)
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"                              // + (-a.y w.y, a.y w.x)
        : "=&v"(r) : "v"(a), "v"(w));
    return r;
}
__device__ __forceinline__ c32 add_mi(c32 s, c32 t) {        // s + (-i) t = (s.x + t.y, s.y - t.x)
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(s), "v"(t));
    return r;
}
__device__ __forceinline__ c32 sub_mi(c32 s, c32 t) {        // s - (-i) t = (s.x - t.y, s.y + t.x)
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(s), "v"(t));
    return r;
}
__device__ __forceinline__ c32 add_conj(c32 a, c32 b) {      // a + conj(b)
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ c32 sub_conj(c32 a, c32 b) {      // a - conj(b)
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// forward 4-point DFT (W4 = -i) in place; CMI: the third input is still to be multiplied by -i (folded into its adds)
template <bool CMI = false>
__device__ __forceinline__ void dft4(c32& a, c32& b, c32& c, c32& d) {
    const c32 s0 = CMI ? add_mi(a, c) : a + c, s1 = CMI ? sub_mi(a, c) : a - c, s2 = b + d, t = b - d;
    a = s0 + s2; c = s0 - s2;
    b = add_mi(s1, t); d = sub_mi(s1, t);
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
    //                                            W^4 = -i on v[8 + 2]: folded into the k1 = 2 butterfly below
    v[8 + 3] = cmul(v[8 + 3], c32{-H, -H});    // W^6
    v[12 + 1] = cmul(v[12 + 1], c32{S1, -C1}); // W^3
    v[12 + 2] = cmul(v[12 + 2], c32{-H, -H});  // W^6
    v[12 + 3] = cmul(v[12 + 3], c32{-C1, S1}); // W^9
    dft4(v[0], v[1], v[2], v[3]);
    dft4(v[4], v[5], v[6], v[7]);
    dft4<true>(v[8], v[9], v[10], v[11]);
    dft4(v[12], v[13], v[14], v[15]);
}

constexpr int FE_WAVES = 4;
#if defined(UITK_LAB) && defined(FE_OCC_LAB)
constexpr int FE_OCC = FE_OCC_LAB;                       // lab builds only (tools/mklib.sh -DUITK_LAB -DFE_OCC_LAB=3 -DFE_NOCONST): the occupancy probe of DESIGN.md 4.1
#else
constexpr int FE_OCC = 2;                                // workgroups per CU the register allocation aims at (a third buys nothing: DESIGN.md 4.1)
#endif
constexpr int FE_ROW = 17;                               // padded row stride (complex) of the transpose tile
constexpr int FE_ZROW = 272;                             // per-frame stride (complex) of the natural-order Z tile
constexpr int FE_WAVE_LDS = 4 * FE_ZROW * 8;             // 8704 B per wave: E[4][16][17] (8704 B) / Z[4][272]
constexpr int FE_PSTRIDE = 2 * FE_ZROW + 16;             // a frame's power row (MEL_PROW floats) overlays the start of its own (dead) Z
                                                         // row, skewed by 16 floats per frame: row stride 560 = 16 mod 32 (the four rows
                                                         // of a wave hit disjoint banks), rows 16-byte aligned (b128 operand reads)
static_assert(3 * 16 + MEL_PROW <= 2 * FE_ZROW && MEL_PROW % 4 == 0, "power row must fit inside its Z row");
constexpr int FE_B_LDS = MEL_BMAX * 64 * 4;              // filterbank B operands [13][64 lanes][4 steps]: 13 KiB
constexpr int FE_LDS = FE_WAVES * FE_WAVE_LDS + FE_B_LDS;                // 47 KiB

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
    // per-lane VGPR constants: window pairs, W256^(p*k1), (cos, sin)(2 pi k / 512)
    // LDS set-up of the workgroup, run AFTER the first pass's sample loads and the per-lane constant loads have been issued, so
    // that the kernel start pays one memory round trip, not three in a row (the fixed cost of a launch was 16 us)
    auto lds_setup = [&]() {
        // the power rows overlay the Z rows; their pad columns (bins 257..271, multiplied by zero filterbank operands) only have
        // to be finite: stale FFT values are, uninitialised LDS need not be
        for (int i = tid; i < FE_WAVES * FE_WAVE_LDS / 4; i += FE_WAVES * 64) ((float*)smem)[i] = 0.f;
        // filterbank operands: the whole banded table if it fits (HTK: 52 steps), zero-padded to MEL_BMAX steps
#pragma unroll
        for (int j = 0; j < (MEL_BMAX * 64 + FE_WAVES * 64 - 1) / (FE_WAVES * 64); ++j) {
            const int i = tid + j * FE_WAVES * 64;
            if (i < MEL_BMAX * 64) Blds[i] = (i >> 8) * 4 < a.mel_steps ? a.mel_b[64 + i] : 0.f;
        }
        __syncthreads();
    };
#if defined(UITK_LAB) && defined(FE_NOCONST)
    // Occupancy probe (timing only, WRONG results): ONE window pair and ONE twiddle serve every m1 / k1 -- the instruction stream is the
    // product's, instruction for instruction, but 60 VGPRs of per-lane constants are gone, so that a third wave per SIMD fits without
    // paying for it with LDS reads or extra arithmetic: what a third wave could buy at best (round 4).
    c32 win1 = *(const c32*)(a.window + 2 * p), tw1;
    { const float2 t = a.tw256[p * 16 + 1]; tw1 = c32{t.x, t.y}; }
    asm volatile("" : "+v"(win1), "+v"(tw1));
    c32 pw_[8];
#define win_(m1) win1
#define tw_(k1) tw1
#else
    c32 win_[16], tw_[16], pw_[8];                                       // per-lane VGPR constants
#pragma unroll
    for (int m1 = 0; m1 < 16; ++m1) win_[m1] = *(const c32*)(a.window + 32 * m1 + 2 * p);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) { const float2 t = a.tw256[p * 16 + k1]; tw_[k1] = c32{t.x, t.y}; }
#define win_(m1) win_[m1]
#define tw_(k1) tw_[k1]
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float2 t = a.tw512[p + 16 * j]; pw_[j] = c32{-t.y, -t.x}; }   // -i conj(w), w = (cos, sin)(2 pi k/512)

    // "full" passes are (clip b, column k): frames 16k..16k+15; the nF % 16 leftover frames of a clip are bundled,
    // per_tail clips at a time, into "tail" passes (for 1 s clips: 6 full passes per clip + one tail pass per 3 clips).
    // A workgroup walks pass = blockIdx.x + i * gridDim.x as the pair (cb, ck), pass = cb * max(full, 1) + ck: full passes
    // first (cb < B), then tail passes.  The two kinds run SPECIALISED loop bodies: a wave's instruction stream -- scalar
    // bookkeeping included -- is its critical path (two waves per SIMD: measured 1.37x from the second wave, nothing from a
    // third), so the full-pass body carries no tail / dead-slot logic, forms its addresses as uniform base + lane-invariant
    // 32-bit offset, and needs no 64-bit arithmetic per lane.
    const int full = a.nF >> 4, rem = a.nF & 15;
    const int per_tail = rem ? 16 / rem : 1;
    const int B = (int)a.B;
    const int ntail = rem ? (B + per_tail - 1) / per_tail : 0;
    // per-lane (clip offset, frame) of the 16 slots of a tail pass: slot = 4*wave+q in the FFT phase, 4q+r in the mel phase
    const int tslot = 4 * wave + q;
    const int t_cl = rem ? tslot / rem : 0, t_fr = rem ? tslot - t_cl * rem : 0;
    int m_cl[4], m_fr[4];                                 // mel phase: rows r of a wave's result = its own slots 4*wave + r
#pragma unroll
    for (int r = 0; r < 4; ++r) { m_cl[r] = rem ? (4 * wave + r) / rem : 0; m_fr[r] = rem ? (4 * wave + r) - m_cl[r] * rem : 0; }
    const int T = (int)a.T;
    const bool pair_ok = (T & 1) == 0;                   // pair loads need every row to start pair-aligned
    constexpr int ES = I16 ? 2 : 4;                      // bytes per sample
    float lmax = -INFINITY;
    // mel phase lane map (v_mfma_f32_4x4x1, 16 blocks): lane = 4*group + x; as A operand x = frame row of this wave's
    // quad, as B operand / result x = mel within the group, i.e. result column = mel `lane`
    const float bn_sc = PATCH ? a.bn_sc[lane] : 0.f, bn_sh = PATCH ? a.bn_sh[lane] : 0.f;   // init_bn scale/shift of mel `lane`
    // raw (unwindowed) sample pairs of a lane's frame: z[m1] = (x[32 m1 + 2p], x[32 m1 + 2p + 1]); `row` = the clip's first
    // sample (uniform for a full pass), su + sl = index of the lane's first sample (su uniform, sl >= 0 lane-invariant)
    auto load_frame = [&](const char* row, int su, int sl, c32 (&raw)[16]) {
        const int s0 = su + sl;
        if (pair_ok && s0 >= 0 && s0 + 32 * 15 + 1 < T) {                 // interior frame: no reflection, pair loads
            const char* src = row + (int64_t)su * ES + (unsigned)(sl * ES);    // scalar base + 32-bit lane offset
            if (I16) {
#pragma unroll
                for (int m1 = 0; m1 < 16; ++m1) {
                    const short2 s2 = *(const short2*)(src + 32 * ES * m1);
                    raw[m1] = c32{(float)s2.x, (float)s2.y} * (1.0f / 32768.0f);
                }
            } else {
#pragma unroll
                for (int m1 = 0; m1 < 16; ++m1) raw[m1] = *(const c32*)(src + 32 * ES * m1);
            }
        } else {
            // slow path (first / last frames of a clip: reflect padding).  The index is made opaque to the optimiser: left alone it
            // splits the reflection arithmetic into a uniform and a lane part and hoists ~250 scalar instructions (and 40 SGPR
            // spills) of it above this branch, into every pass of the drawn-pass loop.
            int s0v = s0;
            asm volatile("" : "+v"(s0v));
            const int64_t row_off = (row - (const char*)a.pcm) / ES;
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1)
                raw[m1] = c32{load_sample<I16>(a.pcm, row_off, T, s0v + 32 * m1),
                              load_sample<I16>(a.pcm, row_off, T, s0v + 32 * m1 + 1)};
        }
    };
    // full pass (clip b, column k): this wave transforms frame quad wq (frames 16k + 4wq .. +3).  The quad a wave takes rotates
    // from pass to pass: quad 0 of column 0 takes the reflect-padding slow path, and pinned to wave 0 it made that wave 3.5 %
    // slower than its three siblings (the workgroup ends with its slowest wave).
    const int sl_full = q * a.hop + 2 * p;                                // lane part of a full pass's first-sample index
    auto load_full = [&](int b, int k, int wq, c32 (&raw)[16]) {
        load_frame((const char*)a.pcm + (int64_t)b * a.T * ES, (16 * k + 4 * wq) * a.hop - a.pad, sl_full, raw);
    };
    auto load_tail = [&](int t, c32 (&raw)[16]) {
        if (4 * wave >= per_tail * rem) return;                           // fully dead quad of a tail pass (uniform)
        const int cb0 = t * per_tail;
        int b = cb0 + t_cl, f = 16 * full + t_fr;
        if (t_cl >= per_tail || b >= B) { b = cb0; f = 16 * full; }       // dead slot: any valid frame
        load_frame((const char*)a.pcm + (int64_t)b * a.T * ES, f * a.hop - a.pad, 2 * p, raw);
    };
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    if constexpr (TIMING) tlast = fe_stamp();

    const int nst = a.mel_steps;
    const int klo_g = ((const int*)a.mel_b)[lane];       // first bin (multiple of 4) of this lane's 4-mel group
    typedef float f4a __attribute__((ext_vector_type(4)));
    float* Pq = (float*)E + q * FE_PSTRIDE;              // this frame's power row (wave-private, over its own Z row)
    const float* Ar = (const float*)E + (lane & 3) * FE_PSTRIDE + klo_g;     // mel A[row = frame lane&3][k = step] of block lane>>2
    const float* Bl = Blds + lane * 4;
    // lane part of the patch-major operand address (single-crop clips): [token pf*6 + k][mel-in-patch][frame-in-patch]
    const unsigned st_lane = (unsigned)((((lane >> 4) * 6 * 16 + (lane & 15)) * 16) * 2);       // + 8 wq (uniform) for frame quad wq
    c32 v[16];

    // -------------------------------------------------------------------------------- FFT phase of one pass (VALU + LDS)
    auto fft_phase = [&](c32 (&v)[16]) {
#pragma unroll
        for (int m1 = 0; m1 < 16; ++m1) v[m1] = v[m1] * win_(m1);
        fft16(v);                                        // over m1 -> k1 (at FPOS), for m2 = p
#if defined(UITK_LAB) && defined(FE_XPOSE_DPP)
        // Lab variant (round 5, VERDICT r4 item 3: "replace the 16x16 transpose between the two radix-16 stages with in-register exchanges"):
        // the frame's 16 lanes are one DPP row, so the (register k1) x (lane m2) transpose is four butterfly stages on lane bits 3, 2, 1, 0:
        // stage b swaps w[r | b] of the lanes with bit b clear against w[r] of the lanes with bit b set (partner = lane ^ b).  Bits 3 and 2 take
        // ONE v_mov_b32_dpp per 32-bit value (row_ror:8 / row_shr:4 + row_shl:4, the receiving lanes chosen by the DPP bank mask, the rest keep
        // `old`); bits 1 and 0 live inside a bank: quad_perm fetch + v_cndmask.  No LDS instruction, no wave barrier: 16 KB of LDS traffic per
        // wave-pass become ~190 VALU instructions.
        {
            c32 w[16];
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) w[k1] = k1 == 0 ? v[FPOS(0)] : cmul(v[FPOS(k1)], tw_(k1));
            FSTAMP(1);
            auto dppf = [](float old, float src, auto ctrl, auto bank) {
                return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), decltype(ctrl)::value, 0xF, decltype(bank)::value, false));
            };
#define FE_XB(A, B, CA, BA, CB, BB) do { const c32 a0_ = A;                                                                                   \
                A.x = dppf(A.x, B.x, std::integral_constant<int, CA>{}, std::integral_constant<int, BA>{});                                    \
                A.y = dppf(A.y, B.y, std::integral_constant<int, CA>{}, std::integral_constant<int, BA>{});                                    \
                B.x = dppf(B.x, a0_.x, std::integral_constant<int, CB>{}, std::integral_constant<int, BB>{});                                  \
                B.y = dppf(B.y, a0_.y, std::integral_constant<int, CB>{}, std::integral_constant<int, BB>{}); } while (0)
#pragma unroll
            for (int r = 0; r < 8; ++r) FE_XB(w[r], w[r | 8], 0x128, 0xC, 0x128, 0x3);                     // bit 3: row_ror:8; banks 2,3 receive into w[r], banks 0,1 into w[r|8]
#pragma unroll
            for (int r = 0; r < 16; ++r) if (!(r & 4)) FE_XB(w[r], w[r | 4], 0x114, 0xA, 0x104, 0x5);      // bit 2: row_shr:4 into banks 1,3; row_shl:4 into banks 0,2
#undef FE_XB
            const bool b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
            auto qx = [](float src, auto ctrl) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), decltype(ctrl)::value, 0xF, 0xF, true)); };
#define FE_XQ(A, B, QP, BIT) do { const c32 ta_ = c32{qx(B.x, std::integral_constant<int, QP>{}), qx(B.y, std::integral_constant<int, QP>{})};   \
                const c32 tb_ = c32{qx(A.x, std::integral_constant<int, QP>{}), qx(A.y, std::integral_constant<int, QP>{})};                     \
                A = BIT ? ta_ : A; B = BIT ? B : tb_; } while (0)
#pragma unroll
            for (int r = 0; r < 16; ++r) if (!(r & 2)) FE_XQ(w[r], w[r | 2], 0x4E, b1);                     // bit 1: quad_perm [2,3,0,1]
#pragma unroll
            for (int r = 0; r < 16; ++r) if (!(r & 1)) FE_XQ(w[r], w[r | 1], 0xB1, b0);                     // bit 0: quad_perm [1,0,3,2]
#undef FE_XQ
#pragma unroll
            for (int m2 = 0; m2 < 16; ++m2) v[m2] = w[m2];
        }
#else
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {                // twiddle + transpose write E[q][k1][m2 = p]
            const c32 y = k1 == 0 ? v[FPOS(0)] : cmul(v[FPOS(k1)], tw_(k1));
            E[(q * 16 + k1) * FE_ROW + p] = y;
        }
        FSTAMP(1);                                       // window + FFT-1 + twiddle + transpose write
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int m2 = 0; m2 < 16; ++m2) v[m2] = E[(q * 16 + p) * FE_ROW + m2];   // lane k1 = p gathers its row
#endif
        fft16(v);                                        // over m2 -> Z[k1 + 16 k2], k1 = p, k2 at FPOS
        __builtin_amdgcn_wave_barrier();                 // every lane has read its E row: the tile may be reused
        asm volatile("" ::: "memory");
#ifndef FE_PAIR_VIA_Z
        // Real-FFT split partners (round 4).  Lane k1 = p holds Z[p + 16 k2] at v[FPOS(k2)].  Bin k = p + 16 j (j < 8) pairs with
        // 256 - k = (16 - p) + 16 (15 - j): the UPPER half (k2 >= 8) of lane 16 - p; for p = 0 with 16 (16 - j): lane 0's own k2 = 16 - j
        // (j = 0: Z[0] itself).  So a lane keeps its lower half in registers and only its upper half (+ Z[k1], row 8, for the p = 0 case)
        // goes through LDS: 9 writes + 8 reads per lane instead of the 16 + 17 of a natural-order Z tile.  Tile U[q][9][16] c32, frame
        // stride 9 rows = 1152 B (= 128 mod 256: the two frames of a 32-lane half hit disjoint banks).
        c32* U = E + q * (9 * 16);
#pragma unroll
        for (int k2 = 8; k2 < 16; ++k2) U[(k2 - 8) * 16 + p] = v[FPOS(k2)];
        U[8 * 16 + p] = v[FPOS(0)];
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        FSTAMP(2);                                       // transpose read + FFT-2 + upper-half write
        const c32* Ur = U + (p == 0 ? 16 : 0) + ((16 - p) & 15);
        c32 za[8], zb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { za[j] = v[FPOS(j)]; zb[j] = Ur[(7 - j) * 16]; }
        const c32 z128 = v[FPOS(8)];                     // lane p = 0: Z[128]
#else
        c32* Z = E;                                      // natural order Z[q][k], frame stride 272 (bank offset 32)
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) Z[q * FE_ZROW + 16 * k2 + p] = v[FPOS(k2)];
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        FSTAMP(2);                                       // transpose read + FFT-2 + Z write
        c32 za[8], zb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = p + 16 * j;
            za[j] = Z[q * FE_ZROW + k];
            zb[j] = Z[q * FE_ZROW + ((256 - k) & 255)];
        }
        const c32 z128 = Z[q * FE_ZROW + 128];
#endif
        __builtin_amdgcn_wave_barrier();                 // every partner value is in registers: the power row may overwrite the tile
        asm volatile("" ::: "memory");
        // power spectrum of bins k and 256-k into the wave's own tile P[frame q][bin]
        // 4 |X[k]|^2 and 4 |X[256-k]|^2 (the 1/4 lives in the filterbank operand, pack.cpp); k = 0 writes bin 256 (Nyquist).  The sixteen
        // values leave as eight ds_write2_b32 (two dwords 16 j apart from one base; round 4: -1.5 % against sixteen b32 writes -- the
        // compiler does not pair them).  LDS operations of a wave complete in order, so the plain reads of the mel phase behind the wave
        // barrier see them; the asm reads VALU results only (no inline-asm reader of an MFMA result: DESIGN_LOG.md 2.2).
        float PX[8], PY[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const c32 A = add_conj(za[j], zb[j]), Bm = sub_conj(za[j], zb[j]);   // Z[k] +- conj Z[256-k]
            const c32 Tw = cmul(Bm, pw_[j]);                                         // -i e^{-2 pi i k/512} (Z[k] - conj Z[256-k])
            const c32 X = A + Tw, Y = A - Tw;                                        // 2 X[k], 2 conj X[256-k]
            PX[j] = X.x * X.x + X.y * X.y;
            PY[j] = Y.x * Y.x + Y.y * Y.y;
        }
        {
            const unsigned bx = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(Pq + p);                 // bins p + 16 j
            const unsigned by = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(Pq + 256 - p - 112);      // bins 256 - p - 16 j
#define FE_W2(base, v0, v1, o0, o1) asm volatile("ds_write2_b32 %0, %1, %2 offset0:" #o0 " offset1:" #o1 :: "v"(base), "v"(v0), "v"(v1) : "memory")
            FE_W2(bx, PX[0], PX[1], 0, 16); FE_W2(bx, PX[2], PX[3], 32, 48); FE_W2(bx, PX[4], PX[5], 64, 80); FE_W2(bx, PX[6], PX[7], 96, 112);
            FE_W2(by, PY[0], PY[1], 112, 96); FE_W2(by, PY[2], PY[3], 80, 64); FE_W2(by, PY[4], PY[5], 48, 32); FE_W2(by, PY[6], PY[7], 16, 0);
#undef FE_W2
        }

        if (p == 0) Pq[128] = 4.0f * (z128.x * z128.x + z128.y * z128.y);
    };

    // ---- mel phase of one pass (matrix pipe): mel[frame][m] = sum_k P[frame][k] fb[k][m] for the wave's OWN four frames,
    // as 16 independent 4x4 blocks (one per 4-mel group, each walking its own bin band) of v_mfma_f32_4x4x1: no power
    // tile shared between waves, no workgroup barrier.  Operands arrive four steps at a time: A = ds_read_b128 of the
    // lane's power row at its group's band, B = ds_read_b128 of the lane's filterbank column.  One straight-line block of
    // MEL_BMAX steps (steps past the band multiply zero B operands by finite pad columns), so the reads of later steps fly
    // under the MFMAs of earlier ones; the next pass's sample loads are already in flight behind it.
    auto mel_mfma = [&]() -> f4v {
        f4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (nst <= MEL_BMAX) {
            // operand reads run MEL_AHEAD groups (of four steps) ahead of their MFMAs, so that the LDS latency is covered by
            // the MFMAs of the groups in between; the sched_barriers pin that order (the scheduler would otherwise sink
            // every read to just before its use and expose the latency once per group)
            constexpr int NG = MEL_BMAX / 4, MEL_AHEAD = 6;
            f4a A4[NG], B4[NG];
            // (round 4, measured: the filterbank operands from global memory instead of LDS +25 %; the first three groups' operands
            // kept in registers: neutral -- DESIGN_LOG.md)
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
        return acc0 + acc1;                              // D[frame slot 4*wave + r][mel lane]
    };
    // 10 log10(max(P, 1e-10)) = (10 log10 2) * v_log_f32: the operand is never denormal, so the library log10's
    // denormal scaling and split-constant multiply (14 instructions per value) buy nothing; |error| <= 2e-5 dB
    auto to_db = [&](float pw) { return 3.0102999566398120f * __builtin_amdgcn_logf(fmaxf(pw, 1e-10f)); };
    // 16-bit operands of the encoder's patch embed: init_bn applied here (y = sc*dB + sh, uit.py:310-313); the top_db
    // clamp commutes with this monotone map and with rounding, so the encoder applies it on y.
    // Layout [clip-crop][token pf*6+pt][mel-in-patch][frame-in-patch].
    auto store_elems = [&](const float (&db)[4], const int (&fb)[4], const int (&ff)[4]) {    // generic: any crop rule, dead slots
        const int pf = lane >> 4, mp = lane & 15;
        unsigned short* out = (unsigned short*)a.mel;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = ff[r];
            const int64_t bb = fb[r];
            if (bb < 0) continue;
            const unsigned short y = to_op16(db[r] * bn_sc + bn_sh, a.op_bf16);
            if (a.ncrops == 1) {
                if (f < 16 * a.nt) out[(((bb * 24 + pf * 6 + (f >> 4)) * 16 + mp) * 16) + (f & 15)] = y;
                continue;
            }
            // long clips: crops start every target_length frames, not patch aligned; a frame belongs to its regular crop
            // and/or the re-cropped tail (models/uit.py:474-480)
            const int cr0 = f / a.tl, t0 = f - cr0 * a.tl;
            if (cr0 < a.ncrops - 1 && t0 < 96)
                out[(((bb * a.ncrops + cr0) * 24 + pf * 6 + (t0 >> 4)) * 16 + mp) * 16 + (t0 & 15)] = y;
            const int t1 = f - (a.nF - a.tl);
            if (t1 >= 0 && t1 < 96)
                out[(((bb * a.ncrops + a.ncrops - 1) * 24 + pf * 6 + (t1 >> 4)) * 16 + mp) * 16 + (t1 & 15)] = y;
        }
    };
    // results of a full pass (clip b, column k): every slot live, frames 16k + 4*wave + r
    auto finish_full = [&](int b, int k, int wq, f4v acc) {
        float db[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) db[r] = to_db(acc[r]);
        // every frame counts for the top_db amax (two v_max3_f32; fmaxf would canonicalise each input with a v_max x, x first)
        asm("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4" : "+v"(lmax) : "v"(db[0]), "v"(db[1]), "v"(db[2]), "v"(db[3]));
        if (!PATCH) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a.mel[((int64_t)b * a.nF + 16 * k + 4 * wq + r) * 64 + lane] = db[r];
        } else if (a.ncrops == 1) {
            // lane = (mel, frames 4*wave .. 4*wave+3 of time patch k): 8 contiguous bytes at uniform base + lane offset
            if (k < a.nt) {
                typedef _Float16 h4v __attribute__((ext_vector_type(4)));
                typedef __bf16 b4v __attribute__((ext_vector_type(4)));
                char* dst = (char*)a.mel + ((int64_t)b * 24 + k) * 512 + 8 * wq + st_lane;
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = db[r] * bn_sc + bn_sh;
                if (a.op_bf16) {                         // uniform; two v_cvt_pk_* either way (RNE, = to_op16)
                    *(b4v*)dst = b4v{(__bf16)y[0], (__bf16)y[1], (__bf16)y[2], (__bf16)y[3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = __builtin_amdgcn_fmed3f(y[r], -65504.0f, 65504.0f);   // see to_op16
                    *(h4v*)dst = h4v{(_Float16)y[0], (_Float16)y[1], (_Float16)y[2], (_Float16)y[3]};
                }
            }
        } else {
            int fb[4], ff[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { fb[r] = b; ff[r] = 16 * k + 4 * wq + r; }
            store_elems(db, fb, ff);
        }
    };
    // results of tail pass t: slot 4*wave + r = (clip t*per_tail + m_cl[r], frame 16*full + m_fr[r]) or dead
    auto finish_tail = [&](int t, f4v acc) {
        float db[4];
        int fb[4], ff[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            db[r] = to_db(acc[r]);
            const int b = t * per_tail + m_cl[r];
            fb[r] = (m_cl[r] < per_tail && b < B) ? b : -1;
            ff[r] = 16 * full + m_fr[r];
            if (fb[r] >= 0) lmax = fmaxf(lmax, db[r]);
        }
        if (!PATCH) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (fb[r] >= 0) a.mel[((int64_t)fb[r] * a.nF + ff[r]) * 64 + lane] = db[r];
        } else {
            store_elems(db, fb, ff);
        }
    };

    // Passes: idx = column * B + clip for the full passes (column-major: every workgroup meets every column), then the tail
    // passes, split STATICALLY over the persistent grid.
    // Asymmetric split (a.asym_pm, chosen by the host when the grid is exactly two workgroups per CU): the workgroups dispatched
    // first (blockIdx < G/2, one per CU) hold the older waves of their SIMDs, the instruction arbiter serves the oldest wave first,
    // and with equal shares they finished 10 % ahead of their siblings while the kernel ended with the slowest workgroup (slowest /
    // median lifetime 1.10 -> 1.03).  They get a pool of asym_pm / 1000 of the passes to themselves; the share is measured per
    // device at uitk_reserve (api.cpp: calibrate_frontend_split).  Drawing the last passes from global counters instead was
    // measured and removed: a drawn pass costs ~1.9x a static one.
    const int G = (int)gridDim.x;
    const int nfull = B * full, npass = nfull + ntail;
    const bool asym = a.asym_pm != 500 && a.asym_pm > 0 && (G & 1) == 0 && G >= 2;
    const int stride = asym ? G / 2 : G;
    const int nA = asym ? (int)((int64_t)npass * a.asym_pm / 1000) : 0;
    const int gk = stride / B, gb = stride % B;                                       // the static stride as (columns, clips)
    int idx = (int)blockIdx.x, rnd = 0;
    int iend = npass;                                                                 // end of this workgroup's pool
    if (asym) { if ((int)blockIdx.x < stride) iend = nA; else idx = nA + (int)blockIdx.x - stride; }
    int ck = idx / B, cb = idx - ck * B;                                              // (column, clip) of a full pass
    auto issue_loads = [&](int i, int k_, int b_, int rot, c32 (&dst)[16]) {
        if (i < nfull) load_full(b_, k_, (wave + rot) & 3, dst);
        else if (i < iend) load_tail(i - nfull, dst);
    };
    if (idx < iend) issue_loads(idx, ck, cb, 0, v);
    lds_setup();
    // Tail frames that cannot matter (round 4).  For single-crop clips the leftover frames of a clip (1 s: frames 96..100) never reach
    // the encoder; they are transformed only because they count for the batch amax (top_db floor).  A frame's loudest mel bin is
    // bounded by Parseval: mel[m] = sum_k fb[k][m] P[k] <= max(fb) * sum_k |X[k]|^2 <= max(fb) * 512 * sum_n (w[n] x[n])^2 -- sixteen
    // packed fmas on the windowed samples the lane holds anyway.  If that bound (+0.05 dB for the roundings on both sides) does not
    // exceed what THIS wave has already seen (a lower bound of the batch amax), the quad's FFT, mel and log are skipped: the amax --
    // and with it every output bit -- is the same.  Wave-uniform decision; the wave's own maximum is reduced once, when its first
    // tail pass starts.
    const bool tail_skippable = PATCH && a.ncrops == 1 && a.fb_max > 0.f;
    float lmax_w = -INFINITY;
    bool lmax_w_valid = false;
    while (idx < iend) {
        FSTAMP(0);
        const bool is_full = idx < nfull;                                             // uniform
        const int wq = (wave + rnd) & 3;
        bool skipped = false;
        if (is_full) {
            fft_phase(v);
        } else if (4 * wave < per_tail * rem) {          // uniform per wave: fully dead quads of a tail pass skip the FFT
            bool need = true;
            if (tail_skippable) {
                if (!lmax_w_valid) {
                    lmax_w = lmax;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) lmax_w = fmaxf(lmax_w, __shfl_xor(lmax_w, o));
                    lmax_w_valid = true;
                }
                c32 e2 = {0.f, 0.f};
#pragma unroll
                for (int m1 = 0; m1 < 16; ++m1) { const c32 t = v[m1] * win_[m1]; e2 = __builtin_elementwise_fma(t, t, e2); }
                float e = e2.x + e2.y;                   // this lane's share of sum (w x)^2; then the sum over the frame's 16 lanes
                e += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
                e += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
                e += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e), 0x141, 0xF, 0xF, true));   // row_half_mirror
                e += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e), 0x140, 0xF, 0xF, true));   // row_mirror
                const float bound_db = to_db(512.0f * a.fb_max * e) + 0.05f;
                need = __builtin_amdgcn_ballot_w64(bound_db > lmax_w) != 0ull;
            }
            if (need) fft_phase(v); else skipped = true;
        }
        if (skipped) {                                   // nothing of this quad is stored or counted: straight to the next pass
            int nidx = idx + stride, nb = cb + gb, nk = ck + gk;
            if (nb >= B) { nb -= B; ++nk; }
            if (nidx < iend) issue_loads(nidx, nk, nb, rnd + 1, v);
            idx = nidx; cb = nb; ck = nk; ++rnd;
            continue;
        }
        FSTAMP(3);                                       // Z read + split + power write
        __builtin_amdgcn_wave_barrier();                 // the power rows are wave-private: LDS ops of one wave complete in
        asm volatile("" ::: "memory");                   // order, so the mel phase below needs no workgroup barrier
        // the sample registers are free again: the NEXT pass's loads go out now, their latency hides under this pass's mel
        // MFMAs, log and stores
        int nidx = idx + stride, nb = cb + gb, nk = ck + gk;
        if (nb >= B) { nb -= B; ++nk; }
        if (nidx < iend) issue_loads(nidx, nk, nb, rnd + 1, v);
        FSTAMP(4);                                       // next pass's load issue
        const f4v acc = mel_mfma();
        FSTAMP(5);                                       // mel MFMAs
        if (is_full) finish_full(cb, ck, wq, acc); else finish_tail(idx - nfull, acc);
        FSTAMP(7);                                       // log + bn + stores
        idx = nidx; cb = nb; ck = nk; ++rnd;
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
    FrontendArgs a2 = a;
    // the asymmetric split only applies to a full persistent grid of two workgroups per CU; -1 (not calibrated yet) = 590, the optimum
    // measured on the boxes of rounds 2 and 3
    const bool two_per_cu = npass >= resident && (wg_per_cu <= 0 ? FE_OCC : wg_per_cu) == 2;
    if (!two_per_cu) a2.asym_pm = 500; else if (a2.asym_pm < 0) a2.asym_pm = 590;
#ifdef UITK_LAB
    if (a.timing != nullptr && patch_layout && !in_i16) return launch_fe<false, true, true>(a2, grid, stream);
#else
    if (a.timing != nullptr) return hipErrorInvalidValue;
#endif
    if (patch_layout) return in_i16 ? launch_fe<true, true>(a2, grid, stream) : launch_fe<false, true>(a2, grid, stream);
    return in_i16 ? launch_fe<true, false>(a2, grid, stream) : launch_fe<false, false>(a2, grid, stream);
}

hipError_t launch_mel_export(const float* mel, const float* amax, float top_db, int64_t B, int nF, float* out,
                             hipStream_t stream) {
    hipLaunchKernelGGL(mel_export_kernel, dim3(512), dim3(256), 0, stream, mel, amax, top_db, B, nF, out);
    return hipGetLastError();
}

}  // namespace uitk
