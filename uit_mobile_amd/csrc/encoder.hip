// encoder.hip -- whole-encoder persistent-per-wave kernel for gfx950 (MI355X).
//
// Replaces everything after the front-end in UITBase.forward (models/uit.py:460-492): top_db clamp, init_bn,
// AudioPatchEmbed (:63-66) + pos-embeds (:380-388), depth x Block (:206-248: BNeckAttention :89-122, Mlp+ReLU
// :181-203), final norm (:395), mean pool + LayerNorm + Linear + sigmoid (:398-404, 358-360).
//
// Design (CDNA4, wave64, MFMA 16x16x32 / 16x16x16, fp32 accumulate):
//   * ONE WAVE OWNS TWO CLIP-CROPS (2 x 24 tokens = 48 token columns = three 16-wide MFMA column tiles, no
//     padding waste) from patch-embed to the pooled feature vector.  Nothing but the weights ever leaves the
//     register file: there is no inter-layer HBM or LDS traffic for activations and no cross-wave reduction.
//   * "feature-major" layout: every activation is held transposed, Y^T[feature][token], as MFMA C/D tiles
//     (lane = token column c = lane&15, registers = 4 consecutive features 4g+r, g = lane>>4).  A linear layer
//     is Y^T = W . X^T with the WEIGHTS as the A operand, so
//        - the fp32 residual stream is itself the accumulator of proj / fc2 (residual add is free),
//        - LayerNorm statistics are an in-lane sum over 32 registers + two cross-lane steps (xor 16, 32),
//        - a C/D tile pair converts in place (v_cvt_pk) to the next MFMA's B operand (k-order permutation is
//          absorbed into the weight packing), so LN->qkv, attention->proj and fc1->ReLU->fc2 chain with no
//          lane movement and no LDS.
//   * attention per (clip, head) on 16x16x16 MFMAs whose K = head_dim 16 exactly: S^T = K.Q^T (softmax axis =
//     registers + 2 cross-lane steps), V is produced token-major by swapping operand roles so O^T = V^T.P^T
//     needs no transpose either.  The 48 columns hold two clips: cross-clip score tiles are masked.
//   * weights are pre-packed (pack.cpp) as a linear stream of 1 KiB MFMA fragments in consumption order and
//     flow global -> LDS by LDS-DMA (global_load_lds_dwordx4) through a 4-slot x 16 KiB ring, two stages
//     ahead, one s_barrier per stage, counted vmcnt (never drained in the loop).  All 8 waves of the
//     workgroup read every fragment (ds_read_b128, lane-linear => conflict-free); each fragment feeds 3 MFMAs.
//   * the classifier head gathers the 16 pooled vectors of the workgroup through LDS so the 128->537 GEMM runs
//     once per workgroup (16 clips = one full column tile) instead of once per wave.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "uitk_dev.h"

namespace uitk {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));

template <int PREC> struct Op;
template <> struct Op<0> {   // fp16 operands
    using v8 = h8; using v4 = h4;
    static __device__ __forceinline__ v8 pack8(f4 lo, f4 hi) {
        return v8{(_Float16)lo[0], (_Float16)lo[1], (_Float16)lo[2], (_Float16)lo[3],
                  (_Float16)hi[0], (_Float16)hi[1], (_Float16)hi[2], (_Float16)hi[3]};
    }
    static __device__ __forceinline__ v4 pack4(f4 x) {
        return v4{(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3]};
    }
    static __device__ __forceinline__ f4 mfma32(v8 a, v8 b, f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f4 mfma16(v4 a, v4 b, f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
    }
    // ReLU after conversion: relu(cvt(x)) == cvt(relu(x)); v_pk_max_f16 does two elements per instruction and
    // avoids the canonicalising v_max the compiler puts in front of an f32 max on MFMA results.
    static __device__ __forceinline__ v8 relu_pack8(f4 lo, f4 hi) {
        const v8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        return __builtin_elementwise_max(pack8(lo, hi), z);
    }
    static __device__ __forceinline__ v4 ones4() { return v4{(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f}; }
    static __device__ __forceinline__ f4 unpack4(v4 x) { return f4{(float)x[0], (float)x[1], (float)x[2], (float)x[3]}; }
    static __device__ __forceinline__ v8 max8(v8 y, v8 t) { return __builtin_elementwise_max(y, t); }
    static __device__ __forceinline__ v8 min8(v8 y, v8 t) { return __builtin_elementwise_min(y, t); }
    static __device__ __forceinline__ v8 from_f32x8(f4 lo, f4 hi) { return pack8(lo, hi); }
    static __device__ __forceinline__ v8 zero8() { return v8{0, 0, 0, 0, 0, 0, 0, 0}; }
};
template <> struct Op<1> {   // bf16 operands
    using v8 = b8; using v4 = b4;
    static __device__ __forceinline__ v8 pack8(f4 lo, f4 hi) {
        return v8{(__bf16)lo[0], (__bf16)lo[1], (__bf16)lo[2], (__bf16)lo[3],
                  (__bf16)hi[0], (__bf16)hi[1], (__bf16)hi[2], (__bf16)hi[3]};
    }
    static __device__ __forceinline__ v4 pack4(f4 x) {
        return v4{(__bf16)x[0], (__bf16)x[1], (__bf16)x[2], (__bf16)x[3]};
    }
    static __device__ __forceinline__ f4 mfma32(v8 a, v8 b, f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f4 mfma16(v4 a, v4 b, f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4, a), __builtin_bit_cast(s4, b), c, 0, 0, 0);
    }
    // ReLU on the converted bits: a negative bf16 is a negative int16, so a packed signed max with 0 clears it
    // (v_pk_max_i16, two elements per instruction, no inline asm => the compiler handles the MFMA hazards).
    static __device__ __forceinline__ v8 relu_pack8(f4 lo, f4 hi) {
        typedef short s8 __attribute__((ext_vector_type(8)));
        const s8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        return __builtin_bit_cast(v8, __builtin_elementwise_max(__builtin_bit_cast(s8, pack8(lo, hi)), z));
    }
    static __device__ __forceinline__ v4 ones4() { return v4{(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f}; }
    static __device__ __forceinline__ f4 unpack4(v4 x) { return f4{(float)x[0], (float)x[1], (float)x[2], (float)x[3]}; }
    static __device__ __forceinline__ v8 max8(v8 y, v8 t) {
        v8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = (float)y[i] > (float)t[i] ? y[i] : t[i];
        return r;
    }
    static __device__ __forceinline__ v8 min8(v8 y, v8 t) {
        v8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = (float)y[i] < (float)t[i] ? y[i] : t[i];
        return r;
    }
    static __device__ __forceinline__ v8 from_f32x8(f4 lo, f4 hi) { return pack8(lo, hi); }
    static __device__ __forceinline__ v8 zero8() { return v8{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f}; }
};

constexpr int RING_SLOTS = 4;                 // power of two, >= PREFETCH + 2 (see ring_step); 8 slots with 3-4 stages in flight measured +0.5-1 %
constexpr int PREFETCH = 2;                   // stages in flight ahead of the one being consumed
static_assert((RING_SLOTS & (RING_SLOTS - 1)) == 0 && RING_SLOTS >= PREFETCH + 2 && RING_SLOTS >= PE_HI_STAGES && PREFETCH <= TAIL_STAGES, "ring geometry");
constexpr int LDS_RING = RING_SLOTS * STAGE_BYTES;            // 64 KiB
constexpr int LDS_BIAS = 2 * BIAS_FLOATS * 4;                 // 8 KiB (double-buffered per block)
constexpr int LDS_ZBUF = 2 * 16 * DIM * 2;                    // 8 KiB: up to 16 pooled vectors, 16-bit hi + lo terms
constexpr int LDS_PELO = PE_HI_STAGES * STAGE_BYTES;          // 64 KiB: low-order patch-embed weights (prologue only; 8-wave workgroups)
constexpr int ENC_LDS = LDS_RING + LDS_BIAS + LDS_ZBUF;       // 80 KiB (4-wave workgroups: two per CU)
constexpr int ENC_LDS_SPLIT_PE = ENC_LDS + LDS_PELO;          // 144 KiB (8-wave workgroups: one per CU anyway, by registers)
constexpr int PEB_STRIDE = DIM + 4;                           // padded row of the LDS copy of the pos/bias table: lane (token c, g) reads
                                                              // 16 B at bank 4 (c + g + 4 ft) mod 64 -> conflict-free per 16-lane slice
constexpr int LDS_PEB = (NTOK + 1 + 4) * PEB_STRIDE * 4;      // 15 312 B: 25 table rows + the 4 rows of final-norm / head-norm parameters
constexpr int ENC_LDS_W8 = ENC_LDS_SPLIT_PE + LDS_PEB;        // 8-wave workgroups: 162 768 B of the CU's 163 840

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---- LDS-DMA weight ring --------------------------------------------------------------------------------
// Stage s lives in slot s % RING_SLOTS.  At step t every wave (1) issues its share of stage t+PREFETCH,
// (2) waits (counted vmcnt) until its share of stage t has landed, (3) s_barrier => the whole stage is visible.
// WAR safety: the slot overwritten at step t held stage t+PREFETCH-RING_SLOTS <= t-2, and a wave can only be
// at step t after all waves passed barrier(t-1), i.e. finished reading stage t-2.
// The stream is padded with PREFETCH dummy stages (pack.cpp) so the issue and the vmcnt immediate never vary.
// (The scalar-base asm form of the DMA -- dma_b128 below, one lane-offset VGPR instead of a v_lshl_add_u64 + VGPR pair per piece --
// was measured here too in round 4: 36 VALU instructions less per block, but an s_mov m0 + s_nop per piece more, same time.)
__device__ __forceinline__ int lane_id_now();
__device__ __forceinline__ void dma_b128(const char* gsrc, unsigned lane16, char* ldst);
// SB (round 5): the scalar-base form of the DMA (dma_b128: uniform 64-bit base in scalar registers + ONE 32-bit lane offset, re-derived
// from v_mbcnt at the point of use) for the instantiations that have no vector register to spare -- GELU / full-width attention /
// pooling 'dm': the builtin's per-lane 64-bit addresses were the values those kernels spilled and RELOADED inside the block loop.
template <int WAVES, bool SB = false>
__device__ __forceinline__ void ring_issue(const char* wstream, char* ring, int stage, int wave, int lane) {
    constexpr int PER_WAVE = STAGE_FRAGS / WAVES;
    const char* src = wstream + (size_t)stage * STAGE_BYTES;
    char* dst = ring + (stage & (RING_SLOTS - 1)) * STAGE_BYTES;
    unsigned lane16 = 0;
    if constexpr (SB) lane16 = (unsigned)lane_id_now() * 16u;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int piece = wave * PER_WAVE + i;
        if constexpr (SB) dma_b128(src + piece * FRAG_BYTES, lane16, dst + piece * FRAG_BYTES);
        else __builtin_amdgcn_global_load_lds(GPTR(src + piece * FRAG_BYTES + lane * 16), LPTR(dst + piece * FRAG_BYTES), 16, 0, 0);
    }
}

template <int WAVES, bool SB = false>
__device__ __forceinline__ void ring_step(const char* wstream, char* ring, int t, int wave, int lane) {
    constexpr int PER_WAVE = STAGE_FRAGS / WAVES;
    ring_issue<WAVES, SB>(wstream, ring, t + PREFETCH, wave, lane);
    // all but the PREFETCH*PER_WAVE youngest LDS-DMAs (stages t+1 .. t+PREFETCH) have landed => this wave's share of stage t
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PREFETCH * PER_WAVE) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ int lane_id_now() {            // lane id from the exec-mask count: two VALU instructions, never a live register
    // (round 6: the two v_mbcnt as BUILTINS on an opaque scalar zero.  The volatile asm statement pins the point of evaluation -- the value
    //  is not hoisted out of the block loop -- while the v_mbcnt pair stays compiler-visible: as an asm statement with a free "=v" output
    //  it landed 0-5 wait states behind MFMAs writing / reading the register the allocator picked, which the hazard recogniser cannot
    //  see: tools/kernel_isa.py --hazards)
    unsigned z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
}
// scalar-base form of a 16-byte-per-lane LDS-DMA, written out: uniform 64-bit global base (SGPR pair) + ONE 32-bit lane offset, uniform
// LDS address in M0.  The builtin selects the 64-bit per-lane address form, i.e. a register pair per distinct piece that stays live
// (or is spilled and reloaded -- a vmem op the counted vmcnt scheme cannot see) across the whole block loop.
__device__ __forceinline__ void dma_b128(const char* gsrc, unsigned lane16, char* ldst) {
    const unsigned l = (unsigned)(uintptr_t)LPTR(ldst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(lane16), "s"(gsrc) : "memory", "m0");
}
__device__ __forceinline__ void dma_b32(const char* gsrc, unsigned lane4, char* ldst) {
    const unsigned l = (unsigned)(uintptr_t)LPTR(ldst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(l), "v"(lane4), "s"(gsrc) : "memory", "m0");
}

template <int WAVES, bool SCALAR_BASE = false>
__device__ __forceinline__ void bias_issue(const float* bias, char* breg, int blk, int wave, int lane) {
    constexpr int PER_WAVE_BYTES = BIAS_FLOATS * 4 / WAVES;
    constexpr int N = PER_WAVE_BYTES / 256;                      // dword x 64 lanes = 256 B per instruction
    const char* src = (const char*)(bias + (size_t)blk * BIAS_FLOATS) + wave * PER_WAVE_BYTES;
    char* dst = breg + (blk & 1) * BIAS_FLOATS * 4 + wave * PER_WAVE_BYTES;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if constexpr (SCALAR_BASE) dma_b32(src + i * 256, (unsigned)lane_id_now() * 4u, dst + i * 256);
        else __builtin_amdgcn_global_load_lds(GPTR(src + i * 256 + lane * 4), LPTR(dst + i * 256), 4, 0, 0);
    }
}

template <int PREC>
__device__ __forceinline__ typename Op<PREC>::v8 lds_frag(const char* slot, int frag, int lane) {
    return *(const typename Op<PREC>::v8*)(slot + frag * FRAG_BYTES + lane * 16);
}

__device__ __forceinline__ f4 splat(float x) { return f4{x, x, x, x}; }
// a + b as four scalar v_add_f32: beside MFMAs a v_pk_add_f32 pair costs more issue time than the scalar adds it replaces
__device__ __forceinline__ f4 add4s(f4 a, f4 b) {
    return f4{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]};
}
__device__ __forceinline__ float hsum(f4 x) { return (x[0] + x[1]) + (x[2] + x[3]); }
// Maxima without llvm.maxnum: for fmaxf on MFMA results or lane-swapped bits the compiler first canonicalises every input with a
// v_max_f32 x, x, x (sNaN quieting) -- 56 extra VALU instructions per block in the softmax alone.  v_med3_f32 against FLT_MAX is the
// maximum of the other two (no canonicalisation; +inf instead would be folded to maxnum + canonicalise); the scores are never NaN or +inf.
// Round 6: NO asm instruction here may write a register of its own choosing.  These helpers used to be asm statements with "=v" outputs
// (v_max_f32 / v_max3_f32), and the register allocator is free to place such an output on a register that an MFMA issued just before
// still has to WRITE -- rows 1..3 of the ones-row MFMA's destination are dead on arrival -- which the hazard recogniser cannot pad
// because it does not see into asm: the MFMA's late write-back replaced a query tile's softmax maximum by a row sum (the f16ax
// BNeckAttention race of round 5: `v_mfma v[102:105]` ; `v_max3_f32 v104` -- profiles/round6/f16ax_race.log; tools/kernel_isa.py
// --hazards finds every such site, tests/test_build_checks.py holds the library at zero).  vmax / vmax3 are builtins now (compiler-visible:
// padded); hmax keeps its single v_max3_f32 but IN PLACE on the register its visible v_med3_f32 has just written.
__device__ __forceinline__ float vmax(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, 3.4028234664e38f); }
__device__ __forceinline__ float vmax3(float a, float b, float c) { return vmax(vmax(a, b), c); }
// hmax takes MFMA results: its FIRST instruction must be one the compiler sees, so that the hazard recogniser pads the matrix-pipe ->
// VALU read (an asm-only hmax read the accumulators zero to two instructions behind the MFMA that writes them -- stale values,
// run-to-run different maxima: round 3, tools/l6_determinism.py) AND the write: the asm instruction overwrites m01's register.
__device__ __forceinline__ float hmax(f4 x) {
    float m01 = __builtin_amdgcn_fmed3f(x[0], x[1], 3.4028234664e38f);
    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(m01) : "v"(x[2]), "v"(x[3]));
    return m01;
}

// Reductions over the 4 lane groups g (lanes c, c+16, c+32, c+48) with the gfx950 VALU swaps (no LDS crossbar):
// v_permlane16_swap(v, v): {rows 0,0,2,2 | rows 1,1,3,3}; v_permlane32_swap(v, v): {lo,lo | hi,hi}.
__device__ __forceinline__ float xsum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float xmax(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = vmax(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// x + (x moved by a DPP row operation within each 16-lane row): quad_perm / row_half_mirror / row_mirror (pure VALU)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// LayerNorm statistics (fp32) of one token column held as R[ft][ct], ft = 0..7.
// pack.cpp removes the channel-mean of everything that is ever added to the residual stream (LayerNorm cannot see a common
// offset, so this is exact algebra): the kernel's stream has zero mean per token up to the rounding of the 16-bit weights
// (measured |mean| < 1e-5 of the token's deviation, 25x below the operand rounding).  So LayerNorm(x) = x * rsqrt(E[x^2] + eps):
// no mean, no subtraction, no cancellation -- 40 VALU instructions less per column tile than the one-pass form with a mean.
// -DUITK_LN_WITH_MEAN restores var = E[x^2] - mean^2 with the mean subtracted, -DUITK_LN_TWO_PASS the textbook two-pass form.
__device__ __forceinline__ void ln_stats(const f4 (&R)[8][3], int ct, float eps, float& mean, float& rstd) {
#if defined(UITK_LN_TWO_PASS)
    f4 acc = R[0][ct];
#pragma unroll
    for (int ft = 1; ft < 8; ++ft) acc = acc + R[ft][ct];
    mean = xsum(hsum(acc)) * (1.0f / DIM);
    f4 q = splat(0.f);
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) { const f4 d = R[ft][ct] - splat(mean); q = q + d * d; }
    rstd = __builtin_amdgcn_rsqf(xsum(hsum(q)) * (1.0f / DIM) + eps);
#elif defined(UITK_LN_WITH_MEAN)
    f4 acc = R[0][ct], q = R[0][ct] * R[0][ct];
#pragma unroll
    for (int ft = 1; ft < 8; ++ft) { acc = acc + R[ft][ct]; q = __builtin_elementwise_fma(R[ft][ct], R[ft][ct], q); }
    mean = xsum(hsum(acc)) * (1.0f / DIM);
    const float ex2 = xsum(hsum(q)) * (1.0f / DIM);
    rstd = __builtin_amdgcn_rsqf(fmaxf(__builtin_fmaf(-mean, mean, ex2), 0.f) + eps);
#else
    f4 q = R[0][ct] * R[0][ct];
#pragma unroll
    for (int ft = 1; ft < 8; ++ft) q = __builtin_elementwise_fma(R[ft][ct], R[ft][ct], q);
    mean = 0.f;
    rstd = __builtin_amdgcn_rsqf(xsum(hsum(q)) * (1.0f / DIM) + eps);
#endif
}

// LN + conversion to the MFMA operand of the next linear (its affine is folded into that linear by pack.cpp).
// CEN (round 5): operand centring -- the calibration mean mu of this LayerNorm's output is subtracted BEFORE the rounding (negmu = the
// block's -mu[128] in LDS; pack.cpp adds W mu to the linear's bias: exact for any mu).  A massive-activation channel is nearly the same
// in every token of every clip: rounded as it is, its fp16 error (half a unit of 2^-7 at |x| ~ 10) is the same in every token and
// survives the pooling; rounded as x - mu it is 10-100x smaller.  One packed fma instead of one packed mul per register pair.
template <int PREC, int NCT = 3, bool CEN = false>                     // NCT: live column tiles (2 for pooling 'token': one clip per wave)
__device__ __forceinline__ void layernorm_to_operand(const f4 (&R)[8][3], float eps, typename Op<PREC>::v8 (&X)[4][3], const float* negmu = nullptr) {
    if constexpr (CEN) {
        float rstd[3];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) { float mean; ln_stats(R, ct, eps, mean, rstd[ct]); }
        const float* nm = negmu + 4 * (lane_id_now() >> 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f4 m0 = *(const f4*)(nm + 32 * s), m1 = *(const f4*)(nm + 32 * s + 16);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                X[s][ct] = Op<PREC>::pack8(__builtin_elementwise_fma(R[2 * s][ct], splat(rstd[ct]), m0), __builtin_elementwise_fma(R[2 * s + 1][ct], splat(rstd[ct]), m1));
        }
        return;
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        float mean, rstd;
        ln_stats(R, ct, eps, mean, rstd);
        const f4 rs = splat(rstd);
#if defined(UITK_LN_TWO_PASS) || defined(UITK_LN_WITH_MEAN)
        const f4 nm = splat(-mean * rstd);
#pragma unroll
        for (int s = 0; s < 4; ++s) X[s][ct] = Op<PREC>::pack8(R[2 * s][ct] * rs + nm, R[2 * s + 1][ct] * rs + nm);
#else
#pragma unroll
        for (int s = 0; s < 4; ++s) X[s][ct] = Op<PREC>::pack8(R[2 * s][ct] * rs, R[2 * s + 1][ct] * rs);
#endif
    }
}

// Precision f16a: the LOW-ORDER term of column tile 0's LayerNorm operand, lo = (the fp32 value the operand was rounded from) - float(hi):
// what the fp16 rounding of X[.][0] dropped, as a second fp16 operand (hi + lo carries ~22 bits).  Same arithmetic as
// layernorm_to_operand (the centred form with CEN), recomputed for tile 0 only.
template <bool CEN>
__device__ __forceinline__ void layernorm_lo_operand(const f4 (&R)[8][3], float eps, const h8 (&X)[4][3], h8 (&Xlo)[4], const float* negmu) {
    float mean, rstd;
    ln_stats(R, 0, eps, mean, rstd);
    const float* nm = CEN ? negmu + 4 * (lane_id_now() >> 4) : nullptr;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        f4 m0 = splat(0.f), m1 = splat(0.f);
        if constexpr (CEN) { m0 = *(const f4*)(nm + 32 * s); m1 = *(const f4*)(nm + 32 * s + 16); }
        const h8 hi = X[s][0];
        const f4 v0 = __builtin_elementwise_fma(R[2 * s][0], splat(rstd), m0), v1 = __builtin_elementwise_fma(R[2 * s + 1][0], splat(rstd), m1);
        Xlo[s] = Op<0>::pack8(v0 - f4{(float)hi[0], (float)hi[1], (float)hi[2], (float)hi[3]}, v1 - f4{(float)hi[4], (float)hi[5], (float)hi[6], (float)hi[7]});
    }
}
// lo of a packed fp32 quad / pair of quads: x - float(pack(x))
__device__ __forceinline__ h4 pack4_lo(f4 x, h4 hi) { return Op<0>::pack4(x - f4{(float)hi[0], (float)hi[1], (float)hi[2], (float)hi[3]}); }
__device__ __forceinline__ h8 pack8_lo(f4 x0, f4 x1, h8 hi) {
    return Op<0>::pack8(x0 - f4{(float)hi[0], (float)hi[1], (float)hi[2], (float)hi[3]}, x1 - f4{(float)hi[4], (float)hi[5], (float)hi[6], (float)hi[7]});
}

typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef int v8i __attribute__((ext_vector_type(8)));

// Score tiles (jt,it) of the 48 token columns of a wave (two clips): S^T[j = 16jt+4g+r][i = 16it+c]; tile (0,2)/(2,0) never
// mixes tokens of one clip and is skipped.
__device__ constexpr int TJ[7] = {0, 1, 0, 1, 2, 1, 2};
__device__ constexpr int TI[7] = {0, 0, 1, 1, 1, 2, 2};

// Attention of one block on 16x16x16 MFMAs (K = head_dim 16 exactly) + the proj linear into the residual stream R.
//   Qh/Kh[h][ct]: q, k of head h as C/D tiles packed to 16 bits (feature-major);  Vh[ct][h]: V token-major (A operand of P.V)
//   tvalid / kinit: see the kernel;  proj: the eight 1 KiB proj fragments (K = 32 = (head, d));  bias: this block's bias table
template <class OP, bool SHORT, bool X2 = false, int NCT = 3, bool ASP = false>
__device__ __forceinline__ void attention_proj(f4 (&R)[8][3], const typename OP::v4 (&Qh)[2][3], const typename OP::v4 (&Kh)[2][3],
                                               const typename OP::v4 (&Vh)[3][2], const bool (&tvalid)[7], const f4 (&kinit)[3],
                                               const char* proj, const float* bias, int lane, int g, const typename OP::v4 (*Qlo)[3] = nullptr,
                                               const typename OP::v4 (*Klo)[3] = nullptr, const typename OP::v4 (*Vlo)[2] = nullptr) {
    // ASP (precisions f16a / f16ax): Qlo / Klo[h][ct], Vlo[ct][h] = the low-order fp16 terms of q, k, v (every tile); the attention output of
    // tile 0 goes into proj as hi + lo
    using v8 = typename OP::v8;
    using v4 = typename OP::v4;
    constexpr int N7 = NCT == 2 ? 4 : 7;                  // one clip in tiles 0 / 1: the four score tiles among them (the first four of TJ / TI)
    f4 O[2][3];
    // both heads advance together phase by phase (scores, maxima, exponentials, P.V): two independent dependency chains
    // in every phase for the scheduler to overlap
    f4 S[2][7];
    float tm[2][7], mx[2][3];
#pragma unroll
    for (int i = 0; i < N7; ++i) {
        // SHORT (padding / dead key columns exist; round 5): the score accumulators START from 0 for a valid key of the query's clip and
        // from -1e30 for every other one (kinit: per key tile, rebuilt per block from a 48-bit column mask in scalar registers -- see the
        // kernel), so the masked scores never win a maximum and their exponentials are exactly 0: no per-element mask registers alive
        // through the block, no select in the softmax (the valid entries see 0 + the same products: the same bits as before)
        f4 init = splat(0.f);
        if constexpr (SHORT) {
            init = kinit[TJ[i]];
            if (NCT == 3 && i != 0 && i != 6) init = tvalid[i] ? init : splat(-1e30f);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            S[h][i] = OP::mfma16(Kh[h][TJ[i]], Qh[h][TI[i]], init);
            if constexpr (ASP) {                           // (k_hi + k_lo) . (q_hi + q_lo) without the lo . lo term
                S[h][i] = OP::mfma16(Klo[h][TJ[i]], Qh[h][TI[i]], S[h][i]);
                S[h][i] = OP::mfma16(Kh[h][TJ[i]], Qlo[h][TI[i]], S[h][i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < N7; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float hm = hmax(S[h][i]);                            // unconditional + select: the conditional form became a divergent branch
            tm[h][i] = (SHORT || tvalid[i]) ? hm : -INFINITY;
        }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = N7; i < 7; ++i) tm[h][i] = -INFINITY;
        mx[h][0] = xmax(vmax(tm[h][0], tm[h][1]));
        mx[h][1] = xmax(vmax3(tm[h][2], tm[h][3], tm[h][4]));
        mx[h][2] = xmax(vmax(tm[h][5], tm[h][6]));
    }
    f4 Lsum[2][3];
#pragma unroll
    for (int i = 0; i < N7; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f4 p;
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(S[h][i][r] - mx[h][TI[i]]);
            // full clips: the diagonal tiles (0,0) and (2,2) hold one clip's keys and queries only -- every probability counts
            // (their masks are all-ones by construction: 4 packed ANDs per block less)
            const bool all_valid = !SHORT && (i == 0 || i == 6);
            // off-diagonal tiles of full clips: a lane's four keys are all its query's clip's or all foreign, so the mask is the tile
            // validity -- a wave mask in scalar registers, where the per-element masks are 10 vector registers alive through the MLP loop
            const u2 pk = __builtin_bit_cast(u2, OP::pack4(p));
            const u2 bits = (all_valid || SHORT) ? pk : (tvalid[i] ? pk : u2{0u, 0u});     // (SHORT: a masked key's exponential is already 0)
            const v4 Pm = __builtin_bit_cast(v4, bits);
            // the first tile of a query tile starts its accumulators from the literal 0 (an inline constant of the MFMA: no v_mov per register)
            const bool first = i == 0 || i == 2 || i == 5;
            O[h][TI[i]] = OP::mfma16(Vh[TJ[i]][h], Pm, first ? splat(0.f) : O[h][TI[i]]);
            if constexpr (ASP) O[h][TI[i]] = OP::mfma16(Vlo[TJ[i]][h], Pm, O[h][TI[i]]);
            Lsum[h][TI[i]] = OP::mfma16(OP::ones4(), Pm, first ? splat(0.f) : Lsum[h][TI[i]]);
        }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int it = 0; it < NCT; ++it) O[h][it] = O[h][it] * splat(__builtin_amdgcn_rcpf(Lsum[h][it][0]));
    // proj: K = 32 = (head, d); slots j<4 head 0, j>=4 head 1 (same universal k-order as pack_frag)
    v8 A[8];
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) A[ft] = *(const v8*)(proj + ft * FRAG_BYTES + lane * 16);
    v8 OX[3];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) OX[ct] = OP::pack8(O[0][ct], O[1][ct]);
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) {
        const f4 bp = *(const f4*)(bias + BIAS_PROJ + 16 * ft + 4 * g);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) R[ft][ct] = OP::mfma32(A[ft], OX[ct], add4s(R[ft][ct], bp));
    }
    if constexpr (ASP) {                                  // precision f16a: the attention output of tile 0 = hi + lo
        const v8 OXlo = pack8_lo(O[0][0], O[1][0], OX[0]);
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) R[ft][0] = OP::mfma32(A[ft], OXlo, R[ft][0]);
    }
    if constexpr (X2) {                                   // precision f16x2: the low-order proj fragments follow the high-order ones
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) A[ft] = *(const v8*)(proj + (8 + ft) * FRAG_BYTES + lane * 16);
#pragma unroll
        for (int ft = 0; ft < 8; ++ft)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) R[ft][ct] = OP::mfma32(A[ft], OX[ct], R[ft][ct]);
    }
}

// ---- precision fp8 (BASELINE config #5): OCP e4m3 operands of v_mfma_scale_f32_16x16x128_f8f6f4 ---------------------
// four f32 -> four e4m3 bytes of one dword (round to nearest even; the conversion does NOT saturate -- 1000.f gives NaN 0x7f,
// tools/probe/mx_layout.hip -- so values are clamped to the format's +-448 first)
__device__ __forceinline__ int cvt4_fp8(f4 x) {                  // caller guarantees |x| <= 448
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], w, true);
}
__device__ __forceinline__ int relu_cvt4_fp8(f4 x) {             // relu and the clamp are one v_med3 per element
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(x[0], 0.f, 448.f), __builtin_amdgcn_fmed3f(x[1], 0.f, 448.f), 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(x[2], 0.f, 448.f), __builtin_amdgcn_fmed3f(x[3], 0.f, 448.f), w, true);
}
// a fragment8: two lane-linear 1 KiB halves (conflict-free ds_read_b128 each)
__device__ __forceinline__ v8i lds_frag8(const char* base, int lane) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i lo = *(const v4i*)(base + lane * 16), hi = *(const v4i*)(base + 1024 + lane * 16);
    return v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
constexpr int E8M0_ONE = 127;
// D = A.B + C with MX block scales; WSEL = byte of the weight operand's scale dword (the activation operand's scale is 1.0)
template <int WSEL>
__device__ __forceinline__ f4 mfma8_wa(v8i w, v8i x, f4 c, int wscale) {      // weights as A (rows = output features)
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, x, c, 0, 0, WSEL, wscale, 0, E8M0_ONE);
}
template <int WSEL>
__device__ __forceinline__ f4 mfma8_wb(v8i x, v8i w, f4 c, int wscale) {      // weights as B (columns = output features)
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(x, w, c, 0, 0, 0, E8M0_ONE, WSEL, wscale);
}
__device__ __forceinline__ f4 mfma8_wa_sel(int sel, v8i w, v8i x, f4 c, int wscale) {
    switch (sel & 3) { case 0: return mfma8_wa<0>(w, x, c, wscale); case 1: return mfma8_wa<1>(w, x, c, wscale);
                       case 2: return mfma8_wa<2>(w, x, c, wscale); default: return mfma8_wa<3>(w, x, c, wscale); }
}
// LN (mean-free on the centred stream, see ln_stats) -> the 128 features of each token column as 32 e4m3 bytes per lane.
// A LayerNorm output is bounded by sqrt(128) = 11.3 (its squares sum to 128), far inside e4m3's +-448: no clamp.
__device__ __forceinline__ void layernorm_to_fp8(const f4 (&R)[8][3], float eps, v8i (&X8)[3]) {
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
        float mean, rstd;
        ln_stats(R, ct, eps, mean, rstd);
        const f4 rs = splat(rstd);
#if defined(UITK_LN_TWO_PASS) || defined(UITK_LN_WITH_MEAN)
        const f4 nm = splat(-mean * rstd);
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) X8[ct][ft] = cvt4_fp8(R[ft][ct] * rs + nm);
#else
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) X8[ct][ft] = cvt4_fp8(R[ft][ct] * rs);
#endif
    }
}

// ---- exact (erf) GELU of the non-uit factories (nn.GELU(), models/uit.py:344): 0.5 x (1 + erf(x / sqrt 2)) with
// Abramowitz & Stegun 7.1.26 (|erf error| <= 1.5e-7): erf(z) = 1 - (a1 t + .. + a5 t^5) exp(-z^2), t = 1 / (1 + p z), z >= 0.
__device__ __forceinline__ float gelu1(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float y = p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);       // 1 - erf(z)
    return 0.5f * x * (x >= 0.f ? 2.0f - y : y);                                       // 1 + erf(x / sqrt 2) = 2 - y | y
}
// The same on register pairs (round 4): the polynomial, the products and the final combination are packed-f32 instructions (two
// elements each); 1 + erf(x / sqrt 2) = 2 - y for x >= 0 and y for x < 0 folds into  gelu = 0.5 (x + |x| - |x| y)  -- no select.
// Round 5: a packed-f32 instruction cannot take a literal, and the compiler kept the nine constant PAIRS in vector registers across
// the MLP loop -- the registers the GELU instantiations spilled inside the block loop (a scratch reload drains the weight ring).  The
// constants now live in SCALAR register pairs (both halves the constant; one scalar operand per instruction is allowed) and the
// instructions that take them are written out; 0.70710678 is folded into its two users (k1 = 0.3275911 / sqrt 2, k2 = -log2(e) / 2),
// so z = |x| / sqrt 2 is never formed.  Every asm operand is a VALU result (never an MFMA accumulator: DESIGN_LOG.md 2.2).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned long long sc2(float c) {                   // {c, c} as a 64-bit scalar
    const unsigned long long b = __float_as_uint(c);
    return b | (b << 32);
}
// Round 6: every one of these statements works IN PLACE on a register pair that a compiler-visible instruction (or another of these
// statements) wrote last.  With a free "=v" output the allocator may hand out a register that an MFMA issued just before still reads as
// SrcC (or has yet to write) -- a hazard the recogniser cannot pad for an asm statement: tools/kernel_isa.py --hazards found the GELU
// builds' v_pk_mul / v_pk_fma 0-2 wait states behind such MFMAs.  A tied operand's register holds a live value from its last visible
// write to the statement, so no MFMA can have it as a destination or a dying source in between.
__device__ __forceinline__ void pk_fma_acc_vs(f2& acc, f2 b, unsigned long long c) {   // acc = acc * b + c
    asm("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(b), "s"(c));
}
__device__ __forceinline__ void pk_fma_into(f2& acc, f2 a, unsigned long long c) {     // acc = a * c + acc
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(c));
}
__device__ __forceinline__ void pk_mul_acc_s(f2& acc, unsigned long long c) {          // acc = acc * c
    asm("v_pk_mul_f32 %0, %0, %1" : "+v"(acc) : "s"(c));
}
__device__ __forceinline__ f2 gelu2(f2 x) {
    const f2 a = f2{fabsf(x[0]), fabsf(x[1])};                                  // (compiler-visible first uses of the accumulators)
    const f2 xh = x * f2{0.5f, 0.5f};
    f2 d = a * f2{1.0f, 1.0f};                                                  // (a visible copy of a claims the register)
    pk_mul_acc_s(d, sc2(0.3275911f * 0.70710678118654752f));
    d = d + f2{1.0f, 1.0f};                                                     // 1 + p z
    const f2 t = f2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f2 p = t * f2{1.0f, 1.0f};                                                  // a visible copy of t (also the wait state a read of a fresh v_rcp result needs)
    pk_mul_acc_s(p, sc2(1.061405429f));
    pk_fma_acc_vs(p, f2{1.0f, 1.0f}, sc2(-1.453152027f));
    pk_fma_acc_vs(p, t, sc2(1.421413741f));
    pk_fma_acc_vs(p, t, sc2(-0.284496736f));
    pk_fma_acc_vs(p, t, sc2(0.254829592f));
    f2 e = a * a;
    pk_mul_acc_s(e, sc2(-0.5f * 1.4426950408889634f));                          // -log2(e) z^2, z = |x| / sqrt 2
    const f2 y = (p * t) * f2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};            // 1 - erf(z)
    return __builtin_elementwise_fma(a * f2{0.5f, 0.5f}, f2{1.0f, 1.0f} - y, xh);                     // 0.5 x + 0.5 |x| (1 - y)
}
__device__ __forceinline__ f4 gelu4(f4 x) {
    // the two halves ONE AFTER THE OTHER (round 6): left to the scheduler, the four chains of a tile pair ran interleaved and the short-clip
    // full-attention builds spilled 11-14 registers, 6 reloads inside the block loop -- once their asm statements had honest register
    // lifetimes (see hmax).  Sequenced: 2 spilled, none in any loop, the same encoder time (profiles/round6/gelu_seq_ab.log)
    const f2 lo = gelu2(f2{x[0], x[1]});
    __builtin_amdgcn_sched_barrier(0);
    const f2 hi = gelu2(f2{x[2], x[3]});
    return f4{lo[0], lo[1], hi[0], hi[1]};
}

// ---- precision f16l6 (round 3): the low-order weight term W - f16(W) as MX-fp6 (e2m3) operands of v_mfma_scale_f32_16x16x128_f8f6f4
// against fp6 / fp4 copies of the activations: one MFMA at the cost of one f16 k-step covers K = 128 (tools/probe/mx6_layout.hip:
// operand field order, row scales, conversion order and rounding, cycles per instruction).
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef _Float16 h16v __attribute__((ext_vector_type(16)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef unsigned u6 __attribute__((ext_vector_type(6)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int E8M0_TWO = 128;                                       // activations are converted with scale 2: e2m3 covers +-15, e2m1 +-12
__device__ __forceinline__ v8i pad6(u6 x) { return v8i{(int)x[0], (int)x[1], (int)x[2], (int)x[3], (int)x[4], (int)x[5], 0, 0}; }
__device__ __forceinline__ v8i pad4(u4 x) { return v8i{(int)x[0], (int)x[1], (int)x[2], (int)x[3], 0, 0, 0, 0}; }
__device__ __forceinline__ u6 lds_frag6(const char* base, int lane) {                 // ds_read_b128 + ds_read_b64, both lane-linear
    typedef unsigned u2x __attribute__((ext_vector_type(2)));
    const u4 lo = *(const u4*)(base + lane * 16);
    const u2x hi = *(const u2x*)(base + 1024 + lane * 8);
    return u6{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1]};
}
// the four k-step operands of a token tile (32 halves per lane) -> their fp6 copy (values / 2, saturating at +-7.5)
__device__ __forceinline__ u6 operand_to_fp6(h8 x0, h8 x1, h8 x2, h8 x3, float scale = 2.0f) {
    const h16v a = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const h16v b = __builtin_shufflevector(x2, x3, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const h32 x = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31);
    return __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x, scale);
}
// eight halves (one MLP chunk's hidden operand of a lane) -> one dword of eight e2m1 fields (values / 2, saturating at +-6)
__device__ __forceinline__ unsigned operand_to_fp4(h8 x) {
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, h2v{x[0], x[1]}, 2.0f, 0);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, h2v{x[2], x[3]}, 2.0f, 1);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, h2v{x[4], x[5]}, 2.0f, 2);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, h2v{x[6], x[7]}, 2.0f, 3);
    return w;
}
// D = Wlo . X + C; WSEL = byte of the row-scale dword that belongs to this fragment6
template <int WSEL> __device__ __forceinline__ f4 mfma6_wa(u6 w, u6 x, f4 c, int wscale) {      // fp6 weights as A, fp6 activations as B
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pad6(w), pad6(x), c, 2, 2, WSEL, wscale, 0, E8M0_TWO);
}
template <int WSEL> __device__ __forceinline__ f4 mfma6_wb(u6 x, u6 w, f4 c, int wscale) {      // fp6 activations as A, fp6 weights as B
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pad6(x), pad6(w), c, 2, 2, 0, E8M0_TWO, WSEL, wscale);
}
template <int WSEL> __device__ __forceinline__ f4 mfma64_wa(u6 w, u4 x, f4 c, int wscale) {     // fp6 weights as A, fp4 activations as B
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pad6(w), pad4(x), c, 2, 4, WSEL, wscale, 0, E8M0_TWO);
}
__device__ __forceinline__ f4 mfma6_wa_sel(int sel, u6 w, u6 x, f4 c, int wscale) {
    switch (sel & 3) { case 0: return mfma6_wa<0>(w, x, c, wscale); case 1: return mfma6_wa<1>(w, x, c, wscale);
                       case 2: return mfma6_wa<2>(w, x, c, wscale); default: return mfma6_wa<3>(w, x, c, wscale); }
}
__device__ __forceinline__ f4 mfma64_wa_sel(int sel, u6 w, u4 x, f4 c, int wscale) {
    switch (sel & 3) { case 0: return mfma64_wa<0>(w, x, c, wscale); case 1: return mfma64_wa<1>(w, x, c, wscale);
                       case 2: return mfma64_wa<2>(w, x, c, wscale); default: return mfma64_wa<3>(w, x, c, wscale); }
}
// Weight ring of precision f16l6: four 32 KiB slots, stages of four sizes (uitk_dev.h).  KIND 0 QK, 1 VP, 2 M, 3 MB.  Every wave issues
// the same number of 1 KiB LDS-DMAs for a given kind (a tail of fewer than 8 pieces is fetched by the waves modulo its length, i.e.
// some pieces twice), so the counted vmcnt is static per call site.
template <int KIND> constexpr int l6_dma_count() { return KIND == 0 ? 2 : KIND == 1 ? 3 : KIND == 2 ? 3 : 5; }
template <int KIND>
__device__ __forceinline__ void l6_issue(const char* src, char* dst, int wave, int lane) {
    const unsigned lane16 = (unsigned)lane * 16u;
    auto piece = [&](int p) { dma_b128(src + (size_t)(unsigned)(p * FRAG_BYTES), lane16, dst + p * FRAG_BYTES); };
    piece(wave * 2); piece(wave * 2 + 1);                            // the sixteen f16 fragments
    if constexpr (KIND == 1) piece(16 + wave);                       // 8 low-order f16 fragments of proj
    if constexpr (KIND >= 2) piece(16 + (wave >= 6 ? wave - 6 : wave >= 3 ? wave - 3 : wave));      // 2 fragment6 of fc1 = 3 KiB
    if constexpr (KIND == 3) { piece(19 + wave); piece(27 + (wave & 3)); }                          // 8 fragment6 of fc2's k-group = 12 KiB
}
// issue the stage two ahead, wait for this wave's share of the current one (OUTSTANDING = DMAs of the two younger stages), barrier
template <int KIND_ISSUE, int OUTSTANDING>
__device__ __forceinline__ void l6_step(const char* src, char* dst, int wave, int lane) {
    l6_issue<KIND_ISSUE>(src, dst, wave, lane);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OUTSTANDING) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

#if !defined(UITK_LAB) || !defined(L6_ABL)
#undef L6_ABL
#define L6_ABL 0      // -DUITK_LAB builds only (timing, wrong results; tools/mklib.sh): 1 = no fc1 low-order term, 2 = no fc2 low-order term,
#endif                // 4 = no proj low-order term.  The product build (uit_mobile_amd/build.py) never defines UITK_LAB.
// In-kernel stamps (diagnostic build only, TIMING = true): one asm statement = s_memtime + lgkmcnt(0), fenced.
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define TSTAMP(slot) do { if constexpr (TIMING) { const unsigned long long _t = stamp(); tacc[slot] += _t - tlast; tlast = _t; } } while (0)

// SHORT: clips with fewer than 6 time patches (masked padding tokens; a separate instantiation keeps the full-clip kernel's
// register allocation untouched).
// ATT: 0 = BNeckAttention (the uit_* family), 1 = full-width Attention (2 heads x 64; 20 ring stages per block).  ACT: 0 ReLU, 1 GELU.
// POOL: 0 = pooling 'mean', 1 = pooling 'dm' (models/uit.py:405-412), 2 = pooling 'token' (:389-392, 399-401): ONE clip per wave,
// column 0 = the cls token, columns 1..24 = the patch tokens, columns 25..47 dead (masked as keys, never read as outputs).
template <int PREC, int WAVES, bool SHORT, bool TIMING, int ATT = 0, int ACT = 0, int POOL = 0>
__global__ __launch_bounds__(WAVES * 64, 2) void uit_encoder_kernel(EncoderArgs a) {
    constexpr int OPC = PREC >= 2 ? 0 : PREC;              // precision fp8: patch embed, attention, proj and head stay f16; f16x2: fp16 operands
    using OP = Op<OPC>;
    using v8 = typename OP::v8;
    using v4 = typename OP::v4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // precision f16l6: four 32 KiB ring slots (the low-order patch-embed weights use the upper half of the ring during the prologue)
    constexpr bool L6 = PREC == 4;
    // instantiations without a vector register to spare issue their weight / bias DMAs in the scalar-base form (see ring_issue)
    constexpr bool SB = (PREC <= 1 || PREC == 5) && ACT != 0;           // (the GELU builds: the ReLU ones keep the builtin form -- it is the faster one where registers allow it)
    // PREC 5 (precision f16a, pooling 'token'): the f16 kernel with the MLP's two activation operands of column tile 0 -- the tile of the
    // cls token, the one row the head reads -- as hi + lo fp16 terms (include/uitk.h: UITK_PREC_F16A)
    // PREC 6 (precision f16ax, round 5): the same on the SPLIT weights of precision f16x2 (W = hi + lo) -- the top of pooling 'token''s ladder.
    // Both also carry q, k and v of EVERY column tile as hi + lo (Qlo / Klo / Vlo below): with one row read by the head, the fp16 rounding
    // of the keys was what carried the error of full-width attention on heavy-tailed checkpoints (tools/err_attrib_token.py)
    constexpr bool ASPLIT = PREC == 5 || PREC == 6;
    constexpr bool XW2 = PREC == 3 || PREC == 6;              // the split-weight stream and stage schedule of precision f16x2
    static_assert(!ASPLIT || POOL == 2, "precisions f16a / f16ax are built for pooling 'token'");
    // operand centring of LayerNorm-1 (uitk_dev.h: operand_centring; pack.cpp folds the same way): -mu[128] of a block in its bias table
    constexpr bool CEN = operand_centring(ATT, ACT, POOL, PREC);
    constexpr int NEGMU_AT = ATT ? BIASF_QKV + 128 : BIAS_NEGMU;
    constexpr int RING_BYTES = L6 ? 4 * L6_SLOT_BYTES : LDS_RING;
    static_assert(!L6 || (WAVES == 8 && RING_BYTES + LDS_BIAS + LDS_ZBUF + LDS_PEB <= ENC_LDS_W8), "f16l6: 8-wave workgroups, LDS budget");
    char* ring = smem;
    char* breg = smem + RING_BYTES;
    char* zbuf = smem + RING_BYTES + LDS_BIAS;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, c = lane & 15, g = lane >> 4;

    unsigned long long tacc[ENC_TIMING_SLOTS] = {}, tlast = 0;
    if constexpr (TIMING) tlast = stamp();
    // ---- rows (clip-crops) of this wave, token maps
    constexpr bool TOK = POOL == 2;
    // pooling 'token': ONE clip per wave = 25 live columns (cls + 24 patches) = column tiles 0 and 1; tile 2 would hold dead columns
    // only, so every per-tile loop stops at NCT = 2 and attention visits the four score tiles among tiles 0 / 1 (N7 = 4 of the 7)
    constexpr int NCT = TOK ? 2 : 3, N7 = TOK ? 4 : 7;
    constexpr bool PERKEY = SHORT || TOK;                   // softmax maxima with per-key masks (padding / dead columns exist)
    const int r0 = TOK ? blockIdx.x * WAVES + wave : (blockIdx.x * WAVES + wave) * 2;
    const int row_lo = r0 < a.rows ? r0 : a.rows - 1, row_hi = TOK ? row_lo : (r0 + 1 < a.rows ? r0 + 1 : a.rows - 1);
    int clip_[3], tk_[3], pf_[3], pt_[3];
    bool patch_[3];                                         // the column holds a patch token (TOK: not the cls token, not dead)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int col = 16 * ct + c;
        if constexpr (TOK) {
            clip_[ct] = 0;
            patch_[ct] = col >= 1 && col <= NTOK;
            tk_[ct] = patch_[ct] ? col - 1 : 0;                              // patch index (clamped for cls / dead columns)
            pf_[ct] = tk_[ct] / GT;
            pt_[ct] = patch_[ct] ? tk_[ct] - GT * pf_[ct] : (col == 0 ? 0 : 99);   // cls: always valid; dead: never
        } else {
            clip_[ct] = col >= NTOK ? 1 : 0;
            patch_[ct] = true;
            tk_[ct] = col - NTOK * clip_[ct];
            pf_[ct] = tk_[ct] / GT;
            pt_[ct] = tk_[ct] - GT * pf_[ct];
        }
    }
    // ---- issue ALL patch loads now (3 token tiles x 8 k-steps x 16 B per lane).  The front-end wrote each token's
    // 256 operands contiguously in k-slot order ((s,g,j) <-> mel 2s+(g>>1), frame 8(g&1)+j = offset 32s+8g+j),
    // already init_bn-normalised and converted, so a wave-instruction reads sixteen 64-byte runs and one HBM round
    // trip covers the whole patch embed.
    v8 XB[3][8];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const unsigned short* src = a.mel + ((int64_t)(clip_[ct] ? row_hi : row_lo) * NTOK + tk_[ct]) * 256 + 8 * g;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#if defined(UITK_LAB) && defined(ENC_NT_LOADS)
            XB[ct][s] = __builtin_nontemporal_load((const v8*)(src + 32 * s));      // lab: streaming hint for the once-read operands
#else
            XB[ct][s] = *(const v8*)(src + 32 * s);
#endif
        }
    }
    // The patch operands come from HBM (the slowest trip of the prologue): their loads are issued FIRST, the L2-resident weight /
    // table DMAs behind them.
    // ---- kick the weight stream: block-0 biases + the whole patch-embed weight set (4 stages = the whole ring)
    if (a.depth > 0) bias_issue<WAVES, SB>(a.bias, breg, 0, wave, lane);
#pragma unroll
    for (int i = 0; i < PE_HI_STAGES; ++i) ring_issue<WAVES, SB>(a.wstream, ring, i, wave, lane);
    // split patch embed (W = hi + lo): the low-order fragments go to their own 64 KiB behind the ring; 8-wave workgroups only
    // (a 4-wave workgroup shares its CU's LDS with a second one and runs the single-term patch embed)
    constexpr bool SPLIT_PE = WAVES == 8;
    char* pelo = L6 ? ring + PE_HI_STAGES * STAGE_BYTES : smem + ENC_LDS;
    if (SPLIT_PE && a.split_pe) {
        constexpr int PER_WAVE = STAGE_FRAGS / WAVES;
#pragma unroll
        for (int i = 0; i < PE_HI_STAGES; ++i)
#pragma unroll
            for (int k = 0; k < PER_WAVE; ++k) {
                const int piece = wave * PER_WAVE + k;
                __builtin_amdgcn_global_load_lds(GPTR(a.wstream + (size_t)(PE_HI_STAGES + i) * STAGE_BYTES + piece * FRAG_BYTES + lane * 16),
                                                 LPTR(pelo + i * STAGE_BYTES + piece * FRAG_BYTES), 16, 0, 0);
            }
    }
    // the conv-bias + positional table (25 x 128 f32) goes to LDS by DMA as well (8-wave workgroups): 24 per-lane global loads of
    // sixteen cache lines each were a quarter of the prologue's trip through the CU's vector-memory address path
    constexpr bool PEB_LDS = WAVES == 8;
    float* peb = (float*)(smem + (L6 ? RING_BYTES + LDS_BIAS + LDS_ZBUF : ENC_LDS_SPLIT_PE));
    if constexpr (PEB_LDS) {
#pragma unroll
        for (int i = 0; i < (NTOK + 1 + WAVES - 1) / WAVES; ++i) {
            const int row = wave + WAVES * i;                               // uniform
            if (row <= NTOK) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    __builtin_amdgcn_global_load_lds(GPTR(a.pe_bias + row * DIM + 64 * hf + lane), LPTR(peb + row * PEB_STRIDE + 64 * hf), 4, 0, 0);
            }
        }
        if (wave < 4) {                                                     // norm.w, norm.b, head_ln.w, head_ln.b: rows 25..28 (epilogue)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                __builtin_amdgcn_global_load_lds(GPTR(a.fin + wave * DIM + 64 * hf + lane), LPTR(peb + (NTOK + 1 + wave) * PEB_STRIDE + 64 * hf), 4, 0, 0);
        }
    }
    // final-norm / head-norm parameter row i (0 norm.w, 1 norm.b, 2 head_ln.w, 3 head_ln.b): LDS copy or global
    auto fin4 = [&](int i, int ft) -> f4 {
        if constexpr (PEB_LDS) return *(const f4*)(peb + (NTOK + 1 + i) * PEB_STRIDE + 16 * ft + 4 * g);
        else return *(const f4*)(a.fin + i * DIM + 16 * ft + 4 * g);
    };
    TSTAMP(12);                                                           // weight-stream kick

    f4 R[8][3];
    // init_bn scale / shift of mel bin `lane`: two coalesced dword loads per wave; the 24 (token tile, k-step) clamp
    // thresholds of a lane are then handed out through a wave-private 512-byte LDS table (zbuf is idle until the head)
    const float my_sc = a.bn_sc[lane], my_sh = a.bn_sh[lane];
    float amx = a.amax[lane];                                             // 64 slots, one per lane; -inf = untouched slot
    TSTAMP(13);                                                           // index math + issue of patch / table loads
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o));
    const float floor_db = amx - a.top_db;
    TSTAMP(8);                                                            // wait: every load of this wave has landed

    // ---- top_db clamp on the normalised operands: y = max(y, sc*floor + sh) for sc >= 0, min(...) for sc < 0
    // (exactly round(init_bn(max(dB, floor))): the affine map and the rounding are monotone).  All 8 operands of a
    // lane's k-step share one mel bin 16pf + 2s + (g>>1).  Masked tokens of short clips -> 0.
    float* clamp_tab = (float*)(zbuf + wave * 512);                        // [th | sc][pf][g>>1][s]: bin 16pf + 2s + (g>>1)
    {
        const int pos = (lane & 48) + 8 * (lane & 1) + ((lane & 15) >> 1);
        clamp_tab[pos] = floor_db * my_sc + my_sh;
        clamp_tab[64 + pos] = my_sc;
        __builtin_amdgcn_wave_barrier();                                  // same-wave LDS ops complete in order
        asm volatile("" ::: "memory");
    }
    if (OPC == 0 && a.bn_nonneg) {
        // every init_bn scale is >= 0 (decided on the host from the packed table): the clamp is ONE v_pk_max_f16 per register
        // against a per-(tile, k-step) threshold pair -- written as asm because the compiler canonicalises both inputs of an
        // fp16 max first (3 instructions per register) and the generic path below computes max AND min and selects
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const bool live = patch_[ct] && pt_[ct] < a.nt;               // cls / dead columns carry no patch: operand 0
            const float* trow = clamp_tab + 16 * pf_[ct] + 8 * (g >> 1);
            const f4 th_lo = *(const f4*)trow, th_hi = *(const f4*)(trow + 4);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float th = s < 4 ? th_lo[s & 3] : th_hi[s & 3];
                const _Float16 hth = (_Float16)th;
                const unsigned t2 = __builtin_bit_cast(unsigned, h2v{hth, hth});
                u4v x = __builtin_bit_cast(u4v, XB[ct][s]);
#pragma unroll
                for (int i = 0; i < 4; ++i) asm("v_pk_max_f16 %0, %0, %1" : "+v"(x[i]) : "v"(t2));       // (in place: see gelu2)
                if constexpr (PERKEY) { if (!live) x = u4v{0u, 0u, 0u, 0u}; }
                XB[ct][s] = __builtin_bit_cast(v8, x);
            }
        }
    } else {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const bool live = patch_[ct] && pt_[ct] < a.nt;                   // cls / dead columns carry no patch: operand 0
        const float* trow = clamp_tab + 16 * pf_[ct] + 8 * (g >> 1);
        const f4 th_lo = *(const f4*)trow, th_hi = *(const f4*)(trow + 4);
        const f4 sc_lo = *(const f4*)(trow + 64), sc_hi = *(const f4*)(trow + 68);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float sc = s < 4 ? sc_lo[s & 3] : sc_hi[s & 3], th = s < 4 ? th_lo[s & 3] : th_hi[s & 3];
            const v8 t8 = OP::from_f32x8(splat(th), splat(th));
            const v8 y = sc >= 0.f ? OP::max8(XB[ct][s], t8) : OP::min8(XB[ct][s], t8);
            XB[ct][s] = live ? y : OP::zero8();
        }
    }
    }
    // accumulator init (conv bias + positional embeddings), 4-wave workgroups (no LDS copy of the table): loaded HERE, behind the clamp
    // -- issued up front (round 4) the 96 accumulator registers were alive next to the 96 operand registers and the clamp's temporaries,
    // and the 4-wave instantiations spilled 57-68 vector registers around the prologue (36-57 scratch instructions, each a memory round
    // trip of its own); the table is L2-resident and its trip overlaps the wait for the patch-embed weights below
    if constexpr (!PEB_LDS) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) R[ft][ct] = *(const f4*)(a.pe_bias + (patch_[ct] ? tk_[ct] : NTOK) * DIM + 16 * ft + 4 * g);   // row 24: cls
    }
    TSTAMP(14);                                                           // clamp (+ its init_bn table loads)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // patch-embed weights landed ...
    __builtin_amdgcn_s_barrier();                                         // ... for every wave
    asm volatile("" ::: "memory");
    TSTAMP(9);                                                            // barrier: slowest wave of the workgroup
    if constexpr (PEB_LDS) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) R[ft][ct] = *(const f4*)(peb + (patch_[ct] ? tk_[ct] : NTOK) * PEB_STRIDE + 16 * ft + 4 * g);   // row 24: cls
    }

    // ---- patch embed: 64 LDS-resident fragments (ft = f/8, s = f%8) x 3 token tiles; accumulators start from the
    // conv bias + positional embeddings table
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) {
        v8 A[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) A[s] = lds_frag<OPC>(ring, ft * 8 + s, lane);
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) R[ft][ct] = OP::mfma32(A[s], XB[ct][s], R[ft][ct]);
        if (SPLIT_PE && a.split_pe) {
#pragma unroll
            for (int s = 0; s < 8; ++s) A[s] = lds_frag<OPC>(pelo, ft * 8 + s, lane);
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) R[ft][ct] = OP::mfma32(A[s], XB[ct][s], R[ft][ct]);
        }
    }
    TSTAMP(10);                                                           // patch-embed MFMAs
#if defined(UITK_LAB) && defined(ENC_DYNPRIO)
    if (tid < 4) ((unsigned*)(zbuf + 4096))[tid] = 0u;                    // the per-SIMD arrival counters of the lab variant
#endif
    __builtin_amdgcn_s_barrier();                                         // every wave is done with the PE weights
    asm volatile("" ::: "memory");
    if constexpr (L6) {
        const char* w0 = a.wstream + (size_t)PE_STAGES * STAGE_BYTES;
        l6_issue<0>(w0, ring, wave, lane);
        l6_issue<1>(w0 + L6_QK_BYTES, ring + L6_SLOT_BYTES, wave, lane);
    } else {
#pragma unroll
        for (int i = 0; i < PREFETCH; ++i) ring_issue<WAVES, SB>(a.wstream, ring, PE_STAGES + i, wave, lane);
    }

    auto dbg_store = [&](int stage) {                      // (debug path: recomputes its token map from v_mbcnt -- volatile, so that no
        const int dl = lane_id_now();                      //  lane-dependent address is hoisted out of the block loop and spilled there)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int col = 16 * ct + (dl & 15);
            const int dclip = (!TOK && col >= NTOK) ? 1 : 0, dtk = TOK ? (col >= 1 && col <= NTOK ? col - 1 : 0) : col - NTOK * dclip;
            const int row = r0 + dclip;
            if (row < a.rows && (!TOK || (col >= 1 && col <= NTOK))) {     // (pooling 'token': the cls and the dead columns are not part of the 24-row dump)
                float* dst = a.dbg + (((int64_t)row * (a.depth + 1) + stage) * NTOK + dtk) * DIM + 4 * (dl >> 4);
#pragma unroll
                for (int ft = 0; ft < 8; ++ft) *(f4*)(dst + 16 * ft) = R[ft][ct];
            }
        }
    };
    if (a.dbg != nullptr) dbg_store(0);

    // ---- attention masks (constant over blocks).  Score tiles (jt,it): S^T[j = 16jt+4g+r][i = 16it+c].
    //   tvalid[i]: this lane's 4 keys of tile i belong to the query's clip (uniform over r: 24 = 16 + 8 and the
    //              key index is 4g+r): wave masks in scalar registers;
    //   colmask  : (PERKEY: short clips / pooling 'token') bit j = key column j of the wave is a real token -- a patch token with
    //              time patch < nt, or the cls token -- as ONE 48-bit scalar.  Round 5: the per-element probability masks (14 vector
    //              registers alive through every block: what the short-clip instantiations spilled inside the block loop) are gone;
    //              a block rebuilds, from colmask and its lane id, the score accumulators' START values kinit[key tile][r] = 0 for a
    //              real key, -1e30 for padding (make_kinit; attention_proj / full_attention start their score MFMAs from them).
    bool tvalid[7];
#pragma unroll
    for (int i = 0; i < N7; ++i) {
        const int qc = (16 * TI[i] + c) >= NTOK ? 1 : 0;
        const int jc = (16 * TJ[i] + 4 * g) >= NTOK ? 1 : 0;
        tvalid[i] = TOK ? true : jc == qc;
    }
    unsigned long long colmask = ~0ull;
    if constexpr (PERKEY) {
        const unsigned nt6 = (1u << a.nt) - 1u;                                  // time patches 0 .. nt-1 of one frequency row
        const unsigned long long clip24 = (unsigned long long)(nt6 * 0x41041u);  // x (1 + 2^6 + 2^12 + 2^18): the clip's 4 x 6 tokens
        colmask = TOK ? ((clip24 << 1) | 1ull) : (clip24 | (clip24 << NTOK));    // TOK: column 0 = cls, 1..24 patches, the rest dead
    }
    auto make_kinit = [&](f4 (&kinit)[3]) {
        if constexpr (PERKEY) {
            const int gl = lane_id_now() >> 4;                                   // (volatile: re-derived per block, never a live register)
            const unsigned long long x = colmask >> (4 * gl);
#pragma unroll
            for (int tj = 0; tj < 3; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) kinit[tj][r] = ((x >> (16 * tj + r)) & 1ull) ? 0.f : -1e30f;
        } else {
#pragma unroll
            for (int tj = 0; tj < 3; ++tj) kinit[tj] = splat(0.f);
        }
    };

    // ---- transformer blocks
#pragma unroll 1
    for (int blk = 0; blk < a.depth; ++blk) {
      // ===== full-width Attention (models/uit.py:124-178), shared by precisions f16 / bf16 (XS2 = false) and f16x2 (XS2 = true).
      // Per head the row groups q_h, k_h, v_h (4 d-tiles x 4 k-steps each), scores between k_h and v_h (q and k die before v is produced),
      // then proj.  f16 / bf16: one ring stage of 16 fragments per row group (6 stages) + 2 stages of proj.  f16x2: every stage holds
      // the hi AND the lo fragments of HALF a group -- tile slot dl: fragments dl*8 + s (hi k-step s), dl*8 + 4 + s (lo) -- so an
      // accumulator takes its hi chain and its lo chain inside one stage (12 + 4 stages).  The LN1 operand is recomputed for head 1
      // instead of being kept alive across head 0's attention.
      auto full_attention = [&](auto xs2_tag, int tb, const float* bias, v8 (&X)[4][3], v8 (&Xlo1)[4]) {
        constexpr bool XS2 = decltype(xs2_tag)::value;
        static_assert(XS2 == XW2, "the split-weight schedule goes with the split-weight stream");
        constexpr int NSUB = XS2 ? 2 : 1, DTS = 4 / NSUB, KF = XS2 ? 8 : 4;      // stages per group, d-tiles per stage, fragments per tile
        // SB instantiations re-derive the lane's (c, g) from v_mbcnt where they are used (a bias-table offset): held across the block
        // they were spilled and reloaded behind a ring barrier
        auto lane_now = [&]() { if constexpr (SB) return lane_id_now(); else return lane; };
        v8 OX[4][3];                                                   // attention output, k-order of proj: k-step 2h + dt/2
        v8 OXlo[4];                                                    // precision f16a: its low-order term for column tile 0
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1) {
                layernorm_to_operand<OPC, NCT, CEN>(R, 1e-6f, X, bias + NEGMU_AT);
                if constexpr (ASPLIT) layernorm_lo_operand<CEN>(R, 1e-6f, X, Xlo1, bias + NEGMU_AT);
            }
            v4 Qp[4][3], Kp[4][3], Qlo[4][3], Klo[4][3];               // ASPLIT: q and k of every tile as hi + lo
#pragma unroll
            for (int which = 0; which < 2; ++which) {
#pragma unroll
                for (int sub = 0; sub < NSUB; ++sub) {
                    const int t = tb + (3 * h + which) * NSUB + sub;
                    ring_step<WAVES, SB>(a.wstream, ring, t, wave, lane);
                    if (h == 0 && which == 0 && sub == 0 && blk + 1 < a.depth) bias_issue<WAVES, SB>(a.bias, breg, blk + 1, wave, lane);
                    const char* slot = ring + (t & (RING_SLOTS - 1)) * STAGE_BYTES;
#pragma unroll
                    for (int dl = 0; dl < DTS; ++dl) {
                        const int dt = sub * DTS + dl;
                        f4 acc[3];
                        // (CEN: the K rows' bias is exactly zero -- its slot carries -mu)
                        const f4 bq = (CEN && which == 1) ? splat(0.f) : *(const f4*)(bias + BIASF_QKV + 128 * which + 64 * h + 16 * dt + 4 * (lane_now() >> 4));
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) acc[ct] = bq;
#pragma unroll
                        for (int part = 0; part < KF / 4; ++part) {            // hi k-steps, then (f16x2) the lo k-steps
                            v8 A[4];
#pragma unroll
                            for (int sk = 0; sk < 4; ++sk) A[sk] = lds_frag<OPC>(slot, dl * KF + 4 * part + sk, lane);
#pragma unroll
                            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                                for (int sk = 0; sk < 4; ++sk) acc[ct] = OP::mfma32(A[sk], X[sk][ct], acc[ct]);
                            if constexpr (ASPLIT) {                    // tile 0: + W . Xlo (the high-order weight fragments only)
                                if (part == 0) {
#pragma unroll
                                    for (int sk = 0; sk < 4; ++sk) acc[0] = OP::mfma32(A[sk], Xlo1[sk], acc[0]);
                                }
                            }
                        }
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) {
                            if (which == 0) Qp[dt][ct] = OP::pack4(acc[ct]); else Kp[dt][ct] = OP::pack4(acc[ct]);
                            if constexpr (ASPLIT) { if (which == 0) Qlo[dt][ct] = pack4_lo(acc[ct], Qp[dt][ct]); else Klo[dt][ct] = pack4_lo(acc[ct], Kp[dt][ct]); }
                        }
                    }
                }
            }
            // S^T = K.Q^T over the 64-wide head: four chained K = 16 MFMAs per score tile (log2 domain, scale folded into Wq)
            f4 S[7];
            float tm[7], mx[3];
            f4 kinit[3];
            make_kinit(kinit);                                 // PERKEY: 0 for a real key, -1e30 for padding (see the kernel's mask section)
#pragma unroll
            for (int i = 0; i < N7; ++i) {
                f4 acc = splat(0.f);
                if constexpr (PERKEY) {
                    acc = kinit[TJ[i]];
                    if (NCT == 3 && i != 0 && i != 6) acc = tvalid[i] ? acc : splat(-1e30f);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) acc = OP::mfma16(Kp[dt][TJ[i]], Qp[dt][TI[i]], acc);
                if constexpr (ASPLIT) {                                // (k_hi + k_lo) . (q_hi + q_lo) without the lo . lo term
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) acc = OP::mfma16(Klo[dt][TJ[i]], Qp[dt][TI[i]], acc);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) acc = OP::mfma16(Kp[dt][TJ[i]], Qlo[dt][TI[i]], acc);
                }
                S[i] = acc;
                const float hm = hmax(S[i]);
                tm[i] = (PERKEY || tvalid[i]) ? hm : -INFINITY;
            }
#pragma unroll
            for (int i = N7; i < 7; ++i) tm[i] = -INFINITY;
            mx[0] = xmax(vmax(tm[0], tm[1]));
            mx[1] = xmax(vmax3(tm[2], tm[3], tm[4]));
            mx[2] = xmax(vmax(tm[5], tm[6]));
            // full clips: the probabilities are formed and packed NOW (the scores die before the V stage: 14 registers instead of 28
            // live through it), masked by tile validity -- a lane's four keys of a tile are all its clip's or all foreign -- which
            // is a wave mask in scalar registers where the per-element masks are 14 vector registers
            v4 Pq[7];
            {
#pragma unroll
                for (int i = 0; i < N7; ++i) {
                    f4 p;
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(S[i][r] - mx[TI[i]]);
                    const u2 bits = __builtin_bit_cast(u2, OP::pack4(p));
                    if constexpr (PERKEY) Pq[i] = __builtin_bit_cast(v4, bits);             // (a masked key's exponential is already 0)
                    else Pq[i] = __builtin_bit_cast(v4, tvalid[i] ? bits : u2{0u, 0u});
                }
            }
            // v_h token-major (operand roles swapped: tokens are the rows)
            v4 Vp[3][4], Vlo[3][4];                                    // ASPLIT: v of every tile as hi + lo
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                const int t = tb + (3 * h + 2) * NSUB + sub;
                ring_step<WAVES, SB>(a.wstream, ring, t, wave, lane);
                const char* slot = ring + (t & (RING_SLOTS - 1)) * STAGE_BYTES;
#pragma unroll
                for (int dl = 0; dl < DTS; ++dl) {
                    const int dt = sub * DTS + dl;
                    f4 acc[3];
                    const f4 bv = splat(bias[BIASF_QKV + 256 + 64 * h + 16 * dt + (lane_now() & 15)]);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) acc[ct] = bv;
#pragma unroll
                    for (int part = 0; part < KF / 4; ++part) {
                        v8 Bw[4];
#pragma unroll
                        for (int sk = 0; sk < 4; ++sk) Bw[sk] = lds_frag<OPC>(slot, dl * KF + 4 * part + sk, lane);
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                            for (int sk = 0; sk < 4; ++sk) acc[ct] = OP::mfma32(X[sk][ct], Bw[sk], acc[ct]);
                        if constexpr (ASPLIT) {
                            if (part == 0) {
#pragma unroll
                                for (int sk = 0; sk < 4; ++sk) acc[0] = OP::mfma32(Xlo1[sk], Bw[sk], acc[0]);
                            }
                        }
                    }
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        Vp[ct][dt] = OP::pack4(acc[ct]);
                        if constexpr (ASPLIT) Vlo[ct][dt] = pack4_lo(acc[ct], Vp[ct][dt]);
                    }
                }
            }
            f4 O[4][3], Lsum[3];
#pragma unroll
            for (int it = 0; it < NCT; ++it) {
                Lsum[it] = splat(0.f);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) O[dt][it] = splat(0.f);
            }
#pragma unroll
            for (int i = 0; i < N7; ++i) {
                const v4 Pm = Pq[i];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) O[dt][TI[i]] = OP::mfma16(Vp[TJ[i]][dt], Pm, O[dt][TI[i]]);
                if constexpr (ASPLIT) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) O[dt][TI[i]] = OP::mfma16(Vlo[TJ[i]][dt], Pm, O[dt][TI[i]]);
                }
                Lsum[TI[i]] = OP::mfma16(OP::ones4(), Pm, Lsum[TI[i]]);
                // a query tile's output is scaled and packed as soon as its last key tile is in (TI is sorted): its sixteen fp32
                // accumulators become eight operand registers before the next query tile's accumulate (48 -> 16 live accumulators)
                if (i + 1 == N7 || TI[i + 1 < 7 ? i + 1 : 6] != TI[i]) {
                    const int it = TI[i];
                    const f4 inv = splat(__builtin_amdgcn_rcpf(Lsum[it][0]));
                    OX[2 * h][it] = OP::pack8(O[0][it] * inv, O[1][it] * inv);
                    OX[2 * h + 1][it] = OP::pack8(O[2][it] * inv, O[3][it] * inv);
                    if constexpr (ASPLIT) {
                        if (it == 0) {
                            OXlo[2 * h] = pack8_lo(O[0][0] * inv, O[1][0] * inv, OX[2 * h][0]);
                            OXlo[2 * h + 1] = pack8_lo(O[2][0] * inv, O[3][0] * inv, OX[2 * h + 1][0]);
                        }
                    }
                }
            }
            // head 0's output is not needed before proj: it waits in LDS -- the region of the prologue's low-order patch-embed weights,
            // idle in the block loop; 6 KiB per wave -- instead of in 24 registers through head 1 (the compiler spilled exactly these to
            // scratch, and a scratch reload is a vector-memory load: its wait drains the weight ring's DMAs in flight)
            if (h == 0) {
#pragma unroll
                for (int sk = 0; sk < 2; ++sk)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) *(v8*)(pelo + ((wave * 6 + sk * 3 + ct) * 64 + lane) * 16) = OX[sk][ct];
                if constexpr (ASPLIT) {                                // (pooling 'token': NCT = 2, the ct = 2 slots are free)
#pragma unroll
                    for (int sk = 0; sk < 2; ++sk) *(v8*)(pelo + ((wave * 6 + sk * 3 + 2) * 64 + lane) * 16) = OXlo[sk];
                }
            }
        }
#pragma unroll
        for (int sk = 0; sk < 2; ++sk)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) OX[sk][ct] = *(const v8*)(pelo + ((wave * 6 + sk * 3 + ct) * 64 + lane) * 16);
        if constexpr (ASPLIT) {
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) OXlo[sk] = *(const v8*)(pelo + ((wave * 6 + sk * 3 + 2) * 64 + lane) * 16);
        }
        // proj 128 -> 128: output tiles ft, four k-steps each (+ the lo k-steps): 2 stages of 4 tiles, or (f16x2) 4 stages of 2 tiles
        constexpr int PST = 2 * NSUB, FTS = 8 / PST;
#pragma unroll
        for (int ps = 0; ps < PST; ++ps) {
            const int t = tb + 6 * NSUB + ps;
            ring_step<WAVES, SB>(a.wstream, ring, t, wave, lane);
            const char* slot = ring + (t & (RING_SLOTS - 1)) * STAGE_BYTES;
#pragma unroll
            for (int fl = 0; fl < FTS; ++fl) {
                const int ft = FTS * ps + fl;
                const f4 bp = *(const f4*)(bias + BIASF_PROJ + 16 * ft + 4 * (lane_now() >> 4));
                f4 acc[3];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = add4s(R[ft][ct], bp);
#pragma unroll
                for (int part = 0; part < KF / 4; ++part) {
                    v8 A[4];
#pragma unroll
                    for (int sk = 0; sk < 4; ++sk) A[sk] = lds_frag<OPC>(slot, fl * KF + 4 * part + sk, lane);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int sk = 0; sk < 4; ++sk) acc[ct] = OP::mfma32(A[sk], OX[sk][ct], acc[ct]);
                    if constexpr (ASPLIT) {
                        if (part == 0) {
#pragma unroll
                            for (int sk = 0; sk < 4; ++sk) acc[0] = OP::mfma32(A[sk], OXlo[sk], acc[0]);
                        }
                    }
                }
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) R[ft][ct] = acc[ct];
            }
        }
      };
      if constexpr (PREC == 2) {
        // ================= precision fp8: 8 ring stages per block (uitk_dev.h) =================
        const int tb = PE_STAGES + blk * BLOCK_STAGES_FP8;
        const float* bias = (const float*)(breg + (blk & 1) * BIAS_FLOATS * 4);
        v8i X8[3];
        layernorm_to_fp8(R, 1e-6f, X8);
        // ----- stage 0: q (h0, h1), k (h0, h1) with the weights as A; v (h0, h1) token-major with the weights as B
        ring_step<WAVES, SB>(a.wstream, ring, tb, wave, lane);
        if (blk + 1 < a.depth) bias_issue<WAVES, SB>(a.bias, breg, blk + 1, wave, lane);
        v4 Qh[2][3], Kh[2][3], Vh[3][2];
        {
            const char* slot = ring + (tb & (RING_SLOTS - 1)) * STAGE_BYTES;
            const int sq0 = *(const int*)(slot + FP8_QKV_SCALES + lane * 4), sq1 = *(const int*)(slot + FP8_QKV_SCALES + 256 + lane * 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const v8i A = lds_frag8(slot + t * FRAG8_BYTES, lane);
                const f4 bq = *(const f4*)(bias + BIAS_QKV + 16 * t + 4 * g);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const f4 acc = mfma8_wa_sel(t, A, X8[ct], bq, sq0);
                    if (t < 2) Qh[t][ct] = OP::pack4(acc); else Kh[t - 2][ct] = OP::pack4(acc);
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const v8i Bw = lds_frag8(slot + (4 + h) * FRAG8_BYTES, lane);
                const f4 bv = splat(bias[BIAS_QKV + 64 + 16 * h + c]);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    Vh[ct][h] = OP::pack4(h == 0 ? mfma8_wb<0>(X8[ct], Bw, bv, sq1) : mfma8_wb<1>(X8[ct], Bw, bv, sq1));
            }
        }
        // ----- stage 1: attention (f16, as in the 16-bit modes) + proj (f16 fragments); MLP scale dwords
        ring_step<WAVES, SB>(a.wstream, ring, tb + 1, wave, lane);
        int s1[6], s2[6];
        {
            const char* slot = ring + ((tb + 1) & (RING_SLOTS - 1)) * STAGE_BYTES;
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                s1[d] = *(const int*)(slot + FP8_MLP_SCALES + d * 256 + lane * 4);
                s2[d] = *(const int*)(slot + FP8_MLP_SCALES + (6 + d) * 256 + lane * 4);
            }
            f4 kinit[3];
            make_kinit(kinit);
            attention_proj<OP, PERKEY, false, NCT>(R, Qh, Kh, Vh, tvalid, kinit, slot, bias, lane, g);
        }
        // ----- MLP: three k-groups of 128 hidden features: fc1 tiles 8q..8q+7 -> relu -> e4m3 -> fc2 k-group q
        layernorm_to_fp8(R, 1e-6f, X8);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            ring_step<WAVES, SB>(a.wstream, ring, tb + 2 + 2 * q, wave, lane);
            v8i HX8[3];
            {
                const char* slot = ring + ((tb + 2 + 2 * q) & (RING_SLOTS - 1)) * STAGE_BYTES;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const v8i A = lds_frag8(slot + i * FRAG8_BYTES, lane);
                    const f4 b1 = *(const f4*)(bias + BIAS_FC1 + 16 * (8 * q + i) + 4 * g);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) HX8[ct][i] = relu_cvt4_fp8(mfma8_wa_sel(i, A, X8[ct], b1, s1[2 * q + (i >> 2)]));
                }
            }
            ring_step<WAVES, SB>(a.wstream, ring, tb + 3 + 2 * q, wave, lane);
            {
                const char* slot = ring + ((tb + 3 + 2 * q) & (RING_SLOTS - 1)) * STAGE_BYTES;
#pragma unroll
                for (int ft = 0; ft < 8; ++ft) {
                    const v8i A = lds_frag8(slot + ft * FRAG8_BYTES, lane);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) R[ft][ct] = mfma8_wa_sel(ft, A, HX8[ct], R[ft][ct], s2[2 * q + (ft >> 2)]);
                }
            }
        }
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
            const f4 b2 = *(const f4*)(bias + BIAS_FC2 + 16 * ft + 4 * g);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) R[ft][ct] = add4s(R[ft][ct], b2);
        }
        if (a.dbg != nullptr) {
            dbg_store(blk + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      } else if constexpr (XW2) {
        // ================= precision f16x2 (and f16ax = f16x2 + the split activation operands of f16a): W = hi + lo, 28 ring stages per block (uitk_dev.h) =================
        // Every stage carries the hi and the lo fragments of the rows it covers; an accumulator takes the hi chain, then the lo
        // chain, and is finished (packed / ReLU'd) inside the stage.  Activations are the fp16 operands of precision f16.
        // Full-width attention (ATT = 1, round 3): 16 attention stages (full_attention above) + the same 24 MLP stages = 40 per block.
        constexpr int BLK_ST = ATT ? BLOCK_STAGES_X2_ATT : BLOCK_STAGES_X2, MLP0 = ATT ? 16 : 4;
        constexpr int B_FC1 = ATT ? BIASF_FC1 : BIAS_FC1, B_FC2 = ATT ? BIASF_FC2 : BIAS_FC2;
        const int tb = PE_STAGES + blk * BLK_ST;
        const float* bias = (const float*)(breg + (blk & 1) * BIAS_FLOATS * 4);
        v8 X[4][3];
        layernorm_to_operand<OPC, NCT, CEN>(R, 1e-6f, X, bias + NEGMU_AT);
        v8 Xlo1[4];                                                    // precision f16ax: the low-order term of tile 0's LayerNorm-1 operand
        if constexpr (ASPLIT) layernorm_lo_operand<CEN>(R, 1e-6f, X, Xlo1, bias + NEGMU_AT);
        if constexpr (ATT) {
        full_attention(std::true_type{}, tb, bias, X, Xlo1);
        } else {
        v4 Qh[2][3], Kh[2][3], Vh[3][2], Qlo[2][3], Klo[2][3], Vlo[3][2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {                             // stage 0: q h0, q h1; stage 1: k h0, k h1
            ring_step<WAVES, SB>(a.wstream, ring, tb + half, wave, lane);
            if (half == 0 && blk + 1 < a.depth) bias_issue<WAVES, SB>(a.bias, breg, blk + 1, wave, lane);
            const char* slot = ring + ((tb + half) & (RING_SLOTS - 1)) * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                v8 A[8];
#pragma unroll
                for (int f = 0; f < 8; ++f) A[f] = lds_frag<OPC>(slot, 8 * i + f, lane);
                const f4 bq = *(const f4*)(bias + BIAS_QKV + 16 * (2 * half + i) + 4 * g);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    f4 acc = bq;
#pragma unroll
                    for (int s = 0; s < 8; ++s) acc = OP::mfma32(A[s], X[s & 3][ct], acc);
                    if constexpr (ASPLIT) {
                        // tile 0: + W . Xlo on an accumulator of its own (one packed add).  Round 5 shipped this form because the CHAINED one (as
                        // the other precisions' stages do it; -DUITK_LAB -DAX_CHAINED) returned NaN / 2.4e-3 on some inputs, "not explained".
                        // Round 6 explains it (profiles/round6/f16ax_race.log): not this stage at all -- with the chained schedule the allocator put
                        // the asm `v_max3_f32` of the softmax on a dead row of the ones-row MFMA's destination one instruction behind that MFMA, and
                        // its late write-back replaced a query tile's maximum by a row sum.  Every asm statement now works in place on a register
                        // a visible instruction wrote (vmax / hmax / lane_id_now / gelu2) and tests/test_build_checks.py holds the shipped ISA free
                        // of such sites; the chained lab build is right now too (6.4e-5, bit-stable under load).  The form stays as shipped.
                        if (ct == 0) {
#if defined(UITK_LAB) && defined(AX_CHAINED)
#pragma unroll
                            for (int s = 0; s < 4; ++s) acc = OP::mfma32(A[s], Xlo1[s], acc);      // lab: the chained form (profiles/round5/f16ax_bisect.log)
#else
                            f4 accl = splat(0.f);
#pragma unroll
                            for (int s = 0; s < 4; ++s) accl = OP::mfma32(A[s], Xlo1[s], accl);
                            acc = acc + accl;
#endif
                        }
                    }
                    if (half == 0) Qh[i][ct] = OP::pack4(acc); else Kh[i][ct] = OP::pack4(acc);
                    if constexpr (ASPLIT) { if (half == 0) Qlo[i][ct] = pack4_lo(acc, Qh[i][ct]); else Klo[i][ct] = pack4_lo(acc, Kh[i][ct]); }
                }
            }
        }
        ring_step<WAVES, SB>(a.wstream, ring, tb + 2, wave, lane);             // stage 2: V (weights as the B operand: V token-major)
        {
            const char* slot = ring + ((tb + 2) & (RING_SLOTS - 1)) * STAGE_BYTES;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                v8 Bw[8];
#pragma unroll
                for (int f = 0; f < 8; ++f) Bw[f] = lds_frag<OPC>(slot, 8 * h + f, lane);
                const f4 bv = splat(bias[BIAS_QKV + 64 + 16 * h + c]);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    f4 acc = bv;
#pragma unroll
                    for (int s = 0; s < 8; ++s) acc = OP::mfma32(X[s & 3][ct], Bw[s], acc);
                    if constexpr (ASPLIT) {
                        if (ct == 0) {
#pragma unroll
                            for (int s = 0; s < 4; ++s) acc = OP::mfma32(Xlo1[s], Bw[s], acc);
                        }
                    }
                    Vh[ct][h] = OP::pack4(acc);
                    if constexpr (ASPLIT) Vlo[ct][h] = pack4_lo(acc, Vh[ct][h]);
                }
            }
        }
        ring_step<WAVES, SB>(a.wstream, ring, tb + 3, wave, lane);             // stage 3: attention + proj (hi, lo)
        f4 kinit[3];
        make_kinit(kinit);
        attention_proj<OP, PERKEY, true, NCT, ASPLIT>(R, Qh, Kh, Vh, tvalid, kinit, ring + ((tb + 3) & (RING_SLOTS - 1)) * STAGE_BYTES, bias, lane, g, Qlo, Klo, Vlo);
        }
        layernorm_to_operand<OPC, NCT>(R, 1e-6f, X);
        v8 Xlo[4], HXlo = OP::zero8();                                 // ASPLIT: the low-order terms of tile 0's two MLP operands
        if constexpr (ASPLIT) layernorm_lo_operand<false>(R, 1e-6f, X, Xlo, nullptr);
#pragma unroll 1
        for (int ch = 0; ch < 12; ++ch) {
            v8 HX[3];
            ring_step<WAVES, SB>(a.wstream, ring, tb + MLP0 + 2 * ch, wave, lane);  // fc1 rows 32ch .. 32ch+31
            {
                const char* slot = ring + ((tb + MLP0 + 2 * ch) & (RING_SLOTS - 1)) * STAGE_BYTES;
                f4 H[2][3];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    v8 A[8];
#pragma unroll
                    for (int f = 0; f < 8; ++f) A[f] = lds_frag<OPC>(slot, 8 * i + f, lane);
                    const f4 b1 = *(const f4*)(bias + B_FC1 + 32 * ch + 16 * i + 4 * g);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        H[i][ct] = b1;
#pragma unroll
                        for (int s = 0; s < 8; ++s) H[i][ct] = OP::mfma32(A[s], X[s & 3][ct], H[i][ct]);
                        if constexpr (ASPLIT) {
                            if (ct == 0) {
#pragma unroll
                                for (int s = 0; s < 4; ++s) H[i][0] = OP::mfma32(A[s], Xlo[s], H[i][0]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    if constexpr (ASPLIT) {
                        if (ct == 0) {                                 // tile 0: the activation as hi + lo
                            f4 a0, a1;
                            if constexpr (ACT == 1) { a0 = gelu4(H[0][0]); a1 = gelu4(H[1][0]); }
                            else { a0 = f4{fmaxf(H[0][0][0], 0.f), fmaxf(H[0][0][1], 0.f), fmaxf(H[0][0][2], 0.f), fmaxf(H[0][0][3], 0.f)};
                                   a1 = f4{fmaxf(H[1][0][0], 0.f), fmaxf(H[1][0][1], 0.f), fmaxf(H[1][0][2], 0.f), fmaxf(H[1][0][3], 0.f)}; }
                            HX[0] = OP::pack8(a0, a1);
                            HXlo = pack8_lo(a0, a1, __builtin_bit_cast(h8, HX[0]));
                            continue;
                        }
                    }
                    if constexpr (ACT == 1) HX[ct] = OP::pack8(gelu4(H[0][ct]), gelu4(H[1][ct]));
                    else HX[ct] = OP::relu_pack8(H[0][ct], H[1][ct]);
                }
            }
            ring_step<WAVES, SB>(a.wstream, ring, tb + MLP0 + 1 + 2 * ch, wave, lane);     // fc2 k-step ch
            {
                const char* slot = ring + ((tb + MLP0 + 1 + 2 * ch) & (RING_SLOTS - 1)) * STAGE_BYTES;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    v8 A2[8];
#pragma unroll
                    for (int ft = 0; ft < 8; ++ft) A2[ft] = lds_frag<OPC>(slot, 8 * half + ft, lane);
#pragma unroll
                    for (int ft = 0; ft < 8; ++ft)
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) R[ft][ct] = OP::mfma32(A2[ft], HX[ct], R[ft][ct]);
                    if constexpr (ASPLIT) {                            // tile 0's low-order activation term on the high-order fc2 fragments
                        if (half == 0) {
#pragma unroll
                            for (int ft = 0; ft < 8; ++ft) R[ft][0] = OP::mfma32(A2[ft], HXlo, R[ft][0]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
            const f4 b2 = *(const f4*)(bias + B_FC2 + 16 * ft + 4 * g);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) R[ft][ct] = add4s(R[ft][ct], b2);
        }
        if (a.dbg != nullptr) {
            dbg_store(blk + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      } else if constexpr (PREC == 4) {
        // ================= precision f16l6: the f16 block + the low-order weight term on the MX-fp6 MFMA (uitk_dev.h) =================
        // Stage k of block blk lives in ring slot (2 blk + k) & 3 (14 stages per block); at step k the stage k + 2 is issued.
        static_assert(ATT == 0 && ACT == 0, "precision f16l6: BNeckAttention + ReLU (any pooling)");
        // lane-derived values are re-derived per block from v_mbcnt (volatile: not hoisted): held across the block loop they are the
        // register allocator's favourite spill candidates, and a reload right behind a ring step drains the DMAs just issued
        const int lane = lane_id_now(), c = lane & 15, g = lane >> 4;
        const char* wblk = a.wstream + (size_t)PE_STAGES * STAGE_BYTES + (size_t)blk * L6_BLOCK_BYTES;
        const int sb = 2 * (blk & 1);
        auto slot_of = [&](int k) -> char* { return ring + ((sb + k) & 3) * L6_SLOT_BYTES; };
        const float* bias = (const float*)(breg + (blk & 1) * BIAS_FLOATS * 4);
        const int* sct = (const int*)(bias + L6_SCALES) + c;              // row-scale dwords [quad][row = c]
        v8 X[4][3];
        layernorm_to_operand<OPC, NCT>(R, 1e-6f, X);
        // ===== stage QK (issue MLP stage 0; younger stages VP + M).  qkv keeps single fp16 weights: its rounding is the smallest of the
        // four weight terms and the stage has no register room for a second operand set
        l6_step<2, l6_dma_count<1>() + l6_dma_count<2>()>(wblk + l6_mlp_offset(0), slot_of(2), wave, lane);
        if (blk + 1 < a.depth) bias_issue<WAVES, true>(a.bias, breg, blk + 1, wave, lane);
        v4 Qh[2][3], Kh[2][3];
        {
            const char* slot = slot_of(0);
            v8 A[16];
#pragma unroll
            for (int f = 0; f < 8; ++f) A[f] = lds_frag<OPC>(slot, f, lane);
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                if (ft == 1) {
#pragma unroll
                    for (int f = 8; f < 16; ++f) A[f] = lds_frag<OPC>(slot, f, lane);
                }
                const f4 bq = *(const f4*)(bias + BIAS_QKV + 16 * ft + 4 * g);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    f4 acc = bq;
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc = OP::mfma32(A[ft * 4 + s], X[s][ct], acc);
                    if (ft < 2) Qh[ft][ct] = OP::pack4(acc); else Kh[ft - 2][ct] = OP::pack4(acc);
                }
            }
        }
        // ===== stage VP (issue MLP stage 1; younger stages M + M); proj on split fp16 weights (K = 32: 24 more MFMAs)
        l6_step<2, 2 * l6_dma_count<2>()>(wblk + l6_mlp_offset(1), slot_of(3), wave, lane);
        {
            const char* slot = slot_of(1);
            v4 Vh[3][2];
            {
                v8 Bw[8];
#pragma unroll
                for (int f = 0; f < 8; ++f) Bw[f] = lds_frag<OPC>(slot, f, lane);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f4 bv = splat(bias[BIAS_QKV + 64 + 16 * h + c]);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        f4 acc = bv;
#pragma unroll
                        for (int s = 0; s < 4; ++s) acc = OP::mfma32(X[s][ct], Bw[h * 4 + s], acc);
                        Vh[ct][h] = OP::pack4(acc);
                    }
                }
            }
            f4 kinit[3];
            make_kinit(kinit);
            attention_proj<OP, PERKEY, !(L6_ABL & 4), NCT>(R, Qh, Kh, Vh, tvalid, kinit, slot + 8 * FRAG_BYTES, bias, lane, g);
        }
        // ===== MLP: 12 stages in three k-groups of four.  fc1: four fp16 k-steps + ONE fp6 MFMA per accumulator; fc2: the fp16 k-step of
        // the chunk, and at the last stage of a group (MB) the group's low-order term on the fp4 copy of its 128 hidden activations.
        // fc2 runs in its own chunk's stage (not under the next chunk's fragment reads as in precision f16): carrying its fragments
        // across the barrier costs 44 registers this mode does not have -- any spill inside the loop would drain the counted vmcnt.
        layernorm_to_operand<OPC, NCT>(R, 1e-6f, X);
        u4 HX4[3];
        // the fp6 copy of the LayerNorm operand is converted ONCE per block and held (18 registers): converting it per chunk instead
        // (3 v_cvt_scalef32_pk32_fp6_f16 of ~16 issue cycles each) was measured at +5 % encoder time
        u6 X6[3];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) X6[ct] = operand_to_fp6(X[0][ct], X[1][ct], X[2][ct], X[3][ct]);
        if constexpr (NCT == 3) asm volatile("" : "+v"(X6[0]), "+v"(X6[1]), "+v"(X6[2]));        // converted HERE, once per block (not re-materialised per chunk)
        else asm volatile("" : "+v"(X6[0]), "+v"(X6[1]));
#pragma unroll 1
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ch = 4 * q + j;
                // ring step: stage ch + 2 of the MLP, or the next block's QK / VP
                if (j == 0) l6_step<2, 2 * l6_dma_count<2>()>(wblk + l6_mlp_offset(ch + 2), slot_of(ch + 4), wave, lane);
                else if (j == 1) l6_step<3, l6_dma_count<2>() + l6_dma_count<3>()>(wblk + l6_mlp_offset(ch + 2), slot_of(ch + 4), wave, lane);
                else if (j == 2) {
                    if (q < 2) l6_step<2, l6_dma_count<3>() + l6_dma_count<2>()>(wblk + l6_mlp_offset(ch + 2), slot_of(ch + 4), wave, lane);
                    else l6_step<0, l6_dma_count<3>() + l6_dma_count<0>()>(wblk + L6_BLOCK_BYTES, slot_of(ch + 4), wave, lane);
                } else {
                    if (q < 2) l6_step<2, 2 * l6_dma_count<2>()>(wblk + l6_mlp_offset(ch + 2), slot_of(ch + 4), wave, lane);
                    else l6_step<1, l6_dma_count<0>() + l6_dma_count<1>()>(wblk + L6_BLOCK_BYTES + L6_QK_BYTES, slot_of(ch + 4), wave, lane);
                }
                const char* slot = slot_of(ch + 2);
                v8 HX[3];
                {
                    const u6 W6a = lds_frag6(slot + STAGE_BYTES, lane), W6b = lds_frag6(slot + STAGE_BYTES + FRAG6_BYTES, lane);
                    const int s1 = sct[(2 * q + (j >> 1)) * 16];      // row scales of fc1 tiles 4 (2q + j/2) .. +3
                    const f4 b10 = *(const f4*)(bias + BIAS_FC1 + 32 * ch + 4 * g), b11 = *(const f4*)(bias + BIAS_FC1 + 32 * ch + 16 + 4 * g);
                    // NT fc1 row tiles at a time (NT = 1: only 12 accumulator registers are live; the accumulators are packed, ReLU on the
                    // packed halves, before the next tile starts).  Each chain starts with its low-order MFMA: that one needs no f16
                    // fragment, so it and the fp6 conversions run while the fragment reads are still in flight.  The f16 k-steps are
                    // written k-step-major so that the 3 NT chains interleave.
                    constexpr int NT = 1;
                    v4 P[2][3];
#pragma unroll
                    for (int t0 = 0; t0 < 2; t0 += NT) {
                        v8 A[8];                                      // this tile's f16 fragments (read per tile: 16 registers, not 32)
#pragma unroll
                        for (int f = 4 * t0; f < 4 * (t0 + NT); ++f) A[f] = lds_frag<OPC>(slot, f, lane);
                        f4 H[NT][3];
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                            for (int ct = 0; ct < NCT; ++ct) {
                                H[tt][ct] = t0 + tt == 0 ? b10 : b11;
#if !(L6_ABL & 1)
                                H[tt][ct] = mfma6_wa_sel(2 * j + t0 + tt, t0 + tt == 0 ? W6a : W6b, X6[ct], H[tt][ct], s1);
#endif
                            }
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                                for (int ct = 0; ct < NCT; ++ct) H[tt][ct] = OP::mfma32(A[(t0 + tt) * 4 + s], X[s][ct], H[tt][ct]);
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                            for (int ct = 0; ct < NCT; ++ct) {
                                P[t0 + tt][ct] = __builtin_elementwise_max(OP::pack4(H[tt][ct]), v4{0, 0, 0, 0});
                                if (NT == 1 && t0 == 0) asm volatile("" : "+v"(P[0][ct]));
                            }
                    }
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        HX[ct] = __builtin_shufflevector(P[0][ct], P[1][ct], 0, 1, 2, 3, 4, 5, 6, 7);
                        unsigned w4 = (L6_ABL & 2) ? 0u : operand_to_fp4(HX[ct]);
                        asm volatile("" : "+v"(w4));                  // convert HERE: the scheduler must not keep four chunks of f16 operands alive instead
                        HX4[ct][j] = w4;
                    }
                }
                {
                    v8 A2[8];
#pragma unroll
                    for (int f = 0; f < 8; ++f) A2[f] = lds_frag<OPC>(slot, 8 + f, lane);
#pragma unroll
                    for (int ft = 0; ft < 8; ++ft)
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) R[ft][ct] = OP::mfma32(A2[ft], HX[ct], R[ft][ct]);
                }
                if (j == 3 && !(L6_ABL & 2)) {                        // fc2 low-order term of k-group q
                    const int s2a = sct[(6 + 2 * q) * 16], s2b = sct[(7 + 2 * q) * 16];
#pragma unroll
                    for (int ft = 0; ft < 8; ++ft) {
                        const u6 W6 = lds_frag6(slot + L6_M_BYTES + ft * FRAG6_BYTES, lane);
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) R[ft][ct] = mfma64_wa_sel(ft, W6, HX4[ct], R[ft][ct], ft < 4 ? s2a : s2b);
                    }
                }
            }
        }
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
            const f4 b2 = *(const f4*)(bias + BIAS_FC2 + 16 * ft + 4 * g);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) R[ft][ct] = add4s(R[ft][ct], b2);
        }
        if (a.dbg != nullptr) {
            dbg_store(blk + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      } else {
        constexpr int BLK_ST = ATT ? BLOCK_STAGES_ATT : BLOCK_STAGES, MLP0 = ATT ? 8 : 2;      // stages per block, first MLP stage
        constexpr int B_FC1 = ATT ? BIASF_FC1 : BIAS_FC1, B_FC2 = ATT ? BIASF_FC2 : BIAS_FC2;
        const int tb = PE_STAGES + blk * BLK_ST;
        const float* bias = (const float*)(breg + (blk & 1) * BIAS_FLOATS * 4);
        v8 X[4][3];
        layernorm_to_operand<OPC, NCT, CEN>(R, 1e-6f, X, bias + NEGMU_AT);
        // precision f16a (round 5): every activation operand of column tile 0 -- the tile of the cls row, the one row pooling 'token' reads
        // -- is carried as hi + lo fp16 terms: the LayerNorm-1 operand (Xlo1: into q, k, v), q, the attention output, the LayerNorm-2
        // operand and the MLP's hidden activations (below)
        v8 Xlo1[4];
        if constexpr (ASPLIT) layernorm_lo_operand<CEN>(R, 1e-6f, X, Xlo1, bias + NEGMU_AT);
        TSTAMP(0);                                                     // LN1

        if constexpr (ATT) {
        full_attention(std::false_type{}, tb, bias, X, Xlo1);
        } else {
        // ===== stage QK: fragments (ft = f/4 in {q h0, q h1, k h0, k h1}, s = f%4)
        ring_step<WAVES, SB>(a.wstream, ring, tb, wave, lane);
        TSTAMP(1);                                                     // ring wait QK
        if (blk + 1 < a.depth) bias_issue<WAVES, SB>(a.bias, breg, blk + 1, wave, lane);
        v4 Qh[2][3], Kh[2][3], Qlo[2][3], Klo[2][3];
        {
            const char* slot = ring + (tb & (RING_SLOTS - 1)) * STAGE_BYTES;
            v8 A[16];
#pragma unroll
            for (int f = 0; f < 8; ++f) A[f] = lds_frag<OPC>(slot, f, lane);
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                if (ft == 1) {
#pragma unroll
                    for (int f = 8; f < 16; ++f) A[f] = lds_frag<OPC>(slot, f, lane);
                }
                const f4 bq = *(const f4*)(bias + BIAS_QKV + 16 * ft + 4 * g);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    f4 acc = bq;
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc = OP::mfma32(A[ft * 4 + s], X[s][ct], acc);
                    if constexpr (ASPLIT) {
                        if (ct == 0) {
#pragma unroll
                            for (int s = 0; s < 4; ++s) acc = OP::mfma32(A[ft * 4 + s], Xlo1[s], acc);
                        }
                    }
                    if (ft < 2) Qh[ft][ct] = OP::pack4(acc); else Kh[ft - 2][ct] = OP::pack4(acc);
                    if constexpr (ASPLIT) { if (ft < 2) Qlo[ft][ct] = pack4_lo(acc, Qh[ft][ct]); else Klo[ft - 2][ct] = pack4_lo(acc, Kh[ft - 2][ct]); }
                }
            }
        }

        TSTAMP(2);                                                     // QK compute
        // ===== stage VP: fragments 0..7 = V (h = f/4, s = f%4) used as the B operand; 8..15 = proj (ft = f-8)
        ring_step<WAVES, SB>(a.wstream, ring, tb + 1, wave, lane);
        TSTAMP(3);                                                     // ring wait VP
        {
            const char* slot = ring + ((tb + 1) & (RING_SLOTS - 1)) * STAGE_BYTES;
            v4 Vh[3][2], Vlo[3][2];            // V[token 16ct+4g+r][d = c] -> A operand of P.V (rows d, k = token)
            {
                v8 Bw[8];
#pragma unroll
                for (int f = 0; f < 8; ++f) Bw[f] = lds_frag<OPC>(slot, f, lane);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f4 bv = splat(bias[BIAS_QKV + 64 + 16 * h + c]);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        f4 acc = bv;
#pragma unroll
                        for (int s = 0; s < 4; ++s) acc = OP::mfma32(X[s][ct], Bw[h * 4 + s], acc);
                        if constexpr (ASPLIT) {
                            if (ct == 0) {
#pragma unroll
                                for (int s = 0; s < 4; ++s) acc = OP::mfma32(Xlo1[s], Bw[h * 4 + s], acc);
                            }
                        }
                        Vh[ct][h] = OP::pack4(acc);
                        if constexpr (ASPLIT) Vlo[ct][h] = pack4_lo(acc, Vh[ct][h]);
                    }
                }
            }
            f4 kinit[3];
            make_kinit(kinit);
            attention_proj<OP, PERKEY, false, NCT, ASPLIT>(R, Qh, Kh, Vh, tvalid, kinit, slot + 8 * FRAG_BYTES, bias, lane, g, Qlo, Klo, Vlo);
        }

        }

        TSTAMP(4);                                                     // V + attention + proj
        // ===== MLP: 12 stages; stage ch = fc1 rows 32ch..32ch+31 (frags j*4+s) and fc2 k-step ch (frags 8+ft)
        layernorm_to_operand<OPC, NCT>(R, 1e-6f, X);
        v8 Xlo[4], HXlo = OP::zero8();                                 // ASPLIT: the low-order terms of tile 0's two MLP operands
        if constexpr (ASPLIT) layernorm_lo_operand<false>(R, 1e-6f, X, Xlo, nullptr);
        TSTAMP(5);                                                     // LN2
        // Software-pipelined over the 12 stages: at stage ch the fc1 fragment reads are issued first, then the fc2
        // MFMAs of chunk ch-1 run from registers (A2/HX carried across the barrier) and cover the LDS latency; the
        // fc2 fragment reads of chunk ch fly under its fc1 MFMAs.
        v8 A2[8], HX[3];
        auto fc1_chunk = [&](const char* slot, int ch, const v8 (&A)[8]) {
            const f4 b10 = *(const f4*)(bias + B_FC1 + 32 * ch + 4 * g), b11 = *(const f4*)(bias + B_FC1 + 32 * ch + 16 + 4 * g);
#pragma unroll
            for (int f = 0; f < 8; ++f) A2[f] = lds_frag<OPC>(slot, 8 + f, lane);   // fc2 fragments of this chunk
            // token-tile-major (round 4: -1.0 % encoder time against the row-tile-major order): a tile's two accumulators finish after its
            // eight MFMAs, so its conversion + ReLU issue under the NEXT tile's MFMAs instead of piling up behind the last one; fc2 below
            // consumes the tiles in the same order.  (Measured no better: one VALU instruction pinned behind each MFMA with
            // sched_group_barrier; the last tile's conversion deferred behind the next ring step.)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                f4 H0 = b10, H1 = b11;
#pragma unroll
                for (int s = 0; s < 4; ++s) { H0 = OP::mfma32(A[s], X[s][ct], H0); H1 = OP::mfma32(A[4 + s], X[s][ct], H1); }
                if constexpr (ASPLIT) {
                    if (ct == 0) {                                     // tile 0: + W . Xlo, and the activation as hi + lo
#pragma unroll
                        for (int s = 0; s < 4; ++s) { H0 = OP::mfma32(A[s], Xlo[s], H0); H1 = OP::mfma32(A[4 + s], Xlo[s], H1); }
                        f4 a0, a1;
                        if constexpr (ACT == 1) { a0 = gelu4(H0); a1 = gelu4(H1); }
                        else { a0 = f4{fmaxf(H0[0], 0.f), fmaxf(H0[1], 0.f), fmaxf(H0[2], 0.f), fmaxf(H0[3], 0.f)};
                               a1 = f4{fmaxf(H1[0], 0.f), fmaxf(H1[1], 0.f), fmaxf(H1[2], 0.f), fmaxf(H1[3], 0.f)}; }
                        HX[0] = OP::pack8(a0, a1);
                        const h8 hi = __builtin_bit_cast(h8, HX[0]);
                        HXlo = OP::pack8(a0 - f4{(float)hi[0], (float)hi[1], (float)hi[2], (float)hi[3]}, a1 - f4{(float)hi[4], (float)hi[5], (float)hi[6], (float)hi[7]});
                        continue;
                    }
                }
                if constexpr (ACT == 1) HX[ct] = OP::pack8(gelu4(H0), gelu4(H1));
                else HX[ct] = OP::relu_pack8(H0, H1);
            }
        };
        auto fc2_lo = [&]() {                                          // ASPLIT: tile 0's low-order activation term through fc2
            if constexpr (ASPLIT) {
#pragma unroll
                for (int ft = 0; ft < 8; ++ft) R[ft][0] = OP::mfma32(A2[ft], HXlo, R[ft][0]);
            }
        };
        {   // chunk 0 (peeled: defines A2 / HX before the steady-state loop)
            ring_step<WAVES, SB>(a.wstream, ring, tb + MLP0, wave, lane);
            TSTAMP(6);
            const char* slot = ring + ((tb + MLP0) & (RING_SLOTS - 1)) * STAGE_BYTES;
            v8 A[8];
#pragma unroll
            for (int f = 0; f < 8; ++f) A[f] = lds_frag<OPC>(slot, f, lane);
            fc1_chunk(slot, 0, A);
            TSTAMP(7);
        }
#pragma unroll 1
        for (int ch = 1; ch < 12; ++ch) {
            const int t = tb + MLP0 + ch;
#if defined(UITK_LAB) && defined(ENC_DYNPRIO)
            // Lab variant (round 5): at every MLP barrier the two waves of a SIMD (wave, wave ^ 4) take a ticket from a
            // per-SIMD LDS counter; the one that arrives SECOND -- the sibling that was behind -- runs the next stage at raised priority.
            unsigned ticket = 0;
            if constexpr (WAVES == 8) {
                unsigned* cnt = (unsigned*)(zbuf + 4096) + (wave & 3);
                if (lane == 0) ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#endif
            ring_step<WAVES, SB>(a.wstream, ring, t, wave, lane);
#if defined(UITK_LAB) && defined(ENC_DYNPRIO)
            if constexpr (WAVES == 8) {
                if (__builtin_amdgcn_readfirstlane(ticket) & 1u) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
            }
#endif
            TSTAMP(6);                                                 // ring wait MLP (sum of 12)
            const char* slot = ring + (t & (RING_SLOTS - 1)) * STAGE_BYTES;
            v8 A[8];
#pragma unroll
            for (int f = 0; f < 8; ++f) A[f] = lds_frag<OPC>(slot, f, lane);   // fc1 fragments: in flight ...
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)                                              // ... under chunk ch-1's fc2, token tile by token tile
#pragma unroll
                for (int ft = 0; ft < 8; ++ft) R[ft][ct] = OP::mfma32(A2[ft], HX[ct], R[ft][ct]);
            fc2_lo();
            fc1_chunk(slot, ch, A);
            TSTAMP(7);                                                 // MLP chunk compute (sum of 12)
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) R[ft][ct] = OP::mfma32(A2[ft], HX[ct], R[ft][ct]);
        fc2_lo();
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
            const f4 b2 = *(const f4*)(bias + B_FC2 + 16 * ft + 4 * g);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) R[ft][ct] = add4s(R[ft][ct], b2);
        }
        if (a.dbg != nullptr) {
            dbg_store(blk + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the dummy tail stages before the LDS can be re-used
    if constexpr (SB || (L6 && POOL == 1) || WAVES == 4) {
        // the epilogue's time-patch index of each column, re-derived here from v_mbcnt: alive from the prologue it was one of the values
        // the register-starved instantiations spilled and reloaded inside the block loop
        const int ce = lane_id_now() & 15;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int col = 16 * ct + ce;
            if constexpr (TOK) {
                const bool pch = col >= 1 && col <= NTOK;
                const int tk = pch ? col - 1 : 0;
                pt_[ct] = pch ? tk - GT * (tk / GT) : (col == 0 ? 0 : 99);
            } else {
                const int tk = col - NTOK * (col >= NTOK ? 1 : 0);
                pt_[ct] = tk - GT * (tk / GT);
            }
        }
    }
    if constexpr (POOL == 1) {
        // ===== pooling 'dm' (models/uit.py:405-412): for every time patch t the mean over the four frequency patches goes through
        // outputlayer (LayerNorm 1e-5 + Linear) and the sigmoid; the output is the mean over the nt time patches.
        // Two phases (round 4; was: pool -> barrier -> head -> barrier per time patch, the classifier re-fetched from L2 six times):
        //   1. while the residual stream is alive: the pooled, normalised z of ALL time patches -> LDS (the weight ring is idle now:
        //      8 KiB per time patch = 16 clip slots x 128 x (hi, lo) 16-bit);
        //   2. the stream is dead: this wave's classifier fragments go to registers ONCE, then one pass over the time patches.
        const int ntiles = (a.C + 15) / 16;
        constexpr int HEAD_PRE = 5;                       // class tiles per wave (5 x 8 waves covers C <= 640; pack.cpp enforces it)
        constexpr int ZB = 2 * 16 * DIM * 2;              // bytes of one time patch's z operands
        float rstd[3], mean3[3];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) ln_stats(R, ct, 1e-6f, mean3[ct], rstd[ct]);
        const bool lo = c < 8;
        const int row0 = blockIdx.x * WAVES * 2 + 4 * g;
        __syncthreads();                                  // every wave has read its last fragments and drained its dummy-stage DMAs: the ring is free
#pragma unroll 1
        for (int tsel = 0; tsel < a.nt; ++tsel) {
            char* zb = ring + tsel * ZB;
            f4 pooled[8];
            const float r0 = pt_[0] == tsel ? rstd[0] : 0.f, r1 = pt_[1] == tsel ? rstd[1] : 0.f, r2 = pt_[2] == tsel ? rstd[2] : 0.f;
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) {
                const f4 x0 = (R[ft][0] - splat(mean3[0])) * splat(r0), x1 = (R[ft][1] - splat(mean3[1])) * splat(r1);
                const f4 x2 = (R[ft][2] - splat(mean3[2])) * splat(r2);
                const f4 A = lo ? x0 + x1 : x0;           // clip 0 = tile 0 (all lanes) + tile 1 lanes c<8
                const f4 B = lo ? x2 : x2 + x1;           // clip 1 = tile 2 (all lanes) + tile 1 lanes c>=8
                const f4 keep = lo ? A : B, send = lo ? B : A;
                f4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = keep[r] + dpp_mov<0x140>(send[r]);
                    t = dpp_add<0x141>(t);
                    t = dpp_add<0x4E>(t);
                    t = dpp_add<0xB1>(t);
                    v[r] = t;
                }
                pooled[ft] = v;
            }
            f4 acc = splat(0.f);
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) {
                const f4 gw = *(const f4*)(a.fin + 0 * DIM + 16 * ft + 4 * g);
                const f4 gb = *(const f4*)(a.fin + 1 * DIM + 16 * ft + 4 * g);
                pooled[ft] = pooled[ft] * splat(1.0f / (float)GF) * gw + gb;
                acc = acc + pooled[ft];
            }
            const float mean = xsum(hsum(acc)) * (1.0f / DIM);
            f4 q = splat(0.f);
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) { const f4 d = pooled[ft] - splat(mean); q = q + d * d; }
            const float rs = __builtin_amdgcn_rsqf(xsum(hsum(q)) * (1.0f / DIM) + 1e-5f);
            const int slot_clip = 2 * wave + (c >= 8 ? 1 : 0);
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) {
                const f4 gw = *(const f4*)(a.fin + 2 * DIM + 16 * ft + 4 * g);
                const f4 gb = *(const f4*)(a.fin + 3 * DIM + 16 * ft + 4 * g);
                const f4 z = (pooled[ft] - splat(mean)) * splat(rs) * gw + gb;
                if ((c & 7) == 0) {
                    const v4 zh = OP::pack4(z);
                    *(v4*)(zb + (slot_clip * DIM + 16 * ft + 4 * g) * 2) = zh;
                    *(v4*)(zb + 16 * DIM * 2 + (slot_clip * DIM + 16 * ft + 4 * g) * 2) = OP::pack4(z - OP::unpack4(zh));
                }
            }
        }
        __syncthreads();
        v8 Hc[HEAD_PRE][4], Hl[HEAD_PRE][4];
        f4 pacc[HEAD_PRE], hb[HEAD_PRE];
#pragma unroll
        for (int i = 0; i < HEAD_PRE; ++i) {
            const int ot = wave + i * WAVES;
            pacc[i] = splat(0.f);
            if (ot < ntiles) {
                hb[i] = splat(a.head_b[16 * ot + c]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    Hc[i][s] = *(const v8*)(a.head_w + (size_t)(ot * 4 + s) * 2 * FRAG_BYTES + lane * 16);
                    Hl[i][s] = *(const v8*)(a.head_w + ((size_t)(ot * 4 + s) * 2 + 1) * FRAG_BYTES + lane * 16);
                }
            }
        }
        const int cz = c < 2 * WAVES ? c : 0;
#pragma unroll 1
        for (int tsel = 0; tsel < a.nt; ++tsel) {
            const char* zb = ring + tsel * ZB;
            v8 Bz[4], Bzl[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const v4 l0 = *(const v4*)(zb + (cz * DIM + 32 * s + 4 * g) * 2);
                const v4 h0 = *(const v4*)(zb + (cz * DIM + 32 * s + 16 + 4 * g) * 2);
                Bz[s] = v8{l0[0], l0[1], l0[2], l0[3], h0[0], h0[1], h0[2], h0[3]};
                const v4 l1 = *(const v4*)(zb + 16 * DIM * 2 + (cz * DIM + 32 * s + 4 * g) * 2);
                const v4 h1 = *(const v4*)(zb + 16 * DIM * 2 + (cz * DIM + 32 * s + 16 + 4 * g) * 2);
                Bzl[s] = v8{l1[0], l1[1], l1[2], l1[3], h1[0], h1[1], h1[2], h1[3]};
            }
#pragma unroll
            for (int i = 0; i < HEAD_PRE; ++i) {
                if (wave + i * WAVES < ntiles) {
                    f4 acc = hb[i];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc = OP::mfma32(Bz[s], Hl[i][s], acc);
                        acc = OP::mfma32(Bzl[s], Hc[i][s], acc);
                        acc = OP::mfma32(Bz[s], Hc[i][s], acc);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) pacc[i][r] += __builtin_amdgcn_rcpf(1.0f + __expf(-acc[r]));
                }
            }
        }
        const float inv_nt = 1.0f / (float)a.nt;
#pragma unroll
        for (int i = 0; i < HEAD_PRE; ++i) {
            const int cls = 16 * (wave + i * WAVES) + c;
            if (wave + i * WAVES < ntiles && cls < a.C) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rowg = row0 + r;
                    if (4 * g + r < 2 * WAVES && rowg < a.rows) a.probs[(int64_t)rowg * a.C + cls] = pacc[i][r] * inv_nt;
                }
            }
        }
        return;
    }
    if constexpr (POOL == 2) {
        // ===== pooling 'token' (models/uit.py:399-401): the cls row (column 0 = tile 0, lanes c == 0) of norm(x) -> outputlayer
        // -> sigmoid.  One clip per wave: WAVES clip slots per workgroup.
        const int ntiles = (a.C + 15) / 16;
        float mean0, rstd0;
        ln_stats(R, 0, 1e-6f, mean0, rstd0);
        f4 pooled[8];
        {
            f4 acc = splat(0.f);
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) {
                const f4 gw = *(const f4*)(a.fin + 0 * DIM + 16 * ft + 4 * g);
                const f4 gb = *(const f4*)(a.fin + 1 * DIM + 16 * ft + 4 * g);
                pooled[ft] = (R[ft][0] - splat(mean0)) * splat(rstd0) * gw + gb;
                acc = acc + pooled[ft];
            }
            const float mean = xsum(hsum(acc)) * (1.0f / DIM);
            f4 q = splat(0.f);
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) { const f4 d = pooled[ft] - splat(mean); q = q + d * d; }
            const float rs = __builtin_amdgcn_rsqf(xsum(hsum(q)) * (1.0f / DIM) + 1e-5f);
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) {
                const f4 gw = *(const f4*)(a.fin + 2 * DIM + 16 * ft + 4 * g);
                const f4 gb = *(const f4*)(a.fin + 3 * DIM + 16 * ft + 4 * g);
                const f4 z = (pooled[ft] - splat(mean)) * splat(rs) * gw + gb;
                if (c == 0) {
                    const v4 zh = OP::pack4(z);
                    *(v4*)(zbuf + (wave * DIM + 16 * ft + 4 * g) * 2) = zh;
                    *(v4*)(zbuf + 16 * DIM * 2 + (wave * DIM + 16 * ft + 4 * g) * 2) = OP::pack4(z - OP::unpack4(zh));
                }
            }
        }
        __syncthreads();
        const int cz = c < WAVES ? c : 0;
        v8 Bz[4], Bzl[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const v4 l0 = *(const v4*)(zbuf + (cz * DIM + 32 * s + 4 * g) * 2);
            const v4 h0 = *(const v4*)(zbuf + (cz * DIM + 32 * s + 16 + 4 * g) * 2);
            Bz[s] = v8{l0[0], l0[1], l0[2], l0[3], h0[0], h0[1], h0[2], h0[3]};
            const v4 l1 = *(const v4*)(zbuf + 16 * DIM * 2 + (cz * DIM + 32 * s + 4 * g) * 2);
            const v4 h1 = *(const v4*)(zbuf + 16 * DIM * 2 + (cz * DIM + 32 * s + 16 + 4 * g) * 2);
            Bzl[s] = v8{l1[0], l1[1], l1[2], l1[3], h1[0], h1[1], h1[2], h1[3]};
        }
        const int row0 = blockIdx.x * WAVES + 4 * g;
        for (int ot = wave; ot < ntiles; ot += WAVES) {
            const int cls = 16 * ot + c;
            f4 acc = splat(a.head_b[cls]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const v8 Hc = *(const v8*)(a.head_w + (size_t)(ot * 4 + s) * 2 * FRAG_BYTES + lane * 16);
                const v8 Hlo = *(const v8*)(a.head_w + ((size_t)(ot * 4 + s) * 2 + 1) * FRAG_BYTES + lane * 16);
                acc = OP::mfma32(Bz[s], Hlo, acc);
                acc = OP::mfma32(Bzl[s], Hc, acc);
                acc = OP::mfma32(Bz[s], Hc, acc);
            }
            if (cls < a.C) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rowg = row0 + r;
                    if (4 * g + r < WAVES && rowg < a.rows) {
                        if (a.logits) a.logits[(int64_t)rowg * a.C + cls] = acc[r];
                        a.probs[(int64_t)rowg * a.C + cls] = __builtin_amdgcn_rcpf(1.0f + __expf(-acc[r]));
                    }
                }
            }
        }
        return;
    }
    unsigned long long t_blocks_end = 0;
    if constexpr (TIMING) { t_blocks_end = stamp(); tlast = t_blocks_end; }
    // classifier weights of this wave's first output tile: fetched now, consumed after pooling
    const int ntiles = (a.C + 15) / 16;
    constexpr int HEAD_PRE = 5;                           // tiles per wave held in VGPRs (5 x 8 waves covers C <= 640)
    v8 Hpre[HEAD_PRE][4];
#pragma unroll
    for (int i = 0; i < HEAD_PRE; ++i)
        if (wave + i * WAVES < ntiles) {
#pragma unroll
            for (int s = 0; s < 4; ++s) Hpre[i][s] = *(const v8*)(a.head_w + (size_t)((wave + i * WAVES) * 4 + s) * 2 * FRAG_BYTES + lane * 16);
        }

    // ---- final norm (eps 1e-6) -> masked mean over tokens -> head LayerNorm (eps 1e-5)
    f4 pooled[8];
    {
        float mean[3], rstd[3];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            ln_stats(R, ct, 1e-6f, mean[ct], rstd[ct]);
            if (pt_[ct] >= a.nt) rstd[ct] = 0.f;          // masked token contributes 0 to the pool
        }
        const bool lo = c < 8;
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
            const f4 x0 = (R[ft][0] - splat(mean[0])) * splat(rstd[0]);
            const f4 x1 = (R[ft][1] - splat(mean[1])) * splat(rstd[1]);
            const f4 x2 = (R[ft][2] - splat(mean[2])) * splat(rstd[2]);
            const f4 A = lo ? x0 + x1 : x0;               // clip 0 = tile 0 (all lanes) + tile 1 lanes c<8
            const f4 B = lo ? x2 : x2 + x1;               // clip 1 = tile 2 (all lanes) + tile 1 lanes c>=8
            const f4 keep = lo ? A : B, send = lo ? B : A;
            f4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = keep[r] + dpp_mov<0x140>(send[r]);          // row_mirror: lane i <- lane 15-i (other half)
                t = dpp_add<0x141>(t);                                // row_half_mirror: i <- 7-i
                t = dpp_add<0x4E>(t);                                 // quad_perm [2,3,0,1]: i <- i^2
                t = dpp_add<0xB1>(t);                                 // quad_perm [1,0,3,2]: i <- i^1
                v[r] = t;
            }
            pooled[ft] = v;                               // lanes c<8: clip 0, lanes c>=8: clip 1
        }
    }
    // the residual stream is dead now: fetch the low-order classifier terms of the same tiles as well, so that no
    // head tile waits on an L2 round trip
    v8 Hlpre[HEAD_PRE][4];
#pragma unroll
    for (int i = 0; i < HEAD_PRE; ++i)
        if (wave + i * WAVES < ntiles) {
#pragma unroll
            for (int s = 0; s < 4; ++s) Hlpre[i][s] = *(const v8*)(a.head_w + ((size_t)((wave + i * WAVES) * 4 + s) * 2 + 1) * FRAG_BYTES + lane * 16);
        }
    TSTAMP(15);                                                           // classifier prefetch issue + final LN + pooling
    {
        const float inv_ntok = 1.0f / (float)(GF * a.nt);
        f4 acc = splat(0.f);
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
            const f4 gw = fin4(0, ft), gb = fin4(1, ft);
            pooled[ft] = pooled[ft] * splat(inv_ntok) * gw + gb;
            acc = acc + pooled[ft];
        }
        const float mean = xsum(hsum(acc)) * (1.0f / DIM);
        f4 q = splat(0.f);
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) { const f4 d = pooled[ft] - splat(mean); q = q + d * d; }
        const float rstd = __builtin_amdgcn_rsqf(xsum(hsum(q)) * (1.0f / DIM) + 1e-5f);
        const int slot_clip = 2 * wave + (c >= 8 ? 1 : 0);
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
            const f4 gw = fin4(2, ft), gb = fin4(3, ft);
            const f4 z = (pooled[ft] - splat(mean)) * splat(rstd) * gw + gb;
            if ((c & 7) == 0) {                           // split operand: z = hi + lo (two 16-bit terms)
                const v4 zh = OP::pack4(z);
                *(v4*)(zbuf + (slot_clip * DIM + 16 * ft + 4 * g) * 2) = zh;
                *(v4*)(zbuf + 16 * DIM * 2 + (slot_clip * DIM + 16 * ft + 4 * g) * 2) = OP::pack4(z - OP::unpack4(zh));
            }
        }
    }
    TSTAMP(16);                                                           // head LN + operand hand-over through LDS
    __syncthreads();
    TSTAMP(17);                                                           // barrier

    // ---- classifier head once per workgroup: logits^T[class][clip slot] = Whead . z^T
    {
        const int cz = c < 2 * WAVES ? c : 0;             // columns beyond the workgroup's clips: any valid data
        v8 Bz[4], Bzl[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const v4 lo = *(const v4*)(zbuf + (cz * DIM + 32 * s + 4 * g) * 2);
            const v4 hi = *(const v4*)(zbuf + (cz * DIM + 32 * s + 16 + 4 * g) * 2);
            Bz[s] = v8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const v4 llo = *(const v4*)(zbuf + 16 * DIM * 2 + (cz * DIM + 32 * s + 4 * g) * 2);
            const v4 lhi = *(const v4*)(zbuf + 16 * DIM * 2 + (cz * DIM + 32 * s + 16 + 4 * g) * 2);
            Bzl[s] = v8{llo[0], llo[1], llo[2], llo[3], lhi[0], lhi[1], lhi[2], lhi[3]};
        }
        TSTAMP(18);                                                       // z operand reads
        // logits[clip slot][class] = z . Whead^T with z as the A operand (rows = clip slots 4g+r) and the classifier
        // fragments as B (columns = classes): a store instruction then writes 64-byte runs of one clip's row.
        // logits = (Wh + Wl).(zh + zl) ~ Wh.zh + Wh.zl + Wl.zh : three MFMAs per k-step, fp32 accumulate
        const int row0 = blockIdx.x * WAVES * 2 + 4 * g;
        auto head_tile = [&](int ot, const v8 (&Hc)[4], const v8 (&Hlo)[4]) {
            const int cls = 16 * ot + c;
            f4 acc = splat(a.head_b[cls]);                  // head_b is padded to whole tiles
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc = OP::mfma32(Bz[s], Hlo[s], acc);
                acc = OP::mfma32(Bzl[s], Hc[s], acc);
                acc = OP::mfma32(Bz[s], Hc[s], acc);
            }
            if (cls < a.C) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rowg = row0 + r;
                    if (4 * g + r < 2 * WAVES && rowg < a.rows) {
                        if (a.logits) a.logits[(int64_t)rowg * a.C + cls] = acc[r];
                        a.probs[(int64_t)rowg * a.C + cls] = __builtin_amdgcn_rcpf(1.0f + __expf(-acc[r]));
                    }
                }
            }
        };
#pragma unroll
        for (int i = 0; i < HEAD_PRE; ++i)
            if (wave + i * WAVES < ntiles) head_tile(wave + i * WAVES, Hpre[i], Hlpre[i]);
        for (int ot = wave + HEAD_PRE * WAVES; ot < ntiles; ot += WAVES) {       // very wide heads: plain loop
            v8 Hc[4], Hlo[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                Hc[s] = *(const v8*)(a.head_w + (size_t)(ot * 4 + s) * 2 * FRAG_BYTES + lane * 16);
                Hlo[s] = *(const v8*)(a.head_w + ((size_t)(ot * 4 + s) * 2 + 1) * FRAG_BYTES + lane * 16);
            }
            head_tile(ot, Hc, Hlo);
        }
    }
    if constexpr (TIMING) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TSTAMP(19);                                                       // head tiles + stores
        tacc[11] = tlast - t_blocks_end;                                  // pooling + head + stores (sum of 15..19)
        if (a.timing != nullptr && lane == 0) {
#pragma unroll
            for (int i = 0; i < ENC_TIMING_SLOTS; ++i) a.timing[((size_t)blockIdx.x * WAVES + wave) * ENC_TIMING_SLOTS + i] = tacc[i];
        }
    }
}

// probs[b][c] = mean / max over crops of rowprobs[b*ncrops + cr][c]   (models/uit.py:482-488)
__global__ void crop_reduce_kernel(const float* __restrict__ rp, float* __restrict__ probs, int64_t B, int ncrops, int C,
                                   int eval_avg) {
    const int64_t n = B * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / C; const int cc = (int)(i - b * C);
        float acc = rp[(b * ncrops) * C + cc];
        for (int cr = 1; cr < ncrops; ++cr) {
            const float v = rp[(b * ncrops + cr) * C + cc];
            acc = eval_avg == 0 ? acc + v : fmaxf(acc, v);
        }
        probs[i] = eval_avg == 0 ? acc / (float)ncrops : acc;
    }
}

template <int PREC, int WAVES, bool SHORT, bool TIMING = false, int ATT = 0, int ACT = 0, int POOL = 0>
static hipError_t launch_enc(const EncoderArgs& a, hipStream_t stream) {
    // the attribute is per device: one bit per device, set once (see launch_fe)
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute((const void*)uit_encoder_kernel<PREC, WAVES, SHORT, TIMING, ATT, ACT, POOL>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, WAVES == 8 ? ENC_LDS_W8 : ENC_LDS);
        if (e != hipSuccess) return e;
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    const int per_wg = POOL == 2 ? WAVES : 2 * WAVES;     // clip rows per workgroup (pooling 'token': one clip per wave)
    const int grid = (a.rows + per_wg - 1) / per_wg;
    hipLaunchKernelGGL((uit_encoder_kernel<PREC, WAVES, SHORT, TIMING, ATT, ACT, POOL>), dim3(grid), dim3(WAVES * 64),
                       WAVES == 8 ? ENC_LDS_W8 : ENC_LDS, stream, a);
    return hipGetLastError();
}

hipError_t launch_encoder(const EncoderArgs& a, int precision, int waves, hipStream_t stream) {
#ifdef UITK_LAB
    if (a.timing != nullptr) {                            // -DUITK_LAB: build with in-kernel stamps (f16, full clips only; tools/phase_timing.py)
        if (a.nt < GT) return hipErrorInvalidValue;       // the instrumented build has no short-clip (masked padding) instantiation
        return waves == 4 ? launch_enc<0, 4, false, true>(a, stream) : launch_enc<0, 8, false, true>(a, stream);
    }
#else
    if (a.timing != nullptr) return hipErrorInvalidValue; // the product library carries no instrumented kernel
#endif
    if (a.pooling && precision == 4) {                    // pooling 'dm' / 'token' in the parity mode: BNeckAttention + ReLU only
        if (a.attention_full || a.act_gelu) return hipErrorInvalidValue;
        if (a.pooling == 2) return launch_enc<4, 8, true, false, 0, 0, 2>(a, stream);
        return a.nt < GT ? launch_enc<4, 8, true, false, 0, 0, 1>(a, stream) : launch_enc<4, 8, false, false, 0, 0, 1>(a, stream);
    }
    if (a.pooling) {                                      // pooling 'dm' / 'token' (f16 / f16x2, 8-wave workgroups), any attention / activation
        if ((precision != 0 && precision != 3 && precision != 5 && precision != 6) || a.timing != nullptr) return hipErrorInvalidValue;
        if ((precision == 5 || precision == 6) && a.pooling != 2) return hipErrorInvalidValue;
#define UITK_POOL(P, SH, PL) (a.attention_full ? (a.act_gelu ? launch_enc<P, 8, SH, false, 1, 1, PL>(a, stream) : launch_enc<P, 8, SH, false, 1, 0, PL>(a, stream)) \
                                               : (a.act_gelu ? launch_enc<P, 8, SH, false, 0, 1, PL>(a, stream) : launch_enc<P, 8, SH, false, 0, 0, PL>(a, stream)))
        if (precision == 3) {                             // split weights (round 3): the parity mode of the full-width-attention factories
            if (a.pooling == 2) return UITK_POOL(3, true, 2);
            return a.nt < GT ? UITK_POOL(3, true, 1) : UITK_POOL(3, false, 1);
        }
        if (precision == 5) return UITK_POOL(5, true, 2);  // precision f16a: split activation operands in the cls tile
        if (precision == 6) return UITK_POOL(6, true, 2);  // precision f16ax: the same on split weights
        if (a.pooling == 2) return UITK_POOL(0, true, 2);  // per-key softmax masks always (dead columns)
        return a.nt < GT ? UITK_POOL(0, true, 1) : UITK_POOL(0, false, 1);
#undef UITK_POOL
    }
    if (a.attention_full || a.act_gelu) {                 // SURVEY 8f-4 variants: f16 / bf16 / f16x2, 8-wave workgroups
        if (precision == 2 || precision == 4 || a.timing != nullptr) return hipErrorInvalidValue;
#define UITK_VARIANT(P, SH) (a.attention_full ? (a.act_gelu ? launch_enc<P, 8, SH, false, 1, 1>(a, stream) : launch_enc<P, 8, SH, false, 1, 0>(a, stream)) \
                                               : launch_enc<P, 8, SH, false, 0, 1>(a, stream))
        if (precision == 3) return a.nt < GT ? UITK_VARIANT(3, true) : UITK_VARIANT(3, false);      // split weights (round 3)
        if (a.nt < GT) return precision == 0 ? UITK_VARIANT(0, true) : UITK_VARIANT(1, true);
        return precision == 0 ? UITK_VARIANT(0, false) : UITK_VARIANT(1, false);
#undef UITK_VARIANT
    }
    if (precision == 4)                                   // f16l6: f16 + low-order fp6 weight term, 8-wave workgroups
        return a.nt < GT ? launch_enc<4, 8, true>(a, stream) : launch_enc<4, 8, false>(a, stream);
    if (precision == 3)                                   // f16x2: split weights (hi + lo), 8-wave workgroups
        return a.nt < GT ? launch_enc<3, 8, true>(a, stream) : launch_enc<3, 8, false>(a, stream);
    if (precision == 2) {                                 // fp8 weights / activations on the MX-scaled 16x16x128 MFMA
        if (waves == 4) return a.nt < GT ? launch_enc<2, 4, true>(a, stream) : launch_enc<2, 4, false>(a, stream);
        return a.nt < GT ? launch_enc<2, 8, true>(a, stream) : launch_enc<2, 8, false>(a, stream);
    }
    if (a.nt < GT) {                                      // short clips: masked padding tokens
        if (waves == 4) return precision == 0 ? launch_enc<0, 4, true>(a, stream) : launch_enc<1, 4, true>(a, stream);
        return precision == 0 ? launch_enc<0, 8, true>(a, stream) : launch_enc<1, 8, true>(a, stream);
    }
    if (waves == 4) return precision == 0 ? launch_enc<0, 4, false>(a, stream) : launch_enc<1, 4, false>(a, stream);
    return precision == 0 ? launch_enc<0, 8, false>(a, stream) : launch_enc<1, 8, false>(a, stream);
}

hipError_t launch_crop_reduce(const float* rowprobs, float* probs, int64_t B, int ncrops, int C, int eval_avg,
                              hipStream_t stream) {
    int grid = (int)((B * C + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(crop_reduce_kernel, dim3(grid), dim3(256), 0, stream, rowprobs, probs, B, ncrops, C, eval_avg);
    return hipGetLastError();
}

}  // namespace uitk
