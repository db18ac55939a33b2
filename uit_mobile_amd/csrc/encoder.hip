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
};

constexpr int ENC_WAVES = 8;
constexpr int RING_SLOTS = 4;           // >= PREFETCH + 2 (see ring_step)
constexpr int PREFETCH = 2;             // stages in flight ahead of the one being consumed
constexpr int LDS_RING = RING_SLOTS * STAGE_BYTES;            // 64 KiB
constexpr int LDS_BIAS = 2 * BIAS_FLOATS * 4;                 // 8 KiB (double-buffered per block)
constexpr int LDS_ZBUF = 2 * ENC_WAVES * DIM * 2;             // 4 KiB: 16 pooled vectors, 16-bit
constexpr int ENC_LDS = LDS_RING + LDS_BIAS + LDS_ZBUF;

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---- LDS-DMA weight ring --------------------------------------------------------------------------------
// Stage s lives in slot s % RING_SLOTS.  At step t every wave (1) issues its share of stage t+PREFETCH,
// (2) waits (counted vmcnt) until its share of stage t has landed, (3) s_barrier => the whole stage is visible.
// WAR safety: the slot overwritten at step t held stage t+PREFETCH-RING_SLOTS <= t-2, and a wave can only be
// at step t after all waves passed barrier(t-1), i.e. finished reading stage t-2.
template <int WAVES>
__device__ __forceinline__ void ring_issue(const char* wstream, char* ring, int stage, int wave, int lane) {
    constexpr int PER_WAVE = STAGE_FRAGS / WAVES;
    const char* src = wstream + (size_t)stage * STAGE_BYTES;
    char* dst = ring + (stage & (RING_SLOTS - 1)) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int piece = wave * PER_WAVE + i;
        __builtin_amdgcn_global_load_lds(GPTR(src + piece * FRAG_BYTES + lane * 16), LPTR(dst + piece * FRAG_BYTES), 16, 0, 0);
    }
}

template <int WAVES>
__device__ __forceinline__ void ring_step(const char* wstream, char* ring, int t, int total, int wave, int lane) {
    constexpr int PER_WAVE = STAGE_FRAGS / WAVES;
    static_assert(PREFETCH == 2, "vmcnt immediates below assume PREFETCH == 2");
    if (t + PREFETCH < total) ring_issue<WAVES>(wstream, ring, t + PREFETCH, wave, lane);
    const int younger = total - 1 - t;          // stages issued after stage t (capped at PREFETCH)
    if (younger >= 2) {
        if constexpr (PER_WAVE == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (younger == 1) {
        if constexpr (PER_WAVE == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int WAVES>
__device__ __forceinline__ void bias_issue(const float* bias, char* breg, int blk, int wave, int lane) {
    // 4 KiB per block = WAVES waves x (512/WAVES*... ) -> each wave copies 4096/WAVES bytes with dword LDS-DMA
    constexpr int PER_WAVE_BYTES = BIAS_FLOATS * 4 / WAVES;      // 512 (8 waves)
    constexpr int N = PER_WAVE_BYTES / 256;                      // dword x 64 lanes = 256 B per instruction
    const char* src = (const char*)(bias + (size_t)blk * BIAS_FLOATS) + wave * PER_WAVE_BYTES;
    char* dst = breg + (blk & 1) * BIAS_FLOATS * 4 + wave * PER_WAVE_BYTES;
#pragma unroll
    for (int i = 0; i < N; ++i)
        __builtin_amdgcn_global_load_lds(GPTR(src + i * 256 + lane * 4), LPTR(dst + i * 256), 4, 0, 0);
}

template <int PREC>
__device__ __forceinline__ typename Op<PREC>::v8 lds_frag(const char* slot, int frag, int lane) {
    return *(const typename Op<PREC>::v8*)(slot + frag * FRAG_BYTES + lane * 16);
}

__device__ __forceinline__ f4 splat(float x) { return f4{x, x, x, x}; }
__device__ __forceinline__ float hsum(f4 x) { return (x[0] + x[1]) + (x[2] + x[3]); }
__device__ __forceinline__ float xsum(float v) {   // sum over the 4 lane groups g (lanes c, c+16, c+32, c+48)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float xmax(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

// LayerNorm statistics + normalisation of the residual stream (affine folded into the next linear by pack.cpp).
// R[ft][ct] holds features 16ft+4g+r of token column 16ct+c.
template <int PREC>
__device__ __forceinline__ void layernorm_to_operand(const f4 (&R)[8][3], float eps, typename Op<PREC>::v8 (&X)[4][3]) {
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
        float s = 0.f;
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) s += hsum(R[ft][ct]);
        const float mean = xsum(s) * (1.0f / DIM);
        float v = 0.f;
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
            f4 d = R[ft][ct] - splat(mean);
            v += hsum(d * d);
        }
        const float rstd = __frsqrt_rn(xsum(v) * (1.0f / DIM) + eps);
#pragma unroll
        for (int s4i = 0; s4i < 4; ++s4i) {
            f4 lo = (R[2 * s4i][ct] - splat(mean)) * splat(rstd);
            f4 hi = (R[2 * s4i + 1][ct] - splat(mean)) * splat(rstd);
            X[s4i][ct] = Op<PREC>::pack8(lo, hi);
        }
    }
}

template <int PREC, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void uit_encoder_kernel(EncoderArgs a) {
    using OP = Op<PREC>;
    using v8 = typename OP::v8;
    using v4 = typename OP::v4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    char* breg = smem + LDS_RING;
    char* zbuf = smem + LDS_RING + LDS_BIAS;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int total = PE_STAGES + a.depth * BLOCK_STAGES;

    // ---- kick the weight stream
    if (a.depth > 0) bias_issue<WAVES>(a.bias, breg, 0, wave, lane);
    ring_issue<WAVES>(a.wstream, ring, 0, wave, lane);
    ring_issue<WAVES>(a.wstream, ring, 1, wave, lane);

    // ---- rows (clip-crops) of this wave, token maps
    const int r0 = (blockIdx.x * WAVES + wave) * 2;
    int64_t mel_base[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        int row = r0 + k; if (row > a.rows - 1) row = a.rows - 1;
        const int b = row / a.ncrops, cr = row - b * a.ncrops;
        int start = 0;
        if (a.ncrops > 1) { start = cr * a.tl; if (start > a.nF - a.tl) start = a.nF - a.tl; }
        mel_base[k] = ((int64_t)b * a.nF + start) * NMEL;
    }
    int clip_[3], tk_[3], pf_[3], pt_[3];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
        const int col = 16 * ct + c;
        clip_[ct] = col >= NTOK ? 1 : 0;
        tk_[ct] = col - NTOK * clip_[ct];
        pf_[ct] = tk_[ct] / GT;
        pt_[ct] = tk_[ct] - GT * pf_[ct];
    }
    float amx = -INFINITY;
    for (int i = 0; i < AMAX_SLOTS; ++i) amx = fmaxf(amx, a.amax[i]);     // NaN (untouched) slots are ignored
    const float floor_db = amx - a.top_db;

    // ---- patch-embed B operand: clamp + init_bn + convert.  k-slot (s,g,j) <-> (ti = 2s + (g>>1), fi = 8(g&1) + j)
    v8 XB[8][3];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
        const int f0 = 16 * pf_[ct] + 8 * (g & 1);
        const f4 sc0 = *(const f4*)(a.bn_sc + f0), sc1 = *(const f4*)(a.bn_sc + f0 + 4);
        const f4 sh0 = *(const f4*)(a.bn_sh + f0), sh1 = *(const f4*)(a.bn_sh + f0 + 4);
        const int64_t base = clip_[ct] ? mel_base[1] : mel_base[0];
        const int fr_limit = a.nF - 1 - (int)((base / NMEL) % a.nF);      // last valid frame offset from crop start
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            int t = 16 * pt_[ct] + 2 * s + (g >> 1);
            if (t > fr_limit) t = fr_limit;                               // masked tokens of short clips: stay in bounds
            const float* src = a.mel + base + (int64_t)t * NMEL + f0;
            f4 x0 = *(const f4*)src, x1 = *(const f4*)(src + 4);
            x0 = __builtin_elementwise_max(x0, splat(floor_db)) * sc0 + sh0;
            x1 = __builtin_elementwise_max(x1, splat(floor_db)) * sc1 + sh1;
            XB[s][ct] = OP::pack8(x0, x1);
        }
    }
    // ---- residual stream, initialised with conv bias + pos-embeds
    f4 R[8][3];
#pragma unroll
    for (int ft = 0; ft < 8; ++ft)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) R[ft][ct] = *(const f4*)(a.pe_bias + tk_[ct] * DIM + 16 * ft + 4 * g);

    // ---- patch embed: 4 stages, stage i holds fragments (ft = 2i + f/8, s = f%8)
#pragma unroll
    for (int i = 0; i < PE_STAGES; ++i) {
        ring_step<WAVES>(a.wstream, ring, i, total, wave, lane);
        const char* slot = ring + (i & (RING_SLOTS - 1)) * STAGE_BYTES;
#pragma unroll
        for (int f = 0; f < STAGE_FRAGS; ++f) {
            const v8 A = lds_frag<PREC>(slot, f, lane);
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) R[2 * i + f / 8][ct] = OP::mfma32(A, XB[f % 8][ct], R[2 * i + f / 8][ct]);
        }
    }

    auto dbg_store = [&](int stage) {
        if (a.dbg == nullptr) return;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            const int row = r0 + clip_[ct];
            if (row < a.rows) {
                float* dst = a.dbg + (((int64_t)row * (a.depth + 1) + stage) * NTOK + tk_[ct]) * DIM + 4 * g;
#pragma unroll
                for (int ft = 0; ft < 8; ++ft) *(f4*)(dst + 16 * ft) = R[ft][ct];
            }
        }
    };
    dbg_store(0);

    // ---- attention validity mask (constant over blocks): tile list (jt,it)
    constexpr int TJ[7] = {0, 1, 0, 1, 2, 1, 2};
    constexpr int TI[7] = {0, 0, 1, 1, 1, 2, 2};
    unsigned vmask = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * TJ[i] + 4 * g + r, q = 16 * TI[i] + c;
            const int jc = j >= NTOK ? 1 : 0, qc = q >= NTOK ? 1 : 0;
            const int jt = (j - NTOK * jc) % GT;
            if (jc == qc && jt < a.nt) vmask |= 1u << (4 * i + r);
        }

    // ---- transformer blocks
#pragma unroll 1
    for (int blk = 0; blk < a.depth; ++blk) {
        const int tb = PE_STAGES + blk * BLOCK_STAGES;
        const float* bias = (const float*)(breg + (blk & 1) * BIAS_FLOATS * 4);
        v8 X[4][3];
        layernorm_to_operand<PREC>(R, 1e-6f, X);

        // ===== stage QK: fragments (ft = f/4 in {q h0, q h1, k h0, k h1}, s = f%4)
        ring_step<WAVES>(a.wstream, ring, tb, total, wave, lane);
        if (blk + 1 < a.depth) bias_issue<WAVES>(a.bias, breg, blk + 1, wave, lane);
        v4 Qh[2][3], Kh[2][3];
        {
            const char* slot = ring + (tb & (RING_SLOTS - 1)) * STAGE_BYTES;
            f4 QK[4][3];
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                const f4 bq = *(const f4*)(bias + BIAS_QKV + 16 * ft + 4 * g);
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) QK[ft][ct] = bq;
            }
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                const v8 A = lds_frag<PREC>(slot, f, lane);
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) QK[f / 4][ct] = OP::mfma32(A, X[f % 4][ct], QK[f / 4][ct]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) { Qh[h][ct] = OP::pack4(QK[h][ct]); Kh[h][ct] = OP::pack4(QK[2 + h][ct]); }
        }

        // ===== stage VP: fragments 0..7 = V (h = f/4, s = f%4) used as the B operand; 8..15 = proj (ft = f-8)
        ring_step<WAVES>(a.wstream, ring, tb + 1, total, wave, lane);
        {
            const char* slot = ring + ((tb + 1) & (RING_SLOTS - 1)) * STAGE_BYTES;
            v4 Vh[3][2];                       // V[token 16ct+4g+r][d = c] -> A operand of P.V (rows d, k = token)
            {
                f4 V[3][2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float bv = bias[BIAS_QKV + 64 + 16 * h + c];
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) V[ct][h] = splat(bv);
                }
#pragma unroll
                for (int f = 0; f < 8; ++f) {
                    const v8 Bw = lds_frag<PREC>(slot, f, lane);
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) V[ct][f / 4] = OP::mfma32(X[f % 4][ct], Bw, V[ct][f / 4]);
                }
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                    for (int h = 0; h < 2; ++h) Vh[ct][h] = OP::pack4(V[ct][h]);
            }
            f4 O[2][3];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // S^T[j = 16jt+4g+r][i = 16it+c] = sum_d K[j][d] Q[i][d]   (scale folded into Wq)
                f4 S[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    S[i] = OP::mfma16(Kh[h][TJ[i]], Qh[h][TI[i]], splat(0.f));
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (!((vmask >> (4 * i + r)) & 1u)) S[i][r] = -INFINITY;
                }
                // softmax over j per query column: tiles {0,1} -> it 0, {2,3,4} -> it 1, {5,6} -> it 2
                float mx[3], inv[3];
                mx[0] = fmaxf(fmaxf(fmaxf(S[0][0], S[0][1]), fmaxf(S[0][2], S[0][3])), fmaxf(fmaxf(S[1][0], S[1][1]), fmaxf(S[1][2], S[1][3])));
                mx[1] = fmaxf(fmaxf(fmaxf(fmaxf(S[2][0], S[2][1]), fmaxf(S[2][2], S[2][3])), fmaxf(fmaxf(S[3][0], S[3][1]), fmaxf(S[3][2], S[3][3]))),
                              fmaxf(fmaxf(S[4][0], S[4][1]), fmaxf(S[4][2], S[4][3])));
                mx[2] = fmaxf(fmaxf(fmaxf(S[5][0], S[5][1]), fmaxf(S[5][2], S[5][3])), fmaxf(fmaxf(S[6][0], S[6][1]), fmaxf(S[6][2], S[6][3])));
#pragma unroll
                for (int it = 0; it < 3; ++it) mx[it] = xmax(mx[it]);
                v4 Pm[7];
                float sm[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    f4 p;
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[r] = __expf(S[i][r] - mx[TI[i]]);
                    sm[TI[i]] += hsum(p);
                    Pm[i] = OP::pack4(p);
                }
#pragma unroll
                for (int it = 0; it < 3; ++it) inv[it] = __frcp_rn(xsum(sm[it]));
                // O^T[d = 4g+r][i] = sum_j V[j][d] P^T[j][i]
#pragma unroll
                for (int it = 0; it < 3; ++it) O[h][it] = splat(0.f);
#pragma unroll
                for (int i = 0; i < 7; ++i) O[h][TI[i]] = OP::mfma16(Vh[TJ[i]][h], Pm[i], O[h][TI[i]]);
#pragma unroll
                for (int it = 0; it < 3; ++it) O[h][it] = O[h][it] * splat(inv[it]);
            }
            // proj: K = 32 = (head, d); slots j<4 head 0, j>=4 head 1 (same universal k-order as pack_A)
            v8 OX[3];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) OX[ct] = OP::pack8(O[0][ct], O[1][ct]);
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) {
                const v8 A = lds_frag<PREC>(slot, 8 + ft, lane);
                const f4 bp = *(const f4*)(bias + BIAS_PROJ + 16 * ft + 4 * g);
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) R[ft][ct] = OP::mfma32(A, OX[ct], R[ft][ct] + bp);
            }
        }

        // ===== MLP: 12 stages; stage ch = fc1 rows 32ch..32ch+31 (frags j*4+s) and fc2 k-step ch (frags 8+ft)
        layernorm_to_operand<PREC>(R, 1e-6f, X);
#pragma unroll 1
        for (int ch = 0; ch < 12; ++ch) {
            const int t = tb + 2 + ch;
            ring_step<WAVES>(a.wstream, ring, t, total, wave, lane);
            const char* slot = ring + (t & (RING_SLOTS - 1)) * STAGE_BYTES;
            f4 Hh[2][3];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f4 b1 = *(const f4*)(bias + BIAS_FC1 + 32 * ch + 16 * j + 4 * g);
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) Hh[j][ct] = b1;
            }
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                const v8 A = lds_frag<PREC>(slot, f, lane);
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) Hh[f / 4][ct] = OP::mfma32(A, X[f % 4][ct], Hh[f / 4][ct]);
            }
            v8 HX[3];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)
                HX[ct] = OP::pack8(__builtin_elementwise_max(Hh[0][ct], splat(0.f)), __builtin_elementwise_max(Hh[1][ct], splat(0.f)));
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) {
                const v8 A = lds_frag<PREC>(slot, 8 + ft, lane);
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) R[ft][ct] = OP::mfma32(A, HX[ct], R[ft][ct]);
            }
        }
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
            const f4 b2 = *(const f4*)(bias + BIAS_FC2 + 16 * ft + 4 * g);
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) R[ft][ct] = R[ft][ct] + b2;
        }
        if (a.dbg != nullptr) {
            dbg_store(blk + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }

    // ---- final norm (eps 1e-6) -> masked mean over tokens -> head LayerNorm (eps 1e-5)
    f4 pooled[8];
    {
        float mean[3], rstd[3];
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            float s = 0.f;
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) s += hsum(R[ft][ct]);
            mean[ct] = xsum(s) * (1.0f / DIM);
            float v = 0.f;
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) { f4 d = R[ft][ct] - splat(mean[ct]); v += hsum(d * d); }
            rstd[ct] = __frsqrt_rn(xsum(v) * (1.0f / DIM) + 1e-6f);
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
                float t = keep[r] + __shfl_xor(send[r], 8);
                t += __shfl_xor(t, 4);
                t += __shfl_xor(t, 2);
                t += __shfl_xor(t, 1);
                v[r] = t;
            }
            pooled[ft] = v;                               // lanes c<8: clip 0, lanes c>=8: clip 1
        }
    }
    {
        const float inv_ntok = 1.0f / (float)(GF * a.nt);
        float s = 0.f;
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
            const f4 gw = *(const f4*)(a.fin + 0 * DIM + 16 * ft + 4 * g);
            const f4 gb = *(const f4*)(a.fin + 1 * DIM + 16 * ft + 4 * g);
            pooled[ft] = pooled[ft] * splat(inv_ntok) * gw + gb;
            s += hsum(pooled[ft]);
        }
        const float mean = xsum(s) * (1.0f / DIM);
        float v = 0.f;
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) { f4 d = pooled[ft] - splat(mean); v += hsum(d * d); }
        const float rstd = __frsqrt_rn(xsum(v) * (1.0f / DIM) + 1e-5f);
        const int slot_clip = 2 * wave + (c >= 8 ? 1 : 0);
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
            const f4 gw = *(const f4*)(a.fin + 2 * DIM + 16 * ft + 4 * g);
            const f4 gb = *(const f4*)(a.fin + 3 * DIM + 16 * ft + 4 * g);
            const f4 z = (pooled[ft] - splat(mean)) * splat(rstd) * gw + gb;
            if ((c & 7) == 0) *(v4*)(zbuf + (slot_clip * DIM + 16 * ft + 4 * g) * 2) = OP::pack4(z);
        }
    }
    __syncthreads();

    // ---- classifier head once per workgroup: logits^T[class][clip slot] = Whead . z^T
    {
        v8 Bz[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const v4 lo = *(const v4*)(zbuf + (c * DIM + 32 * s + 4 * g) * 2);
            const v4 hi = *(const v4*)(zbuf + (c * DIM + 32 * s + 16 + 4 * g) * 2);
            Bz[s] = v8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        const int ntiles = (a.C + 15) / 16;
        const int rowg = blockIdx.x * WAVES * 2 + c;
        for (int ot = wave; ot < ntiles; ot += WAVES) {
            f4 acc = *(const f4*)(a.head_b + 16 * ot + 4 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const v8 A = *(const v8*)(a.head_w + (size_t)(ot * 4 + s) * FRAG_BYTES + lane * 16);
                acc = OP::mfma32(A, Bz[s], acc);
            }
            if (c < 2 * WAVES && rowg < a.rows) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cls = 16 * ot + 4 * g + r;
                    if (cls < a.C) {
                        if (a.logits) a.logits[(int64_t)rowg * a.C + cls] = acc[r];
                        a.probs[(int64_t)rowg * a.C + cls] = __frcp_rn(1.0f + __expf(-acc[r]));
                    }
                }
            }
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

hipError_t launch_encoder(const EncoderArgs& a, int precision, hipStream_t stream) {
    const int grid = (a.rows + 2 * ENC_WAVES - 1) / (2 * ENC_WAVES);
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)uit_encoder_kernel<0, ENC_WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, ENC_LDS);
        hipFuncSetAttribute((const void*)uit_encoder_kernel<1, ENC_WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, ENC_LDS);
        attr_done = true;
    }
    if (precision == 0)
        hipLaunchKernelGGL((uit_encoder_kernel<0, ENC_WAVES>), dim3(grid), dim3(ENC_WAVES * 64), ENC_LDS, stream, a);
    else
        hipLaunchKernelGGL((uit_encoder_kernel<1, ENC_WAVES>), dim3(grid), dim3(ENC_WAVES * 64), ENC_LDS, stream, a);
    return hipGetLastError();
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
