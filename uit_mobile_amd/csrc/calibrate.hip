// calibrate.hip -- the plain fp32 forward of the folded network.  PACK TIME: operand statistics for the calibrated rounding of pack.cpp
// (round_gptq) and the reference of the held-out check.  Round 5, late: the same kernels + a head kernel are also
// precision 'f32' (UITK_PREC_F32, launch_naive_forward below): the LAST rung of the parity ladder -- fp32 arithmetic on the vector
// ALUs, one simple kernel per operation, ~20x slower than the MFMA encoder (542 k clips/s for UiT-XS) and right for every checkpoint.
//
// uitk_calibrate (api.cpp) runs a calibration batch through the product front-end kernel and then through the plain fp32 forward
// below -- the folded network of pack.h (`Folded`: exactly the numbers pack.cpp rounds), one simple kernel per operation -- and
// accumulates, for every single-weight linear of the path, the uncentred second moment E[x x^T] of its input operand:
//     patch embed  <- the clamped log-mel patch operands                     (models/uit.py:63-66)
//     qkv          <- LayerNorm-1 output (its affine is folded into qkv)     (:223, 103 / 139)
//     proj         <- attention output                                       (:119-120 / :175-176)
//     fc1          <- LayerNorm-2 output                                     (:235, 192)
//     fc2          <- ReLU / GELU output                                     (:193-194)
// Nothing here is on the inference path and nothing here produces an output of the model: the statistics only decide, at pack
// time, in which direction each fp16 weight is rounded.  The kernels are written for clarity, not speed (a calibration pass over
// 32 clips of UiT-XS is ~2 GFLOP).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "uitk_dev.h"
#include "naive_forward.h"

namespace uitk {

constexpr int TS = 16;

__device__ __forceinline__ float op16_to_f32(unsigned short u, int bf16) {
    if (bf16) return __uint_as_float((unsigned)u << 16);
    return (float)__builtin_bit_cast(_Float16, u);
}
__device__ __forceinline__ float round_op16(float v, int bf16) {          // the value the MFMA operand holds
    if (bf16) return (float)(__bf16)v;
    return (float)(_Float16)v;
}

// X0[(row * ntok + tok)][k] = the patch operand the encoder's patch embed multiplies: init_bn(dB) as stored by the front-end
// ([rows][24][fi 16][ti 16], 16-bit), clamped at the batch's top_db floor (encoder.hip prologue); the cls row of pooling 'token'
// (tok 0 when ntok == 25) carries no patch: zero.
__global__ void cal_patch_kernel(const unsigned short* __restrict__ mel, const float* __restrict__ amax, float top_db,
                                 const float* __restrict__ bn_sc, const float* __restrict__ bn_sh, int bf16, int rows, int ntok,
                                 float* __restrict__ X0) {
    float amx = amax[0];
    for (int i = 1; i < AMAX_SLOTS; ++i) amx = fmaxf(amx, amax[i]);
    const float floor_db = amx - top_db;
    const int64_t n = (int64_t)rows * ntok * 256;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i & 255);
        const int64_t rt = i >> 8;
        const int tok = (int)(rt % ntok), row = (int)(rt / ntok);
        const int ptok = ntok == NTOK ? tok : tok - 1;                     // patch index; -1 = the cls token
        float v = 0.f;
        if (ptok >= 0) {
            const int melbin = 16 * (ptok / GT) + (k >> 4);
            const float sc = bn_sc[melbin], th = round_op16(floor_db * sc + bn_sh[melbin], bf16);
            const float y = op16_to_f32(mel[((int64_t)row * NTOK + ptok) * 256 + k], bf16);
            v = sc >= 0.f ? fmaxf(y, th) : fminf(y, th);
        }
        X0[i] = v;
    }
}

// The same operand from the front-end's fp32 dB frames (precision 'f32'): crop logic of models/uit.py:468-481 (crops start every
// target_length frames; the last one is the last target_length frames), top_db clamp, init_bn -- nothing rounded to 16 bits.
__global__ void naive_patch_from_frames_kernel(NaiveFramesArgs fa, const float* __restrict__ amax, float top_db, const float* __restrict__ bn_sc,
                                               const float* __restrict__ bn_sh, int rows, int ntok, int nt, float* __restrict__ X0) {
    float amx = amax[0];
    for (int i = 1; i < AMAX_SLOTS; ++i) amx = fmaxf(amx, amax[i]);
    const float floor_db = amx - top_db;
    const int64_t n = (int64_t)rows * ntok * 256;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i & 255);
        const int64_t rt = i >> 8;
        const int tok = (int)(rt % ntok);
        const int64_t rowg = fa.row0 + rt / ntok;
        const int ptok = ntok == NTOK ? tok : tok - 1;
        float v = 0.f;
        if (ptok >= 0 && ptok % GT < nt) {
            const int64_t b = rowg / fa.ncrops;
            const int crop = (int)(rowg % fa.ncrops);
            const int melbin = 16 * (ptok / GT) + (k >> 4), t = 16 * (ptok % GT) + (k & 15);
            const int f = fa.ncrops == 1 ? t : (crop < fa.ncrops - 1 ? crop * fa.tl + t : fa.nF - fa.tl + t);
            if (f >= 0 && f < fa.nF) v = fmaxf(fa.frames[(b * fa.nF + f) * NMEL + melbin], floor_db) * bn_sc[melbin] + bn_sh[melbin];
        }
        X0[i] = v;
    }
}

// R[m][c] = table[token row of m][c]: conv bias + positional embeddings (rows 0..23), cls token (row 24)
__global__ void cal_init_stream_kernel(const float* __restrict__ table, int M, int ntok, float* __restrict__ R) {
    const int64_t n = (int64_t)M * DIM;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i & (DIM - 1));
        const int tok = (int)((i >> 7) % ntok);
        const int trow = ntok == NTOK ? tok : (tok == 0 ? NTOK : tok - 1);
        R[i] = table[trow * DIM + c];
    }
}

constexpr int LT = 64;
// Y[m][o] = (accumulate ? Y[m][o] : 0) + act(bias[o] + sum_k X[m][k] W[o][k]);  K multiple of 16.  act: 0 none, 1 ReLU, 2 exact GELU.
// The ONE linear of the plain fp32 forward, for every token count (round 6: a 16 x 16-tile scalar kernel used to take M < 2048, so the
// 'f32' rung and the calibration statistics changed bits with the batch size -- ADVICE r5; a batch must not change its arithmetic
// with its size): 64 x 64 outputs per workgroup on the matrix pipe's EXACT-fp32 mode (v_mfma_f32_16x16x4_f32: fp32 operands, fp32
// accumulate -- what the front-end's mel stage uses in its 4x4x1 form; precision 'f32': 185 k clips/s with a scalar tile -> 387 k with
// plain register tiling -> 415 k here; 542 k with the attention kernel below).  Wave w owns token rows 16w .. 16w+15 of the tile and
// all four 16-wide output column tiles; per 16-deep LDS stage four k-steps of 4.  Operand maps: A lane l = X[m = l % 16][k = l / 16],
// B lane l = W[o = l % 16][k = l / 16], D lane l, register r = Y[m = 4 (l / 16) + r][o = l % 16].  Sums in the MFMA's order.
typedef float f4m __attribute__((ext_vector_type(4)));
// (Round 6: the global loads of a stage are one 16-byte load per thread and operand, issued a whole stage ahead into registers, and the
// LDS tiles ping-pong -- one __syncthreads per 16-deep stage instead of two, the memory round trip under the previous stage's MFMAs.
// The same products summed in the same order: bit-identical with the kernel as it was.)
__global__ void __launch_bounds__(256) naive_linear_mfma_kernel(const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ bias,
                                                                float* __restrict__ Y, int M, int K, int O, int act, int accumulate) {
    __shared__ float xs[2][TS][LT + 4], ws[2][TS][LT + 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int m0 = blockIdx.y * LT, o0 = blockIdx.x * LT;
    const int lrow = tid >> 2, lk = 4 * (tid & 3);                        // this thread's row of the tile and its four k of a stage
    const bool xok = m0 + lrow < M, wok = o0 + lrow < O;
    const float* xp = X + (int64_t)(m0 + lrow) * K + lk;
    const float* wp = W + (int64_t)(o0 + lrow) * K + lk;
    const f4m zero4 = {0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int k0, f4m& xv, f4m& wv) {
        xv = xok ? *(const f4m*)(xp + k0) : zero4;
        wv = wok ? *(const f4m*)(wp + k0) : zero4;
    };
    auto stash = [&](int buf, const f4m& xv, const f4m& wv) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { xs[buf][lk + j][lrow] = xv[j]; ws[buf][lk + j][lrow] = wv[j]; }
    };
    f4m acc[4] = {};
    f4m xv, wv;
    fetch(0, xv, wv);
    stash(0, xv, wv);
    __syncthreads();
    const int nst = K / TS;
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        if (st + 1 < nst) fetch((st + 1) * TS, xv, wv);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float a = xs[buf][4 * ks + lq][16 * wave + lr];
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ws[buf][4 * ks + lq][16 * ot + lr], acc[ot], 0, 0, 0);
        }
        if (st + 1 < nst) stash(buf ^ 1, xv, wv);
        __syncthreads();
    }
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) {
        const int o = o0 + 16 * ot + lr;
        if (o >= O) continue;
        const float bo = bias ? bias[o] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * wave + 4 * lq + r;
            if (m >= M) continue;
            float v = acc[ot][r] + bo;
            if (act == 1) v = fmaxf(v, 0.f);
            else if (act == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
            float* y = Y + (int64_t)m * O + o;
            *y = accumulate ? *y + v : v;
        }
    }
}

// The same linear on 128 x 128 outputs per workgroup, for WIDE layers only (K >= 512 and O >= 256 at M >= 8192: the general engine on
// models several hundred wide, e.g. the reference's default UITBase, 768 / 3072): twice the FLOP per byte moved from L2 into LDS.  On the
// 128-wide layers of the published geometry it LOSES (measured: `f32` rung 826 k -> 719 k clips/s -- three workgroups per CU instead of
// eight, a padded 96-wide qkv, tails), so they never take it.  Wave w owns rows 32 w .. 32 w + 31 (two row tiles) and all eight 16-wide
// column tiles.  LDS tiles [row][k] with a row's 16 k of a stage at position 4 (k % 4) + k / 4 (pitch 20 floats): a lane finds its
// operands of all four k-steps, k = 4 ks + lq, in one 16-byte read; conflict-free.  Every output is the same sum of the same products in
// the same order as in the 64 x 64 kernel: BIT-IDENTICAL (which tile runs never changes a result; tests/test_gpu_general.py).
constexpr int BT = 128;
__global__ void __launch_bounds__(256) naive_linear_mfma_big_kernel(const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ bias,
                                                                    float* __restrict__ Y, int M, int K, int O, int act, int accumulate) {
    constexpr int KP = TS + 4;
    __shared__ __attribute__((aligned(16))) float xs[2][BT][KP], ws[2][BT][KP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int m0 = blockIdx.y * BT, o0 = blockIdx.x * BT;
    const int lrow = tid >> 2, c = tid & 3;                                // this thread's two rows of a tile (lrow, lrow + 64) and its four k of a stage
    const f4m zero4 = {0.f, 0.f, 0.f, 0.f};
    f4m xv[2], wv[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = lrow + 64 * r;
            xv[r] = m0 + row < M ? *(const f4m*)(X + (int64_t)(m0 + row) * K + k0 + 4 * c) : zero4;
            wv[r] = o0 + row < O ? *(const f4m*)(W + (int64_t)(o0 + row) * K + k0 + 4 * c) : zero4;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) { xs[buf][lrow + 64 * r][4 * j + c] = xv[r][j]; ws[buf][lrow + 64 * r][4 * j + c] = wv[r][j]; }
    };
    f4m acc[2][8] = {};
    fetch(0);
    stash(0);
    __syncthreads();
    const int nst = K / TS;
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        if (st + 1 < nst) fetch((st + 1) * TS);
        f4m a[2], b[8];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) a[rt] = *(const f4m*)&xs[buf][32 * wave + 16 * rt + lr][4 * lq];
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) b[ot] = *(const f4m*)&ws[buf][16 * ot + lr][4 * lq];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ot = 0; ot < 8; ++ot) acc[rt][ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][ks], b[ot][ks], acc[rt][ot], 0, 0, 0);
        if (st + 1 < nst) stash(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) {
        const int o = o0 + 16 * ot + lr;
        if (o >= O) continue;
        const float bo = bias ? bias[o] : 0.f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 32 * wave + 16 * rt + 4 * lq + r;
                if (m >= M) continue;
                float v = acc[rt][ot][r] + bo;
                if (act == 1) v = fmaxf(v, 0.f);
                else if (act == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                float* y = Y + (int64_t)m * O + o;
                *y = accumulate ? *y + v : v;
            }
    }
}

// Xn[m][:] = (R[m][:] - mean) / sqrt(var + eps): one wave per token, two elements per lane (the affine lives in the next linear)
__global__ void cal_layernorm_kernel(const float* __restrict__ R, float* __restrict__ Xn, int M, float eps) {
    const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    const float a = R[(int64_t)m * DIM + lane], b = R[(int64_t)m * DIM + 64 + lane];
    float s = a + b;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / DIM);
    const float da = a - mean, db = b - mean;
    float q = da * da + db * db;
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q * (1.0f / DIM) + eps);
    Xn[(int64_t)m * DIM + lane] = da * rstd;
    Xn[(int64_t)m * DIM + 64 + lane] = db * rstd;
}

// softmax attention of one (clip-crop, head): QKV[m][which * inner + h * hd + d] (models/uit.py:110-111 / 147-148), the q rows
// already carry 0.125 log2(e) (pack-time fold), so probabilities are exp2(s - max).  O[m][h * hd + d] (:119 / :175).
// nt: valid time patches of the crop (short clips: tokens of time patch >= nt do not exist -- never a key; their own rows are garbage nobody reads)
__device__ __forceinline__ bool naive_token_valid(int tok, int ntok, int nt) {
    if (ntok != NTOK) { if (tok == 0) return true; tok -= 1; }          // the cls token of pooling 'token'
    return tok % GT < nt;
}
// One workgroup per clip-crop, both heads: the crop's QKV rows are staged in LDS (coalesced), thread t = (head, query token).
__global__ void __launch_bounds__(64) cal_attention_kernel(const float* __restrict__ QKV, float* __restrict__ O, int ntok, int inner, int nt) {
    extern __shared__ float qkv[];                                         // [ntok][3 * inner + 1] (odd row pitch: the queries' reads spread over the banks)
    const int row = blockIdx.x, hd = inner / 2, lg = 3 * inner, ld = lg + 1;
    const float* base = QKV + (int64_t)row * ntok * lg;
    for (int i = threadIdx.x; i < ntok * lg; i += 64) qkv[(i / lg) * ld + i % lg] = base[i];
    __syncthreads();
    const int h = threadIdx.x / 32, t = threadIdx.x % 32;
    if (t >= ntok) return;
    const float* q = qkv + t * ld + h * hd;
    float s[NTOK + 1], mx = -INFINITY;
    for (int j = 0; j < ntok; ++j) {
        s[j] = -INFINITY;
        if (!naive_token_valid(j, ntok, nt)) continue;
        const float* k = qkv + j * ld + inner + h * hd;
        float acc = 0.f;
        for (int d = 0; d < hd; ++d) acc += q[d] * k[d];
        s[j] = acc; mx = fmaxf(mx, acc);
    }
    float sum = 0.f;
    for (int j = 0; j < ntok; ++j) { s[j] = s[j] == -INFINITY ? 0.f : exp2f(s[j] - mx); sum += s[j]; }
    const float inv = 1.0f / sum;
    float* o = O + ((int64_t)row * ntok + t) * inner + h * hd;
    for (int d = 0; d < hd; ++d) {
        float acc = 0.f;
        for (int j = 0; j < ntok; ++j) if (s[j] != 0.f) acc += s[j] * qkv[j * ld + 2 * inner + h * hd + d];
        o[d] = acc * inv;
    }
}

// G[i][j] = scale * sum_m x[m][i] x[m][j], x = the 16-bit operand value of X (of X - mu when mu is given: the centred operand);  K multiple of 16
__global__ void cal_gram_kernel(const float* __restrict__ X, float* __restrict__ G, int M, int K, float scale, int bf16, const float* __restrict__ mu) {
    __shared__ float xi[TS][TS + 1], xj[TS][TS + 1];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int i0 = blockIdx.y * TS, j0 = blockIdx.x * TS;
    float acc = 0.f;
    for (int m0 = 0; m0 < M; m0 += TS) {
        const int m = m0 + ty;
        xi[ty][tx] = m < M ? round_op16(X[(int64_t)m * K + i0 + tx] - (mu ? mu[i0 + tx] : 0.f), bf16) : 0.f;
        xj[ty][tx] = m < M ? round_op16(X[(int64_t)m * K + j0 + tx] - (mu ? mu[j0 + tx] : 0.f), bf16) : 0.f;
        __syncthreads();
#pragma unroll
        for (int mm = 0; mm < TS; ++mm) acc += xi[mm][ty] * xj[mm][tx];
        __syncthreads();
    }
    G[(int64_t)(i0 + ty) * K + j0 + tx] = acc * scale;
}

// mu[k] = scale * sum_m X[m][k]  (the fp32 values, not the rounded operands: what the kernel subtracts BEFORE it rounds)
__global__ void cal_colmean_kernel(const float* __restrict__ X, float* __restrict__ mu, int M, int K, float scale) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float acc = 0.f, comp = 0.f;
    for (int m = 0; m < M; ++m) {                                           // Kahan: M is up to ~25 000 rows
        const float y = X[(int64_t)m * K + k] - comp, t = acc + y;
        comp = (t - acc) - y; acc = t;
    }
    mu[k] = acc * scale;
}

static hipError_t linear(const float* X, const float* W, const float* b, float* Y, int M, int K, int O, int act, int accumulate, hipStream_t s) {
    if (M >= 8192 && K >= 512 && O >= 256) {                                 // wide layers (a speed choice only: both kernels give the same bits)
        hipLaunchKernelGGL(naive_linear_mfma_big_kernel, dim3((O + BT - 1) / BT, (M + BT - 1) / BT), dim3(256), 0, s, X, W, b, Y, M, K, O, act, accumulate);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(naive_linear_mfma_kernel, dim3((O + LT - 1) / LT, (M + LT - 1) / LT), dim3(256), 0, s, X, W, b, Y, M, K, O, act, accumulate);
    return hipGetLastError();
}
// the same dense layer for the general engine (general.hip)
hipError_t launch_naive_linear(const float* X, const float* W, const float* b, float* Y, int M, int K, int O, int act, int accumulate, hipStream_t s) {
    return linear(X, W, b, Y, M, K, O, act, accumulate, s);
}
static hipError_t gram(const float* X, float* G, int M, int K, float scale, int bf16, hipStream_t s, const float* mu = nullptr) {
    hipLaunchKernelGGL(cal_gram_kernel, dim3(K / TS, K / TS), dim3(TS, TS), 0, s, X, G, M, K, scale, bf16, mu);
    return hipGetLastError();
}

// The head of the reference on the last block's residual stream (models/uit.py:395, 398-412): final LayerNorm (eps 1e-6) per token ->
// pooling over the VALID tokens ('mean' :402-404, 'token' = the cls row :399-401, 'dm' :405-412) -> outputlayer = LayerNorm (eps 1e-5)
// + Linear -> sigmoid.  One 128-thread workgroup per clip-crop; sums in double (cheap here, and the rung exists to be right).
__global__ void __launch_bounds__(128) naive_head_kernel(const float* __restrict__ R, NaiveHeadArgs h, int ntok, int nt) {
    __shared__ double red[128];
    __shared__ float zn[DIM];
    const int row = blockIdx.x, c = threadIdx.x;
    auto block_sum = [&](double v) {
        red[c] = v; __syncthreads();
        for (int o = 64; o > 0; o >>= 1) { if (c < o) red[c] += red[c + o]; __syncthreads(); }
        const double r = red[0]; __syncthreads();
        return r;
    };
    const int ngroups = h.pooling == 1 ? nt : 1;                          // 'dm': one pooled vector per time patch
    double acc[GT] = {};
    int cnt[GT] = {};
    for (int t = 0; t < ntok; ++t) {
        if (!naive_token_valid(t, ntok, nt)) continue;
        if (h.pooling == 2 && t != 0) continue;
        const double x = R[((int64_t)row * ntok + t) * DIM + c];
        const double mean = block_sum(x) / DIM, d = x - mean;
        const double rstd = 1.0 / sqrt(block_sum(d * d) / DIM + 1e-6);
        const double y = d * rstd * h.norm_w[c] + h.norm_b[c];
        const int tp = ntok == NTOK ? t % GT : (t == 0 ? 0 : (t - 1) % GT);
        const int gsel = h.pooling == 1 ? tp : 0;
        acc[gsel] += y; cnt[gsel] += 1;
    }
    float* probs = h.rowprobs + (int64_t)row * h.C;
    for (int o = c; o < h.C; o += 128) probs[o] = 0.f;
    for (int gq = 0; gq < ngroups; ++gq) {
        const double z = acc[gq] / (double)(cnt[gq] > 0 ? cnt[gq] : 1);
        const double mean = block_sum(z) / DIM, d = z - mean;
        const double rstd = 1.0 / sqrt(block_sum(d * d) / DIM + 1e-5);
        zn[c] = (float)(d * rstd * h.hln_w[c] + h.hln_b[c]);
        __syncthreads();
        for (int o = c; o < h.C; o += 128) {
            double lg = h.head_b[o];
            const float* wrow = h.head_w + (int64_t)o * DIM;
            for (int i = 0; i < DIM; ++i) lg += (double)zn[i] * (double)wrow[i];
            if (h.rowlogits != nullptr && h.pooling != 1) h.rowlogits[(int64_t)row * h.C + o] = (float)lg;
            probs[o] += (float)(1.0 / (1.0 + exp(-lg)) / ngroups);
        }
        __syncthreads();
    }
}

hipError_t launch_calibration(const CalibArgs& a, hipStream_t s) { return launch_naive_forward(a, GT, nullptr, s, nullptr); }

// the forward (statistics when a.stats is set), with the attention masked for nt < GT valid time patches, + the head when asked for
hipError_t launch_naive_forward(const CalibArgs& a, int nt, const NaiveHeadArgs* head, hipStream_t s, const NaiveFramesArgs* frames) {
#define CAL_TRY(e) do { hipError_t _e = (e); if (_e != hipSuccess) return _e; } while (0)
    const int M = a.rows * a.ntok, inner = a.inner;
    const int npatch = a.rows * NTOK;                                      // tokens that carry a patch
    if (frames != nullptr) hipLaunchKernelGGL(naive_patch_from_frames_kernel, dim3(1024), dim3(256), 0, s, *frames, a.amax, a.top_db, a.bn_sc, a.bn_sh, a.rows, a.ntok, nt, a.X0);
    else hipLaunchKernelGGL(cal_patch_kernel, dim3(1024), dim3(256), 0, s, a.mel, a.amax, a.top_db, a.bn_sc, a.bn_sh, a.op_bf16, a.rows, a.ntok, a.X0);
    CAL_TRY(hipGetLastError());
    if (a.stats) CAL_TRY(gram(a.X0, a.stats + a.off_pe, M, 256, 1.0f / (float)npatch, a.op_bf16, s));
    hipLaunchKernelGGL(cal_init_stream_kernel, dim3(512), dim3(256), 0, s, a.pe_bias, M, a.ntok, a.R);
    CAL_TRY(hipGetLastError());
    CAL_TRY(linear(a.X0, a.pe_w, nullptr, a.R, M, 256, DIM, 0, 1, s));
    const float inv_m = 1.0f / (float)M;
    for (int b = 0; b < a.depth; ++b) {
        const float* wb = a.blocks + (size_t)b * a.block_floats;
        float* st = a.stats ? a.stats + (size_t)b * a.per_block : nullptr;      // nullptr: forward only (the acceptance check)
        hipLaunchKernelGGL(cal_layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, s, a.R, a.Xn, M, 1e-6f);
        CAL_TRY(hipGetLastError());
        if (st) {
            float* mu = a.stats + a.off_mu1 + (size_t)b * DIM;
            hipLaunchKernelGGL(cal_colmean_kernel, dim3(DIM / 64), dim3(64), 0, s, a.Xn, mu, M, DIM, inv_m);
            CAL_TRY(hipGetLastError());
            CAL_TRY(gram(a.Xn, st + a.off_ln1, M, DIM, inv_m, a.op_bf16, s, a.centre_ln1 ? mu : nullptr));
        }
        CAL_TRY(linear(a.Xn, wb + a.o_qkv_w, wb + a.o_qkv_b, a.QKV, M, DIM, 3 * inner, 0, 0, s));
        hipLaunchKernelGGL(cal_attention_kernel, dim3(a.rows), dim3(64), (size_t)a.ntok * (3 * inner + 1) * 4, s, a.QKV, a.O, a.ntok, inner, nt);
        CAL_TRY(hipGetLastError());
        if (st) CAL_TRY(gram(a.O, st + a.off_att, M, inner, inv_m, a.op_bf16, s));
        CAL_TRY(linear(a.O, wb + a.o_proj_w, wb + a.o_proj_b, a.R, M, inner, DIM, 0, 1, s));
        hipLaunchKernelGGL(cal_layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, s, a.R, a.Xn, M, 1e-6f);
        CAL_TRY(hipGetLastError());
        if (st) CAL_TRY(gram(a.Xn, st + a.off_ln2, M, DIM, inv_m, a.op_bf16, s));
        CAL_TRY(linear(a.Xn, wb + a.o_fc1_w, wb + a.o_fc1_b, a.Hd, M, DIM, HID, a.act_gelu ? 2 : 1, 0, s));
        if (st) CAL_TRY(gram(a.Hd, st + a.off_hid, M, HID, inv_m, a.op_bf16, s));
        CAL_TRY(linear(a.Hd, wb + a.o_fc2_w, wb + a.o_fc2_b, a.R, M, HID, DIM, 0, 1, s));
    }
    if (head != nullptr) {
        hipLaunchKernelGGL(naive_head_kernel, dim3(a.rows), dim3(128), 0, s, a.R, *head, a.ntok, nt);
        CAL_TRY(hipGetLastError());
    }
#undef CAL_TRY
    return hipSuccess;
}

}  // namespace uitk
