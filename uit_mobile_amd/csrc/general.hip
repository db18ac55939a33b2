// general.hip -- the GENERAL engine: UITBase.forward (eval) for every geometry the reference's constructor admits, in fp32.
//
// The MFMA kernels of this library (frontend.hip, encoder.hip) are specialised to the geometry of the seven published factories
// (models/uit.py:514-635: n_fft 512, 64 mels, 16 x 16 patches, 4 x 6 tokens, embed 128, 2 heads, MLP x 3).  `UITBase(**kwargs)` itself
// admits any of n_mels / n_fft / win_size / hop_size / patch_size / target_length / embed_dim / depth / num_heads / mlp_ratio
// (models/uit.py:253-363); a checkpoint trained with other values used to be REFUSED here.  This file runs those on the device too:
// plain fp32 arithmetic in the reference's own order of operations, one simple kernel per operation (the dense layers on the matrix
// pipe's exact-fp32 mode: calibrate.hip's naive_linear_mfma_kernel), nothing rounded to 16 bits, no calibration, no tuning.  It is the
// general-geometry sibling of precision 'f32' (calibrate.hip: launch_naive_forward): right first, an order of magnitude slower than
// the MFMA encoder, two orders faster than the CPU path.  Never used for a configuration the MFMA kernels accept.
//
//   front-end   models/uit.py:298-308: torchaudio 0.13 MelSpectrogram + AmplitudeToDB(top_db=120), the batch-global floor applied by
//               the consumer (gen_patch_kernel) once the whole call's amax is known -- as in the fast path
//   patches     models/uit.py:460-462 (init_bn), :63-66 (Conv2d k = stride = patch), :381-392 (positional embeddings, cls token)
//   blocks      models/uit.py:206-248; attention :89-122 (BNeckAttention) / :124-178 (Attention); Mlp :181-203
//   head        models/uit.py:395-412 (norm, pooling 'mean' / 'token' / 'dm', outputlayer, sigmoid), crops :468-488
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/uitk.h"
#include "naive_forward.h"
#include "uitk_dev.h"

namespace uitk {

int set_error(int code, const std::string& msg);                          // api.cpp: the thread's uitk_last_error text

// ------------------------------------------------------------------------------------------------------------------ front-end
struct GenFrontArgs {
    const void* pcm; int i16;
    int64_t B, T;
    int nF, hop, pad, N, nb, M, log2N;      // frames per clip, hop, reflect padding, n_fft, n_fft/2 + 1, n_mels, log2(n_fft) or 0 (direct DFT)
    const float* wpad;                      // [N]  the window, zero-padded to n_fft as torch.stft does (left = (n_fft - win) / 2)
    const float2* tw;                       // [N]  (cos, sin)(2 pi j / N), rounded from double
    const float* fb;                        // [nb][M]
    const int2* band;                       // [M]  [first, last + 1) bins with a non-zero filterbank weight: the sum skips the exact zeros outside (same bits)
    float* frames;                          // [B][nF][M] dB (10 log10 max(., 1e-10)), no top_db clamp yet
    float* amax;                            // [AMAX_SLOTS], armed with -inf by gen_arm_kernel
};

__global__ void gen_arm_kernel(float* amax) { if (threadIdx.x < AMAX_SLOTS) amax[threadIdx.x] = -INFINITY; }

// One workgroup per frame at a time.  Power-of-two n_fft: an in-LDS radix-2 FFT of the (real, 0) samples -- the rounding of an
// FFT, as in the reference (torch.stft); other n_fft: the DFT sum itself, accumulated in double.
__global__ void __launch_bounds__(256) gen_frontend_kernel(GenFrontArgs a) {
    extern __shared__ float2 lds[];
    float2* buf = lds;                              // [N]
    float* pw = (float*)(lds + a.N);                // [nb]
    float2* twl = (float2*)(pw + ((a.nb + 1) & ~1));   // [N] the DFT path's twiddles (a gather per term: LDS, not global memory)
    const int tid = threadIdx.x;
    if (!a.log2N) {
        for (int n = tid; n < a.N; n += 256) twl[n] = a.tw[n];
        __syncthreads();
    }
    float lmax = -INFINITY;
    const int64_t total = a.B * (int64_t)a.nF;
    for (int64_t fid = blockIdx.x; fid < total; fid += gridDim.x) {
        const int64_t b = fid / a.nF;
        const int f = (int)(fid - b * a.nF);
        const int64_t base = (int64_t)f * a.hop - a.pad;
        for (int n = tid; n < a.N; n += 256) {
            int64_t i = base + n;
            if (i < 0) i = -i;                                             // reflect, the edge sample not repeated (torch.stft center=True)
            if (i >= a.T) i = 2 * (a.T - 1) - i;
            const float x = a.i16 ? (float)((const int16_t*)a.pcm)[b * a.T + i] * (1.0f / 32768.0f) : ((const float*)a.pcm)[b * a.T + i];
            const float v = x * a.wpad[n];
            const int dst = a.log2N ? (int)(__brev((unsigned)n) >> (32 - a.log2N)) : n;
            buf[dst] = make_float2(v, 0.f);
        }
        __syncthreads();
        if (a.log2N) {
            for (int s = 1; s <= a.log2N; ++s) {
                const int half = 1 << (s - 1), step = a.N >> s;
                for (int j = tid; j < a.N / 2; j += 256) {
                    const int pos = j & (half - 1), i0 = ((j >> (s - 1)) << s) + pos, i1 = i0 + half;
                    const float2 w = a.tw[pos * step];                     // exp(-2 pi i pos / 2^s) = (cos, -sin)
                    const float2 u = buf[i0], q = buf[i1];
                    const float tr = q.x * w.x + q.y * w.y, ti = q.y * w.x - q.x * w.y;
                    buf[i0] = make_float2(u.x + tr, u.y + ti);
                    buf[i1] = make_float2(u.x - tr, u.y - ti);
                }
                __syncthreads();
            }
            for (int k = tid; k < a.nb; k += 256) pw[k] = buf[k].x * buf[k].x + buf[k].y * buf[k].y;
        } else {
            for (int k = tid; k < a.nb; k += 256) {
                double re = 0.0, im = 0.0;
                int idx = 0;
                for (int n = 0; n < a.N; ++n) {
                    const float2 w = twl[idx];
                    const double x = (double)buf[n].x;
                    re += x * (double)w.x; im -= x * (double)w.y;
                    idx += k; if (idx >= a.N) idx -= a.N;
                }
                const float fr = (float)re, fi = (float)im;
                pw[k] = fr * fr + fi * fi;
            }
        }
        __syncthreads();
        for (int m = tid; m < a.M; m += 256) {
            float acc = 0.f;
            const int2 bd = a.band[m];
            for (int k = bd.x; k < bd.y; ++k) acc = fmaf(pw[k], a.fb[(int64_t)k * a.M + m], acc);
            const float db = 10.0f * log10f(fmaxf(acc, 1e-10f));
            a.frames[(b * a.nF + f) * a.M + m] = db;
            lmax = fmaxf(lmax, db);
        }
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if ((tid & 63) == 0 && lmax > -INFINITY) {                             // float max by integer atomics on the bits (see frontend.hip)
        float* slot = a.amax + (blockIdx.x & (AMAX_SLOTS - 1));
        if ((__float_as_uint(lmax) >> 31) == 0u) atomicMax((int*)slot, __float_as_int(lmax));
        else atomicMin((unsigned*)slot, __float_as_uint(lmax));
    }
}

// out[b][m][f] = max(frames[b][f][m], amax - top_db): what self.front_end(x) returns in the reference ('b f t')
__global__ void gen_mel_export_kernel(const float* __restrict__ frames, const float* __restrict__ amax, float top_db, int64_t B, int nF, int M,
                                      float* __restrict__ out) {
    float mx = -INFINITY;
    for (int i = 0; i < AMAX_SLOTS; ++i) mx = fmaxf(mx, amax[i]);
    const float flo = mx - top_db;
    const int64_t n = B * (int64_t)nF * M;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / ((int64_t)nF * M);
        const int r = (int)(i - b * nF * M), m = r / nF, f = r - m * nF;
        out[i] = fmaxf(frames[(b * nF + f) * M + m], flo);
    }
}

// ------------------------------------------------------------------------------------------------------------------ network
struct GenGeom {
    int D, heads, inner, hd, hidden, P, GF, gt_model, M, C, depth, pooling, act_gelu;
    int ntok, np, cls, nt;                  // tokens per clip-crop (incl. cls), patch tokens, 1 with a cls token, time patches of this call
};

// X0[(row * ntok + tok)][fi * P + ti] = init_bn(max(dB, floor)) of patch (pf, pt) of the row's crop; the cls row: zeros.
// Crops (models/uit.py:468-481): start every target_length frames, the last one is the last target_length frames.
__global__ void gen_patch_kernel(const float* __restrict__ frames, const float* __restrict__ amax, float top_db, const float* __restrict__ bn,
                                 GenGeom g, int nF, int ncrops, int tl, int64_t row0, int rows, float* __restrict__ X0) {
    float amx = amax[0];
    for (int i = 1; i < AMAX_SLOTS; ++i) amx = fmaxf(amx, amax[i]);
    const float floor_db = amx - top_db;
    const int K = g.P * g.P;
    const int64_t n = (int64_t)rows * g.ntok * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int64_t rt = i / K;
        const int tok = (int)(rt % g.ntok) - g.cls;
        float v = 0.f;
        if (tok >= 0) {
            const int64_t rowg = row0 + rt / g.ntok, b = rowg / ncrops;
            const int crop = (int)(rowg % ncrops);
            const int pf = tok / g.nt, pt = tok % g.nt, fi = k / g.P, ti = k % g.P;
            const int mel = pf * g.P + fi, t = pt * g.P + ti;
            const int f = ncrops == 1 ? t : (crop < ncrops - 1 ? crop * tl + t : nF - tl + t);
            const float x = fmaxf(frames[(b * nF + f) * g.M + mel], floor_db);
            // eval BatchNorm2d over the mel bins (models/uit.py:310-313): ((x - mean) * invstd) * weight + bias
            v = (x - bn[2 * g.M + mel]) * bn[3 * g.M + mel] * bn[mel] + bn[g.M + mel];
        }
        X0[i] = v;
    }
}

// R[m][c] = conv bias + time_pos_embed[.., pt] + freq_pos_embed[.., pf, ..]  (models/uit.py:381-383); cls row: cls_token + token_pos_embed (:389-391)
__global__ void gen_init_stream_kernel(const float* __restrict__ pe_b, const float* __restrict__ time_pos, const float* __restrict__ freq_pos,
                                       const float* __restrict__ cls, const float* __restrict__ token_pos, GenGeom g, int rows, float* __restrict__ R) {
    const int64_t n = (int64_t)rows * g.ntok * g.D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % g.D);
        const int tok = (int)((i / g.D) % g.ntok) - g.cls;
        if (tok < 0) { R[i] = cls[c] + token_pos[c]; continue; }
        const int pf = tok / g.nt, pt = tok % g.nt;
        R[i] = (pe_b[c] + time_pos[c * g.gt_model + pt]) + freq_pos[c * g.GF + pf];
    }
}

// Y[m][:] = LayerNorm(X[m][:]) * w + b, one wave per row, any D <= 64 * 32
__global__ void __launch_bounds__(256) gen_layernorm_kernel(const float* __restrict__ X, float* __restrict__ Y, const float* __restrict__ w,
                                                            const float* __restrict__ bias, int64_t M, int D, float eps) {
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (m >= M) return;
    const float* x = X + m * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += x[c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)D;
    float q = 0.f;
    for (int c = lane; c < D; c += 64) { const float d = x[c] - mean; q += d * d; }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)D + eps);
    for (int c = lane; c < D; c += 64) Y[m * D + c] = (x[c] - mean) * rstd * w[c] + bias[c];
}

// softmax attention, thread = one query of one (clip-crop, head): scores (q . k) * scale, max, exp, sum, (p . v) / sum -- two passes over
// the keys (the reference's order: softmax first, then @ v).  QKV[m][which * inner + h * hd + d] (models/uit.py:110-111 / 147-148).
// A workgroup = up to 256 queries of one (clip-crop, head); every thread walks the same keys, so the head's k and v rows are staged in
// LDS in chunks of CH keys (all of them at once when they fit: one staging serves both passes) and every read is a broadcast.
// VEC (head widths that are multiples of 4): the staged rows are read 16 bytes at a time -- the kernel is bound by LDS instructions
// (one broadcast read per multiply-add otherwise); the products are summed in the same order either way.
typedef float gen_f4 __attribute__((ext_vector_type(4)));
template <int HDP, bool VEC>
__device__ __forceinline__ float gen_dot(const float (&q)[HDP], const float* k, int hd) {
    float s = 0.f;
    if constexpr (VEC) {
#pragma unroll
        for (int d4 = 0; d4 < HDP / 4; ++d4)
            if (4 * d4 < hd) {
                const gen_f4 kk = ((const gen_f4*)k)[d4];
                s = fmaf(q[4 * d4], kk[0], s); s = fmaf(q[4 * d4 + 1], kk[1], s); s = fmaf(q[4 * d4 + 2], kk[2], s); s = fmaf(q[4 * d4 + 3], kk[3], s);
            }
    } else {
#pragma unroll
        for (int d = 0; d < HDP; ++d) if (d < hd) s = fmaf(q[d], k[d], s);
    }
    return s;
}
template <int HDP, bool VEC>
__global__ void __launch_bounds__(256) gen_attention_kernel(const float* __restrict__ QKV, float* __restrict__ O, int ntok, int heads, int inner, int hd, float scale, int CH) {
    extern __shared__ __attribute__((aligned(16))) float kv[];             // K [CH][hd], then V [CH][hd]
    const int row = blockIdx.x / heads, h = blockIdx.x % heads, t = blockIdx.y * blockDim.x + threadIdx.x;
    const int lg = 3 * inner, nch = (ntok + CH - 1) / CH;
    const float* base = QKV + (int64_t)row * ntok * lg;
    const bool active = t < ntok;                                          // (idle lanes still stage and synchronise)
    auto stage = [&](int j0, int n) {
        for (int i = threadIdx.x; i < n * hd; i += blockDim.x) {
            const int j = i / hd, d = i - j * hd;
            kv[i] = base[(int64_t)(j0 + j) * lg + inner + h * hd + d];
            kv[CH * hd + i] = base[(int64_t)(j0 + j) * lg + 2 * inner + h * hd + d];
        }
    };
    float q[HDP], acc[HDP];
#pragma unroll
    for (int d = 0; d < HDP; ++d) { q[d] = active && d < hd ? base[(int64_t)t * lg + h * hd + d] : 0.f; acc[d] = 0.f; }
    float mx = -INFINITY;
    for (int c = 0; c < nch; ++c) {
        const int j0 = c * CH, n = min(CH, ntok - j0);
        __syncthreads();
        stage(j0, n);
        __syncthreads();
        if (active)
            for (int j = 0; j < n; ++j) {
                const float s = gen_dot<HDP, VEC>(q, kv + j * hd, hd);
                mx = fmaxf(mx, s * scale);
            }
    }
    float sum = 0.f;
    for (int c = 0; c < nch; ++c) {
        const int j0 = c * CH, n = min(CH, ntok - j0);
        if (nch > 1) {                                                     // (a single chunk is still there from the first pass)
            __syncthreads();
            stage(j0, n);
            __syncthreads();
        }
        if (active)
            for (int j = 0; j < n; ++j) {
                const float* v = kv + CH * hd + j * hd;
                const float s = gen_dot<HDP, VEC>(q, kv + j * hd, hd);
                const float p = expf(s * scale - mx);
                sum += p;
                if constexpr (VEC) {
#pragma unroll
                    for (int d4 = 0; d4 < HDP / 4; ++d4)
                        if (4 * d4 < hd) {
                            const gen_f4 vv = ((const gen_f4*)v)[d4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[4 * d4 + e] = fmaf(p, vv[e], acc[4 * d4 + e]);
                        }
                } else {
#pragma unroll
                    for (int d = 0; d < HDP; ++d) if (d < hd) acc[d] = fmaf(p, v[d], acc[d]);
                }
            }
    }
    if (!active) return;
    float* o = O + ((int64_t)row * ntok + t) * inner + h * hd;
#pragma unroll
    for (int d = 0; d < HDP; ++d) if (d < hd) o[d] = acc[d] / sum;
}

// heads wider than 128 (a single head on a wide model): thread = (query, 64-wide slice of the head).  Every thread forms the whole q . k
// (q and k re-read from memory -- nothing of that width fits the register file) and keeps its slice of the output.
__global__ void __launch_bounds__(64) gen_attention_wide_kernel(const float* __restrict__ QKV, float* __restrict__ O, int ntok, int heads, int inner, int hd, float scale) {
    const int row = blockIdx.x / heads, h = blockIdx.x % heads, t = blockIdx.y * 64 + threadIdx.x, d0 = blockIdx.z * 64;
    if (t >= ntok) return;
    const int lg = 3 * inner, nd = min(64, hd - d0);
    const float* base = QKV + (int64_t)row * ntok * lg;
    const float* q = base + (int64_t)t * lg + h * hd;
    float acc[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) acc[d] = 0.f;
    float mx = -INFINITY;
    for (int j = 0; j < ntok; ++j) {
        const float* k = base + (int64_t)j * lg + inner + h * hd;
        float s = 0.f;
        for (int d = 0; d < hd; ++d) s = fmaf(q[d], k[d], s);
        mx = fmaxf(mx, s * scale);
    }
    float sum = 0.f;
    for (int j = 0; j < ntok; ++j) {
        const float* k = base + (int64_t)j * lg + inner + h * hd;
        const float* v = base + (int64_t)j * lg + 2 * inner + h * hd + d0;
        float s = 0.f;
        for (int d = 0; d < hd; ++d) s = fmaf(q[d], k[d], s);
        const float p = expf(s * scale - mx);
        sum += p;
#pragma unroll
        for (int d = 0; d < 64; ++d) if (d < nd) acc[d] = fmaf(p, v[d], acc[d]);
    }
    float* o = O + ((int64_t)row * ntok + t) * inner + h * hd + d0;
#pragma unroll
    for (int d = 0; d < 64; ++d) if (d < nd) o[d] = acc[d] / sum;
}

struct GenHeadArgs {
    const float* hln_w; const float* hln_b;          // outputlayer.0 (LayerNorm, eps 1e-5)  models/uit.py:358-360
    const float* head_w; const float* head_b;        // outputlayer.1 (Linear) [C][D], [C]
    float* rowprobs;                                 // [rows][C]
    float* rowlogits;                                // [rows][C] or nullptr (never written for pooling 'dm')
};

// Xn = norm(x) per token (gen_layernorm_kernel, eps 1e-6) -> pooling -> outputlayer -> sigmoid  (models/uit.py:395-412).  One workgroup
// per clip-crop; the sums in double.
__global__ void __launch_bounds__(256) gen_head_kernel(const float* __restrict__ Xn, GenHeadArgs h, GenGeom g) {
    extern __shared__ float zn[];                    // [D]
    __shared__ double red[256];
    const int row = blockIdx.x, tid = threadIdx.x;
    auto block_sum = [&](double v) {
        red[tid] = v; __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
        const double r = red[0]; __syncthreads();
        return r;
    };
    const float* x = Xn + (int64_t)row * g.ntok * g.D;
    float* probs = h.rowprobs + (int64_t)row * g.C;
    for (int o = tid; o < g.C; o += 256) probs[o] = 0.f;
    const int ngroups = g.pooling == 1 ? g.nt : 1;
    for (int gq = 0; gq < ngroups; ++gq) {
        // pooled vector of this group: 'mean' all patch tokens (:402-404); 'token' the cls row (:399-401); 'dm' the GF tokens of time patch gq (:405-412)
        double part = 0.0, part2 = 0.0;
        for (int c = tid; c < g.D; c += 256) {
            double z = 0.0;
            if (g.pooling == 2) z = x[c];
            else if (g.pooling == 0) { for (int t = 0; t < g.ntok; ++t) z += x[(int64_t)t * g.D + c]; z /= (double)g.ntok; }
            else { for (int pf = 0; pf < g.GF; ++pf) z += x[(int64_t)(pf * g.nt + gq) * g.D + c]; z /= (double)g.GF; }
            zn[c] = (float)z;
            part += z;
        }
        const double mean = block_sum(part) / g.D;
        for (int c = tid; c < g.D; c += 256) { const double d = (double)zn[c] - mean; part2 += d * d; }
        const double rstd = 1.0 / sqrt(block_sum(part2) / g.D + 1e-5);
        for (int c = tid; c < g.D; c += 256) zn[c] = (float)(((double)zn[c] - mean) * rstd * h.hln_w[c] + h.hln_b[c]);
        __syncthreads();
        for (int o = tid; o < g.C; o += 256) {
            double lg = h.head_b[o];
            const float* wrow = h.head_w + (int64_t)o * g.D;
            for (int i = 0; i < g.D; ++i) lg += (double)zn[i] * (double)wrow[i];
            if (h.rowlogits != nullptr && g.pooling != 1) h.rowlogits[(int64_t)row * g.C + o] = (float)lg;
            probs[o] += (float)(1.0 / (1.0 + exp(-lg)) / ngroups);
        }
        __syncthreads();
    }
}

}  // namespace uitk

// ------------------------------------------------------------------------------------------------------------------ host side + C ABI
using namespace uitk;

struct GenBlockPtrs { const float *n1w, *n1b, *qkv_w, *qkv_b, *proj_w, *proj_b, *n2w, *n2b, *fc1_w, *fc1_b, *fc2_w, *fc2_b; };

struct uitk_general {
    uitk_config cfg;
    GenGeom g{};
    int n_cu = 256, log2N = 0, nb = 0;
    float* params = nullptr;                 // every parameter, one device blob
    const float *wpad = nullptr, *fb = nullptr, *bn = nullptr, *pe_w = nullptr, *pe_b = nullptr, *time_pos = nullptr, *freq_pos = nullptr,
                *cls = nullptr, *token_pos = nullptr, *norm_w = nullptr, *norm_b = nullptr, *hln_w = nullptr, *hln_b = nullptr, *head_w = nullptr, *head_b = nullptr;
    const float2* tw = nullptr;
    const int2* band = nullptr;
    std::vector<GenBlockPtrs> blocks;
    float* amax = nullptr;                   // [AMAX_SLOTS]
    float* frames = nullptr; size_t frames_cap = 0;
    float* work = nullptr; size_t work_cap = 0; int64_t work_rows = 0; int work_ntok = 0;
    float* rowprobs = nullptr; size_t rowprobs_cap = 0;
    int64_t fe_B = 0, fe_T = 0;
};

#define GEN_HIP(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess) return set_error(UITK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

static int gen_num_frames(const uitk_general* m, int64_t T) {
    if (m->cfg.center) return (int)(1 + T / m->cfg.hop_size);
    return T < m->cfg.n_fft ? 0 : (int)(1 + (T - m->cfg.n_fft) / m->cfg.hop_size);
}
static int gen_num_crops(const uitk_general* m, int64_t T) {
    const int nF = gen_num_frames(m, T), tl = m->cfg.target_length;
    return nF <= tl ? 1 : (nF + tl - 1) / tl;
}
// time patches of a clip-crop: a conv with kernel = stride = P over min(nF, target_length) frames
static int gen_time_patches(const uitk_general* m, int64_t T) {
    const int nF = gen_num_frames(m, T);
    return (nF <= m->cfg.target_length ? nF : m->cfg.target_length) / m->cfg.patch_size;
}

static int gen_validate(const uitk_config* c, std::string* why) {
    auto bad = [&](const char* s) { *why = s; return UITK_ERR_UNSUPPORTED; };
    if (c->precision != UITK_PREC_F32) return bad("the general engine computes in fp32: precision must be UITK_PREC_F32");
    if (c->n_fft < 16 || c->n_fft > 4096 || (c->n_fft & 1)) return bad("general engine: n_fft must be even, 16..4096");
    if ((c->n_fft & (c->n_fft - 1)) && c->n_fft > 2048) return bad("general engine: an n_fft that is no power of two must be <= 2048 (the DFT path keeps its twiddles in LDS)");
    if (c->win_size < 1 || c->win_size > c->n_fft) return bad("win_size must be 1..n_fft");
    if (c->hop_size < 1) return bad("hop_size must be positive");
    if (c->n_mels < 1 || c->n_mels > 512) return bad("general engine: n_mels must be 1..512");
    if (c->patch_size < 4 || c->patch_size > 64 || (c->patch_size & 3)) return bad("general engine: patch_size must be a multiple of 4, 4..64 (the dense layers run 16-deep k-steps)");
    if (c->n_mels / c->patch_size < 1) return bad("n_mels smaller than one patch");
    if (c->target_length / c->patch_size < 1) return bad("target_length smaller than one patch");
    if (c->embed_dim < 16 || c->embed_dim > 2048 || (c->embed_dim & 15)) return bad("general engine: embed_dim must be a multiple of 16, 16..2048");
    if (c->mlp_hidden < 16 || (c->mlp_hidden & 15)) return bad("general engine: the MLP's hidden width must be a multiple of 16");
    if (c->num_heads < 1 || c->embed_dim % c->num_heads) return bad("embed_dim must be divisible by num_heads");
    if ((c->attention_full != 0 && c->attention_full != 1) || (c->act_gelu != 0 && c->act_gelu != 1)) return bad("attention_full / act_gelu must be 0 or 1");
    const int inner = c->attention_full ? c->embed_dim : c->embed_dim / 4;
    if (inner < 16 || (inner & 15)) return bad("general engine: the attention's inner width (embed_dim, or embed_dim / 4 for BNeckAttention) must be a multiple of 16");
    if (inner % c->num_heads) return bad("the attention's inner width must be divisible by num_heads");
    if (c->center != 0 && c->center != 1) return bad("center must be 0 or 1");
    if (c->depth < 1 || c->depth > 64) return bad("depth out of range");
    if (c->outputdim < 1 || c->outputdim > 65536) return bad("outputdim out of range");
    if (c->eval_avg != UITK_AVG_MEAN && c->eval_avg != UITK_AVG_MAX) return bad("unknown eval_avg");
    if (c->pooling < 0 || c->pooling > 2) return bad("pooling must be 0 (mean), 1 (dm) or 2 (token)");
    return UITK_OK;
}

static int gen_check_shape(const uitk_general* m, int64_t B, int64_t T) {
    if (!m) return set_error(UITK_ERR_INVALID_ARG, "null model");
    if (B < 1 || T < 1) return set_error(UITK_ERR_INVALID_ARG, "B and T must be positive");
    if (m->cfg.center && T <= m->cfg.n_fft / 2) return set_error(UITK_ERR_TOO_SHORT, "clip must be longer than n_fft/2 samples (reflect padding)");
    if (!m->cfg.center && T < m->cfg.n_fft) return set_error(UITK_ERR_TOO_SHORT, "center=False: clip shorter than one n_fft window");
    if (gen_time_patches(m, T) < 1) return set_error(UITK_ERR_TOO_SHORT, "clip shorter than one patch");
    if (T >= (int64_t)1 << 30) return set_error(UITK_ERR_INVALID_ARG, "T too large");
    if (B * (int64_t)gen_num_frames(m, T) >= (int64_t)1 << 31) return set_error(UITK_ERR_INVALID_ARG, "B * frames must be below 2^31 per call");
    return UITK_OK;
}

// floats of work buffers per clip-crop with ntok tokens: X0 | R | Xn | QKV | O | Hd
static size_t gen_row_floats(const GenGeom& g, int ntok) {
    return (size_t)ntok * ((size_t)g.P * g.P + 2 * (size_t)g.D + 3 * (size_t)g.inner + g.inner + g.hidden);
}
constexpr size_t GEN_WORK_BYTES = (size_t)3 << 30;                        // rows are processed in chunks that fit this

static int gen_grow(float** p, size_t* cap, size_t bytes, hipStream_t s) {
    if (bytes <= *cap) return UITK_OK;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;          // growing synchronises: never inside a stream capture
    if (hipStreamIsCapturing(s, &capturing) != hipSuccess) (void)hipGetLastError();
    else if (capturing != hipStreamCaptureStatusNone)
        return set_error(UITK_ERR_WORKSPACE, "general engine: the workspace has to grow for this shape and the stream is being captured (uitk_general_reserve first)");
    GEN_HIP(hipDeviceSynchronize());                                       // (the device may still be reading the old buffer)
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    GEN_HIP(hipMalloc((void**)p, bytes));
    *cap = bytes;
    return UITK_OK;
}

template <int HDP>
static void gen_launch_attention(const float* QKV, float* O, int rows, const GenGeom& g, int ntok, float scale, hipStream_t s) {
    // a workgroup = up to 256 queries of one (clip-crop, head); its keys and values pass through LDS in chunks of CH keys (<= 48 KiB)
    const int CH = std::min(ntok, std::max(1, (48 * 1024) / (2 * g.hd * 4)));
    const int threads = std::min(256, (ntok + 63) / 64 * 64);
    const dim3 grid(rows * g.heads, (ntok + threads - 1) / threads);
    if (g.hd % 4 == 0) hipLaunchKernelGGL((gen_attention_kernel<HDP, true>), grid, dim3(threads), (size_t)2 * CH * g.hd * 4, s, QKV, O, ntok, g.heads, g.inner, g.hd, scale, CH);
    else hipLaunchKernelGGL((gen_attention_kernel<HDP, false>), grid, dim3(threads), (size_t)2 * CH * g.hd * 4, s, QKV, O, ntok, g.heads, g.inner, g.hd, scale, CH);
}

extern "C" {

int uitk_general_create(const uitk_config* cfg, const uitk_weights* w, uitk_general** out) {
    if (!cfg || !w || !out) return set_error(UITK_ERR_INVALID_ARG, "null argument");
    std::string why;
    if (int rc = gen_validate(cfg, &why)) return set_error(rc, why);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return set_error(UITK_ERR_NO_DEVICE, "no HIP device: libuitk has no CPU path");
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return set_error(UITK_ERR_NO_DEVICE, "cannot query HIP device");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return set_error(UITK_ERR_NO_DEVICE, std::string("libuitk is built for gfx950 only, found ") + prop.gcnArchName);
    uitk_general* m = new uitk_general();
    m->cfg = *cfg;
    m->n_cu = prop.multiProcessorCount;
    GenGeom& g = m->g;
    g.D = cfg->embed_dim; g.heads = cfg->num_heads; g.inner = cfg->attention_full ? g.D : g.D / 4; g.hd = g.inner / g.heads;
    g.hidden = cfg->mlp_hidden; g.P = cfg->patch_size; g.GF = cfg->n_mels / g.P; g.gt_model = cfg->target_length / g.P;
    g.M = cfg->n_mels; g.C = cfg->outputdim; g.depth = cfg->depth; g.pooling = cfg->pooling; g.act_gelu = cfg->act_gelu;
    g.cls = cfg->pooling == 2 ? 1 : 0;
    const int N = cfg->n_fft, nb = N / 2 + 1, D = g.D, K = g.P * g.P;
    m->nb = nb;
    m->log2N = 0;
    if ((N & (N - 1)) == 0) { while ((1 << m->log2N) < N) ++m->log2N; }
    // ---- the parameter blob (every array 256-byte aligned)
    std::vector<float> host;
    auto put = [&](const float* src, size_t n) { const size_t o = host.size(); host.resize((o + n + 63) & ~(size_t)63, 0.f); if (src) std::memcpy(host.data() + o, src, n * 4); return o; };
    std::vector<float> wpad(N, 0.f), tw(2 * (size_t)N), bn(4 * (size_t)g.M);
    {
        const int left = (N - cfg->win_size) / 2;                          // torch.stft: the window is centred in n_fft
        for (int i = 0; i < cfg->win_size; ++i) wpad[left + i] = w->window[i];
        for (int j = 0; j < N; ++j) { const double a = 2.0 * M_PI * (double)j / (double)N; tw[2 * j] = (float)std::cos(a); tw[2 * j + 1] = (float)std::sin(a); }
        for (int i = 0; i < g.M; ++i) {
            bn[i] = w->bn_weight[i]; bn[g.M + i] = w->bn_bias[i]; bn[2 * g.M + i] = w->bn_mean[i];
            bn[3 * g.M + i] = 1.0f / std::sqrt(w->bn_var[i] + 1e-5f);
        }
    }
    std::vector<float> band(2 * (size_t)g.M);                               // int pairs, carried in the float blob bit for bit
    for (int mm = 0; mm < g.M; ++mm) {
        int lo = nb, hi = 0;
        for (int k = 0; k < nb; ++k) if (w->mel_fb[(size_t)k * g.M + mm] != 0.0f) { lo = std::min(lo, k); hi = k + 1; }
        if (hi == 0) lo = 0;
        std::memcpy(&band[2 * mm], &lo, 4); std::memcpy(&band[2 * mm + 1], &hi, 4);
    }
    const size_t o_band = put(band.data(), band.size());
    const size_t o_wpad = put(wpad.data(), N), o_tw = put(tw.data(), 2 * (size_t)N), o_fb = put(w->mel_fb, (size_t)nb * g.M), o_bn = put(bn.data(), bn.size());
    const size_t o_pe_w = put(w->pe_w, (size_t)D * K), o_pe_b = put(w->pe_b, D), o_tp = put(w->time_pos, (size_t)D * g.gt_model), o_fp = put(w->freq_pos, (size_t)D * g.GF);
    if (g.cls && (!w->cls_token || !w->token_pos)) { delete m; return set_error(UITK_ERR_INVALID_ARG, "pooling 'token' needs cls_token and token_pos"); }
    const size_t o_cls = put(g.cls ? w->cls_token : nullptr, D), o_tokp = put(g.cls ? w->token_pos : nullptr, D);
    const size_t o_nw = put(w->norm_w, D), o_nb = put(w->norm_b, D), o_hw = put(w->head_ln_w, D), o_hb = put(w->head_ln_b, D);
    const size_t o_ow = put(w->head_w, (size_t)g.C * D), o_ob = put(w->head_b, g.C);
    std::vector<size_t> ob((size_t)g.depth * 12);
    for (int b = 0; b < g.depth; ++b) {
        const uitk_block_weights& bw = w->blocks[b];
        size_t* o = ob.data() + (size_t)b * 12;
        o[0] = put(bw.norm1_w, D); o[1] = put(bw.norm1_b, D);
        o[2] = put(bw.qkv_w, (size_t)3 * g.inner * D); o[3] = put(bw.qkv_b, 3 * (size_t)g.inner);
        o[4] = put(bw.proj_w, (size_t)D * g.inner); o[5] = put(bw.proj_b, D);
        o[6] = put(bw.norm2_w, D); o[7] = put(bw.norm2_b, D);
        o[8] = put(bw.fc1_w, (size_t)g.hidden * D); o[9] = put(bw.fc1_b, g.hidden);
        o[10] = put(bw.fc2_w, (size_t)D * g.hidden); o[11] = put(bw.fc2_b, D);
    }
    if (hipMalloc((void**)&m->params, host.size() * 4) != hipSuccess || hipMalloc((void**)&m->amax, AMAX_SLOTS * 4) != hipSuccess ||
        hipMemcpy(m->params, host.data(), host.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        uitk_general_destroy(m);
        return set_error(UITK_ERR_HIP, "general engine: parameter upload failed");
    }
    const float* p = m->params;
    m->band = (const int2*)(p + o_band);
    m->wpad = p + o_wpad; m->tw = (const float2*)(p + o_tw); m->fb = p + o_fb; m->bn = p + o_bn; m->pe_w = p + o_pe_w; m->pe_b = p + o_pe_b;
    m->time_pos = p + o_tp; m->freq_pos = p + o_fp; m->cls = p + o_cls; m->token_pos = p + o_tokp; m->norm_w = p + o_nw; m->norm_b = p + o_nb;
    m->hln_w = p + o_hw; m->hln_b = p + o_hb; m->head_w = p + o_ow; m->head_b = p + o_ob;
    m->blocks.resize(g.depth);
    for (int b = 0; b < g.depth; ++b) {
        const size_t* o = ob.data() + (size_t)b * 12;
        m->blocks[b] = GenBlockPtrs{p + o[0], p + o[1], p + o[2], p + o[3], p + o[4], p + o[5], p + o[6], p + o[7], p + o[8], p + o[9], p + o[10], p + o[11]};
    }
    float armed[AMAX_SLOTS];
    for (float& v : armed) v = -INFINITY;
    (void)hipMemcpy(m->amax, armed, sizeof armed, hipMemcpyHostToDevice);
    *out = m;
    return UITK_OK;
}

int uitk_general_destroy(uitk_general* m) {
    if (!m) return UITK_OK;
    if (m->params) (void)hipFree(m->params);
    if (m->amax) (void)hipFree(m->amax);
    if (m->frames) (void)hipFree(m->frames);
    if (m->work) (void)hipFree(m->work);
    if (m->rowprobs) (void)hipFree(m->rowprobs);
    delete m;
    return UITK_OK;
}

int uitk_general_num_frames(const uitk_general* m, int64_t T) { return m ? gen_num_frames(m, T) : -1; }
int uitk_general_num_crops(const uitk_general* m, int64_t T) { return m ? gen_num_crops(m, T) : -1; }
float* uitk_general_amax_ptr(uitk_general* m) { return m ? m->amax : nullptr; }

static int gen_reserve(uitk_general* m, int64_t B, int64_t T, hipStream_t s) {
    if (int rc = gen_check_shape(m, B, T)) return rc;
    const int nF = gen_num_frames(m, T), nc = gen_num_crops(m, T), nt = gen_time_patches(m, T);
    const int ntok = m->g.GF * nt + m->g.cls;
    if (int rc = gen_grow(&m->frames, &m->frames_cap, (size_t)B * nF * m->g.M * 4, s)) return rc;
    const int64_t rows = B * nc;
    const size_t per_row = gen_row_floats(m->g, ntok) * 4;
    int64_t chunk = (int64_t)std::max<size_t>(1, GEN_WORK_BYTES / per_row);
    chunk = std::min<int64_t>(chunk, rows);
    chunk = std::min<int64_t>(chunk, (int64_t)65535 * 64 / ntok);          // the dense kernel's grid: 64 token rows per workgroup in gridDim.y
    if (chunk < 1) return set_error(UITK_ERR_UNSUPPORTED, "general engine: one clip-crop exceeds the work buffer");
    if (int rc = gen_grow(&m->work, &m->work_cap, (size_t)chunk * per_row, s)) return rc;
    m->work_rows = (int64_t)(m->work_cap / per_row); m->work_ntok = ntok;
    if (int rc = gen_grow(&m->rowprobs, &m->rowprobs_cap, nc > 1 ? (size_t)rows * m->g.C * 4 : 0, s)) return rc;
    return UITK_OK;
}

int uitk_general_reserve(uitk_general* m, int64_t B, int64_t T) { return gen_reserve(m, B, T, nullptr); }

static int gen_run_frontend(uitk_general* m, const void* pcm, int in_dtype, int64_t B, int64_t T, hipStream_t s) {
    GenFrontArgs a{};
    a.pcm = pcm; a.i16 = in_dtype == UITK_IN_I16; a.B = B; a.T = T; a.nF = gen_num_frames(m, T); a.hop = m->cfg.hop_size;
    a.pad = m->cfg.center ? m->cfg.n_fft / 2 : 0; a.N = m->cfg.n_fft; a.nb = m->nb; a.M = m->g.M; a.log2N = m->log2N;
    a.wpad = m->wpad; a.tw = m->tw; a.fb = m->fb; a.band = m->band; a.frames = m->frames; a.amax = m->amax;
    hipLaunchKernelGGL(gen_arm_kernel, dim3(1), dim3(64), 0, s, m->amax);
    const int64_t total = B * (int64_t)a.nF;
    const int grid = (int)std::min<int64_t>(total, (int64_t)m->n_cu * 8);
    const size_t lds = (size_t)a.N * 8 + (size_t)((a.nb + 1) & ~1) * 4 + (a.log2N ? 0 : (size_t)a.N * 8);
    hipLaunchKernelGGL(gen_frontend_kernel, dim3(grid), dim3(256), lds, s, a);
    GEN_HIP(hipGetLastError());
    return UITK_OK;
}

int uitk_general_frontend(uitk_general* m, const void* pcm, int in_dtype, int64_t B, int64_t T, void* stream) {
    if (int rc = gen_check_shape(m, B, T)) return rc;
    if (!pcm) return set_error(UITK_ERR_INVALID_ARG, "null pcm");
    if (in_dtype != UITK_IN_F32 && in_dtype != UITK_IN_I16) return set_error(UITK_ERR_INVALID_ARG, "unknown in_dtype");
    if (int rc = gen_reserve(m, B, T, (hipStream_t)stream)) return rc;
    if (int rc = gen_run_frontend(m, pcm, in_dtype, B, T, (hipStream_t)stream)) return rc;
    m->fe_B = B; m->fe_T = T;
    return UITK_OK;
}

int uitk_general_encode(uitk_general* m, int64_t B, int64_t T, float* probs, float* logits, void* stream) {
    if (int rc = gen_check_shape(m, B, T)) return rc;
    if (!probs) return set_error(UITK_ERR_INVALID_ARG, "null probs");
    if (m->fe_B != B || m->fe_T != T) return set_error(UITK_ERR_INVALID_ARG, "uitk_general_encode: no front-end output of this shape (uitk_general_frontend first)");
    if (logits && m->g.pooling == 1) return set_error(UITK_ERR_INVALID_ARG, "pooling 'dm' has no pre-sigmoid output");
    hipStream_t s = (hipStream_t)stream;
    GenGeom g = m->g;
    const int nF = gen_num_frames(m, T), nc = gen_num_crops(m, T);
    g.nt = gen_time_patches(m, T); g.np = g.GF * g.nt; g.ntok = g.np + g.cls;
    if (g.ntok != m->work_ntok || m->work_rows < 1) return set_error(UITK_ERR_WORKSPACE, "general engine: no workspace for this shape");
    const int64_t rows = B * nc;
    float* rowprobs = nc > 1 ? m->rowprobs : probs;
    const int K = g.P * g.P, D = g.D;
    // the reference's scale is (dim // num_heads) ** -0.5 for BOTH attention classes (models/uit.py:99-100, 138-139)
    const float scale = 1.0f / std::sqrt((float)(D / g.heads));
    for (int64_t r0 = 0; r0 < rows; r0 += m->work_rows) {
        const int n = (int)std::min<int64_t>(m->work_rows, rows - r0);
        const int64_t Mr = (int64_t)n * g.ntok;
        float* X0 = m->work;
        float* R = X0 + Mr * K;
        float* Xn = R + Mr * D;
        float* QKV = Xn + Mr * D;
        float* O = QKV + Mr * 3 * g.inner;
        float* Hd = O + Mr * g.inner;
        hipLaunchKernelGGL(gen_patch_kernel, dim3(2048), dim3(256), 0, s, m->frames, m->amax, m->cfg.top_db, m->bn, g, nF, nc, m->cfg.target_length, r0, n, X0);
        hipLaunchKernelGGL(gen_init_stream_kernel, dim3(1024), dim3(256), 0, s, m->pe_b, m->time_pos, m->freq_pos, m->cls, m->token_pos, g, n, R);
        GEN_HIP(hipGetLastError());
        GEN_HIP(launch_naive_linear(X0, m->pe_w, nullptr, R, (int)Mr, K, D, 0, 1, s));
        const int lnb = (int)((Mr + 3) / 4);
        for (int b = 0; b < g.depth; ++b) {
            const GenBlockPtrs& w = m->blocks[b];
            hipLaunchKernelGGL(gen_layernorm_kernel, dim3(lnb), dim3(256), 0, s, R, Xn, w.n1w, w.n1b, Mr, D, 1e-6f);
            GEN_HIP(launch_naive_linear(Xn, w.qkv_w, w.qkv_b, QKV, (int)Mr, D, 3 * g.inner, 0, 0, s));
            if (g.hd <= 8) gen_launch_attention<8>(QKV, O, n, g, g.ntok, scale, s);
            else if (g.hd <= 16) gen_launch_attention<16>(QKV, O, n, g, g.ntok, scale, s);
            else if (g.hd <= 32) gen_launch_attention<32>(QKV, O, n, g, g.ntok, scale, s);
            else if (g.hd <= 64) gen_launch_attention<64>(QKV, O, n, g, g.ntok, scale, s);
            else if (g.hd <= 128) gen_launch_attention<128>(QKV, O, n, g, g.ntok, scale, s);
            else hipLaunchKernelGGL(gen_attention_wide_kernel, dim3(n * g.heads, (g.ntok + 63) / 64, (g.hd + 63) / 64), dim3(64), 0, s, QKV, O, g.ntok, g.heads, g.inner, g.hd, scale);
            GEN_HIP(hipGetLastError());
            GEN_HIP(launch_naive_linear(O, w.proj_w, w.proj_b, R, (int)Mr, g.inner, D, 0, 1, s));
            hipLaunchKernelGGL(gen_layernorm_kernel, dim3(lnb), dim3(256), 0, s, R, Xn, w.n2w, w.n2b, Mr, D, 1e-6f);
            GEN_HIP(launch_naive_linear(Xn, w.fc1_w, w.fc1_b, Hd, (int)Mr, D, g.hidden, g.act_gelu ? 2 : 1, 0, s));
            GEN_HIP(launch_naive_linear(Hd, w.fc2_w, w.fc2_b, R, (int)Mr, g.hidden, D, 0, 1, s));
        }
        hipLaunchKernelGGL(gen_layernorm_kernel, dim3(lnb), dim3(256), 0, s, R, Xn, m->norm_w, m->norm_b, Mr, D, 1e-6f);
        GenHeadArgs h{m->hln_w, m->hln_b, m->head_w, m->head_b, rowprobs + r0 * g.C, logits ? logits + r0 * g.C : nullptr};
        hipLaunchKernelGGL(gen_head_kernel, dim3(n), dim3(256), (size_t)D * 4, s, Xn, h, g);
        GEN_HIP(hipGetLastError());
    }
    if (nc > 1) GEN_HIP(launch_crop_reduce(rowprobs, probs, B, nc, g.C, m->cfg.eval_avg, s));
    return UITK_OK;
}

int uitk_general_forward(uitk_general* m, const void* pcm, int in_dtype, int64_t B, int64_t T, float* probs, float* logits, void* stream) {
    if (int rc = uitk_general_frontend(m, pcm, in_dtype, B, T, stream)) return rc;
    return uitk_general_encode(m, B, T, probs, logits, stream);
}

int uitk_general_mel_db(uitk_general* m, const void* pcm, int in_dtype, int64_t B, int64_t T, float* out, void* stream) {
    if (!out) return set_error(UITK_ERR_INVALID_ARG, "null out");
    if (int rc = uitk_general_frontend(m, pcm, in_dtype, B, T, stream)) return rc;
    hipLaunchKernelGGL(gen_mel_export_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, m->frames, m->amax, m->cfg.top_db, B, gen_num_frames(m, T), m->g.M, out);
    GEN_HIP(hipGetLastError());
    return UITK_OK;
}

}  // extern "C"
