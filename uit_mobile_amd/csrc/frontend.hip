// frontend.hip -- fused log-mel front-end for gfx950 (MI355X).
//
// Replaces self.front_end(x) of the reference (models/uit.py:298-308, call :455): torchaudio 0.13
// MelSpectrogram(sr 16k, n_fft = win = 512, hop 160, center/reflect, power 2, 64 HTK mels 0..8 kHz) followed by
// AmplitudeToDB('power', top_db=120).  One kernel does reflect-pad indexing + window + 512-point real FFT +
// |X|^2 + banded mel accumulate + 10*log10(max(.,1e-10)) and the batch amax; the top_db clamp itself
// (max(x, amax-120)) needs the amax of the WHOLE call, so it is applied by the consumer (encoder prologue).
//
// Mapping (wave64, no workgroup barriers -- every wave is independent):
//   * one wave = 4 consecutive frames (a "quad") of the flattened [B*nF] frame list; 16 lanes per frame.
//   * the 512 real samples of a frame are packed as 256 complex z[m] = x[2m] + i x[2m+1]; the 256-point
//     complex FFT is a 16x16 Cooley-Tukey: lane p holds z[16*m1 + p] (m1 = 0..15, 32 VGPRs), does a
//     16-point FFT in registers, multiplies by W256^(p*k1), transposes through a conflict-free padded LDS
//     tile (row stride 17 complex), does the second 16-point FFT, and the real-FFT split
//     X[k] = ((Z[k]+conj Z[256-k]) - i w^k (Z[k]-conj Z[256-k]))/2 pairs bins k and 256-k through LDS.
//   * window, both twiddle sets are per-lane constants held in VGPRs for the whole (persistent) kernel.
//   * mel: lane p owns mels p, p+16, p+32, p+48 (one from each quarter of the band-width range, so the
//     500-nonzero banded filterbank is balanced across lanes); power spectrum read from LDS.
//   * HBM traffic: the PCM is read once from HBM (frames overlap 3.2x, absorbed by L1/L2); output is
//     mel[B][nF][64] fp32 (frame-major so the encoder's patch gather reads 32-byte runs).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "uitk_dev.h"

namespace uitk {

struct c32 { float x, y; };
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c32 cmul(c32 a, c32 w) { return {a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
__device__ __forceinline__ c32 mul_mi(c32 a) { return {a.y, -a.x}; }   // a * (-i)

// forward 4-point DFT (W4 = -i) in place
__device__ __forceinline__ void dft4(c32& a, c32& b, c32& c, c32& d) {
    c32 s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = mul_mi(csub(b, d));
    a = cadd(s0, s2); c = csub(s0, s2);
    b = cadd(s1, s3); d = csub(s1, s3);
}

// forward 16-point DFT, natural order in -> natural order out:  Y[k] = sum_n v[n] e^{-2 pi i n k / 16}
// n = 4 n1 + n2, k = k1 + 4 k2:  Y = DFT4_{n2}( W16^{n2 k1} * DFT4_{n1}(v) )
__device__ __forceinline__ void fft16(c32 (&v)[16]) {
    const float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);   // v[4*k1 + n2] = t[k1][n2]
    // twiddle t[k1][n2] *= W16^(n2*k1)
    v[4 + 1] = cmul(v[4 + 1], {C1, -S1});   // W^1
    v[4 + 2] = cmul(v[4 + 2], {H, -H});     // W^2
    v[4 + 3] = cmul(v[4 + 3], {S1, -C1});   // W^3
    v[8 + 1] = cmul(v[8 + 1], {H, -H});     // W^2
    v[8 + 2] = mul_mi(v[8 + 2]);            // W^4 = -i
    v[8 + 3] = cmul(v[8 + 3], {-H, -H});    // W^6
    v[12 + 1] = cmul(v[12 + 1], {S1, -C1}); // W^3
    v[12 + 2] = cmul(v[12 + 2], {-H, -H});  // W^6
    v[12 + 3] = cmul(v[12 + 3], {-C1, S1}); // W^9
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4(v[4 * k1 + 0], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
    // now v[4*k1 + k2] = Y[k1 + 4*k2]  -> reorder to natural
    c32 t[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) t[k1 + 4 * k2] = v[4 * k1 + k2];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = t[i];
}

constexpr int FE_WAVES = 4;
constexpr int FE_ROW = 17;                          // padded row stride (complex) of the transpose tile
constexpr int FE_WAVE_LDS = 4 * 16 * FE_ROW * 8;     // 8704 B per wave (also holds Z[4][256] and P[4][260])

template <bool I16>
__device__ __forceinline__ float load_sample(const void* pcm, int64_t row_off, int64_t T, int s) {
    if (s < 0) s = -s;                               // reflect, edge sample not repeated (torch 'reflect')
    if (s >= T) s = 2 * ((int)T - 1) - s;
    if (I16) return (float)((const int16_t*)pcm)[row_off + s] * (1.0f / 32768.0f);
    return ((const float*)pcm)[row_off + s];
}

template <bool I16>
__global__ __launch_bounds__(FE_WAVES * 64) void mel_frontend_kernel(FrontendArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: [mel weights (nnz floats, padded)] [per-wave scratch]
    float* melw = (float*)smem;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, p = lane & 15;
    for (int i = tid; i < a.mel_nnz; i += FE_WAVES * 64) melw[i] = a.mel_w[i];
    char* wbase = smem + a.melw_lds_bytes + wave * FE_WAVE_LDS;
    c32* E = (c32*)wbase;                            // [4][16][17] transpose tile / [4][256] Z
    float* P = (float*)wbase;                        // [4][260] power (aliases E, phases separated)
    __syncthreads();

    // per-lane constants
    float win[32];
#pragma unroll
    for (int m1 = 0; m1 < 16; ++m1) {
        win[2 * m1] = a.window[32 * m1 + 2 * p];
        win[2 * m1 + 1] = a.window[32 * m1 + 2 * p + 1];
    }
    c32 tw[16];                                      // W256^(p*k1)
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) { float2 t = a.tw256[p * 16 + k1]; tw[k1] = {t.x, t.y}; }
    float2 pw[8];                                    // (cos, sin)(2 pi k / 512), k = p + 16 j
#pragma unroll
    for (int j = 0; j < 8; ++j) pw[j] = a.tw512[p + 16 * j];
    int mlo[4], mn[4], moff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = p + 16 * i;
        mlo[i] = a.mel_lo[m]; mn[i] = a.mel_n[m]; moff[i] = a.mel_off[m];
    }

    const int64_t total = a.B * (int64_t)a.nF;
    const int64_t nquads = (total + 3) >> 2;
    float lmax = -INFINITY;
    for (int64_t quad = (int64_t)blockIdx.x * FE_WAVES + wave; quad < nquads; quad += (int64_t)gridDim.x * FE_WAVES) {
        int64_t fid = quad * 4 + q;
        const bool live = fid < total;
        if (!live) fid = total - 1;
        const int64_t b = fid / a.nF;
        const int f = (int)(fid - b * a.nF);
        const int64_t row_off = b * a.T;
        const int s0 = f * a.hop - 256 + 2 * p;
        c32 v[16];
        if (s0 >= 0 && s0 + 32 * 15 + 1 < a.T) {     // interior frame: no reflection
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) {
                float x0, x1;
                if (I16) {
                    const int16_t* src = (const int16_t*)a.pcm + row_off + s0 + 32 * m1;
                    x0 = (float)src[0] * (1.0f / 32768.0f); x1 = (float)src[1] * (1.0f / 32768.0f);
                } else {
                    const float* src = (const float*)a.pcm + row_off + s0 + 32 * m1;
                    x0 = src[0]; x1 = src[1];
                }
                v[m1] = {x0 * win[2 * m1], x1 * win[2 * m1 + 1]};
            }
        } else {
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) {
                float x0 = load_sample<I16>(a.pcm, row_off, a.T, s0 + 32 * m1);
                float x1 = load_sample<I16>(a.pcm, row_off, a.T, s0 + 32 * m1 + 1);
                v[m1] = {x0 * win[2 * m1], x1 * win[2 * m1 + 1]};
            }
        }
        fft16(v);                                    // over m1 -> index k1, for m2 = p
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) v[k1] = cmul(v[k1], tw[k1]);
        // transpose: E[q][k1][m2 = p]  ->  lane k1 = p reads E[q][p][m2]
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) E[(q * 16 + k1) * FE_ROW + p] = v[k1];
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int m2 = 0; m2 < 16; ++m2) v[m2] = E[(q * 16 + p) * FE_ROW + m2];
        fft16(v);                                    // over m2 -> Z[k1 + 16 k2], k1 = p, k2 = index
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        c32* Z = E;                                  // natural order Z[q][k]
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) Z[q * 256 + 16 * k2 + p] = v[k2];
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        c32 za[8], zb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int k = p + 16 * j;
            za[j] = Z[q * 256 + k];
            zb[j] = Z[q * 256 + ((256 - k) & 255)];
        }
        c32 z128 = Z[q * 256 + 128];
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        // power spectrum of bins k and 256-k
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int k = p + 16 * j;
            float Ar = za[j].x + zb[j].x, Ai = za[j].y - zb[j].y;
            float Br = za[j].x - zb[j].x, Bi = za[j].y + zb[j].y;
            float c = pw[j].x, s = pw[j].y;
            float Tr = -s * Br + c * Bi, Ti = -s * Bi - c * Br;
            float xr = 0.5f * (Ar + Tr), xi = 0.5f * (Ai + Ti);
            float yr = 0.5f * (Ar - Tr), yi = 0.5f * (Ti - Ai);
            P[q * 260 + k] = xr * xr + xi * xi;
            P[q * 260 + 256 - k] = yr * yr + yi * yi;      // k = 0 writes bin 256 (Nyquist)
        }
        if (p == 0) P[q * 260 + 128] = z128.x * z128.x + z128.y * z128.y;
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        // banded mel + dB
        float* orow = a.mel + fid * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float acc = 0.f;
            const float* pp = P + q * 260 + mlo[i];
            const float* ww = melw + moff[i];
            for (int t = 0; t < mn[i]; ++t) acc += pp[t] * ww[t];
            float db = 10.0f * __log10f(fmaxf(acc, 1e-10f));
            if (live) { orow[p + 16 * i] = db; lmax = fmaxf(lmax, db); }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    }
    // batch amax: wave reduce, then one atomic per wave into one of UITK_AMAX_SLOTS slots.  Slots are
    // initialised to 0xFFFFFFFF (hipMemsetAsync 0xFF): signed-max for v >= 0, unsigned-min for v < 0.
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if (lane == 0 && lmax > -INFINITY) {
        float* slot = a.amax + (blockIdx.x & (AMAX_SLOTS - 1));
        if (lmax >= 0.f) atomicMax((int*)slot, __float_as_int(lmax));
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

hipError_t launch_frontend(const FrontendArgs& a, int in_i16, int n_cu, hipStream_t stream) {
    const int64_t total = a.B * (int64_t)a.nF;
    const int64_t nquads = (total + 3) / 4;
    int64_t want = (nquads + FE_WAVES - 1) / FE_WAVES;
    int grid = (int)(want < (int64_t)n_cu * 4 ? want : (int64_t)n_cu * 4);
    if (grid < 1) grid = 1;
    size_t lds = a.melw_lds_bytes + FE_WAVES * FE_WAVE_LDS;
    if (in_i16) hipLaunchKernelGGL(mel_frontend_kernel<true>, dim3(grid), dim3(FE_WAVES * 64), lds, stream, a);
    else hipLaunchKernelGGL(mel_frontend_kernel<false>, dim3(grid), dim3(FE_WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_mel_export(const float* mel, const float* amax, float top_db, int64_t B, int nF, float* out,
                             hipStream_t stream) {
    hipLaunchKernelGGL(mel_export_kernel, dim3(512), dim3(256), 0, stream, mel, amax, top_db, B, nF, out);
    return hipGetLastError();
}

}  // namespace uitk
