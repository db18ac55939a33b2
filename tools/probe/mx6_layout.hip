// Probe (GPU box): fp6 (e2m3) / fp4 (e2m1) operands of v_mfma_scale_f32_16x16x128_f8f6f4 and the gfx950 conversions that produce
// them (v_cvt_scalef32_pk32_fp6_f16, v_cvt_scalef32_pk_fp4_f16) -- checked with exactly representable data before the encoder's
// low-order weight term (precision f16l6) relies on them.  Also prices the instructions (cycles per instruction, one SIMD).
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe/mx6_layout tools/probe/mx6_layout.hip && tools/probe/mx6_layout
// Hypotheses checked (exit code 0 = all hold):
//   H1  cvt_pk32_fp6_f16: source half i of the 16-VGPR tuple -> 6-bit field at bits [6i, 6i+6) of the 192-bit result; value = x / scale
//   H2  cvt_pk_fp4_f16(old, {a,b}, scale, sel): byte `sel` of old <- enc(a) | enc(b) << 4
//   H3  MFMA fp6 x fp6: D[m][n] = sum_g 2^(sa[m,g]-127) 2^(sb[n,g]-127) sum_i A(m,g)[i] B(n,g)[i]  -- position-wise inside a lane group,
//       lane (row, g) supplies the E8M0 scale of ITS OWN 32 values (cbsz = blgp = 2)
//   H4  MFMA fp6 x fp4 (cbsz 2, blgp 4): the same with 4-bit fields at bits [4i, 4i+4) of a 4-VGPR operand
//   H5  operand produced in-kernel by the conversions multiplies as the host model says (conversion order == MFMA order)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned u6 __attribute__((ext_vector_type(6)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v8i pad8(u6 x) { return v8i{(int)x[0], (int)x[1], (int)x[2], (int)x[3], (int)x[4], (int)x[5], 0, 0}; }

// mode 0: A, B raw fp6 (6 dwords per lane) from memory.  mode 1: A raw fp6, B raw fp4 (4 dwords).  mode 2: A raw fp6, B converted
// in-kernel from 32 halves per lane (fp6, scale 2 -> E8M0 128).  mode 3: A raw fp6, B converted in-kernel to fp4 (scale 1).
__global__ void mm(const unsigned* A, const unsigned* B, const _Float16* Bh, const int* sa, const int* sb, float* C, unsigned* raw, int mode, int opsel) {
    const int l = threadIdx.x;
    u6 a;
    for (int i = 0; i < 6; ++i) a[i] = A[l * 6 + i];
    f4 c = {0.f, 0.f, 0.f, 0.f};
    int scale_a = sa[l], scale_b = sb[l];
    if (mode == 0) {
        u6 b;
        for (int i = 0; i < 6; ++i) b[i] = B[l * 6 + i];
        if (opsel) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pad8(a), pad8(b), c, 2, 2, 2, (scale_a << 16) | 0x11007733, 1, (scale_b << 8) | 0x22440055);
        else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pad8(a), pad8(b), c, 2, 2, 0, scale_a, 0, scale_b);
    } else if (mode == 1) {
        v8i b = {(int)B[l * 4], (int)B[l * 4 + 1], (int)B[l * 4 + 2], (int)B[l * 4 + 3], 0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pad8(a), b, c, 2, 4, 0, scale_a, 0, scale_b);
    } else if (mode == 2) {
        h32 x;
        for (int i = 0; i < 32; ++i) x[i] = Bh[l * 32 + i];
        const u6 b = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x, 2.0f);
        for (int i = 0; i < 6; ++i) raw[l * 6 + i] = b[i];
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pad8(a), pad8(b), c, 2, 2, 0, scale_a, 0, 128);
    } else {
        unsigned w[4] = {0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
        for (int d = 0; d < 4; ++d) {
            h2 p0 = {Bh[l * 32 + 8 * d + 0], Bh[l * 32 + 8 * d + 1]}, p1 = {Bh[l * 32 + 8 * d + 2], Bh[l * 32 + 8 * d + 3]};
            h2 p2 = {Bh[l * 32 + 8 * d + 4], Bh[l * 32 + 8 * d + 5]}, p3 = {Bh[l * 32 + 8 * d + 6], Bh[l * 32 + 8 * d + 7]};
            w[d] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w[d], p0, 1.0f, 0);
            w[d] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w[d], p1, 1.0f, 1);
            w[d] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w[d], p2, 1.0f, 2);
            w[d] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w[d], p3, 1.0f, 3);
            raw[l * 6 + d] = w[d];
        }
        v8i b = {(int)w[0], (int)w[1], (int)w[2], (int)w[3], 0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pad8(a), b, c, 2, 4, 0, scale_a, 0, 127);
    }
    const int r = l & 15, g = l >> 4;
    for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = c[i];
}

// ---- instruction pricing: one workgroup, WAVES waves (4 = one per SIMD, 8 = two per SIMD), s_memtime around N instructions
template <int KIND>
__global__ void rate(unsigned long long* out, float* sink, int n) {
    h8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    v8i a6 = {0x11111111, 0x22222222, 0x33333333, 0x11111111, 0x22222222, 0x33333333, 0, 0}, b6 = a6;
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    h32 x;
    for (int i = 0; i < 32; ++i) x[i] = (_Float16)(float)(threadIdx.x + i);
    u6 y = {0, 0, 0, 0, 0, 0};
    unsigned w = 0;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
        if constexpr (KIND == 0) {            // f16 16x16x32
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
        } else if constexpr (KIND == 1) {     // fp6 x fp6 16x16x128
            c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c0, 2, 2, 0, 127, 0, 127); c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c1, 2, 2, 0, 127, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c2, 2, 2, 0, 127, 0, 127); c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c3, 2, 2, 0, 127, 0, 127);
        } else if constexpr (KIND == 2) {     // fp6 x fp4
            c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c0, 2, 4, 0, 127, 0, 127); c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c1, 2, 4, 0, 127, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c2, 2, 4, 0, 127, 0, 127); c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c3, 2, 4, 0, 127, 0, 127);
        } else if constexpr (KIND == 3) {     // fp8 x fp8 (reference: twice the f16 cycles)
            c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c0, 0, 0, 0, 127, 0, 127); c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c1, 0, 0, 0, 127, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c2, 0, 0, 0, 127, 0, 127); c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c3, 0, 0, 0, 127, 0, 127);
        } else if constexpr (KIND == 4) {     // 4 f16 MFMAs + 1 fp6 MFMA: the hi + lo pattern of one K = 128 accumulator
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c0, 2, 2, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c1, 2, 2, 0, 127, 0, 127);
        } else if constexpr (KIND == 5) {     // cvt_pk32_fp6_f16, independent
            y = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x, 2.0f);
            asm volatile("" : "+v"(y));
            u6 y2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x, 4.0f);
            asm volatile("" : "+v"(y2));
            u6 y3 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x, 8.0f);
            asm volatile("" : "+v"(y3));
            u6 y4 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x, 1.0f);
            asm volatile("" : "+v"(y4));
        } else if constexpr (KIND == 6) {     // cvt_pk_fp4_f16 x 4 (one dword of 8 values), dependent through the destination
            h2 p = {x[0], x[1]};
            w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, p, 1.0f, 0); w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, p, 1.0f, 1);
            w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, p, 1.0f, 2); w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, p, 1.0f, 3);
            asm volatile("" : "+v"(w));
        } else if constexpr (KIND == 7) {     // one f16 MFMA + one cvt_pk32 per iteration x 4: does the conversion hide in the MFMA shadow?
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            y = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x, 2.0f); asm volatile("" : "+v"(y));
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
            y = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x, 2.0f); asm volatile("" : "+v"(y));
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
            y = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x, 2.0f); asm volatile("" : "+v"(y));
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
            y = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x, 2.0f); asm volatile("" : "+v"(y));
        } else {                              // KIND 8: one f16 MFMA + two cvt_pk_fp4 per iteration x 4
            h2 p = {x[0], x[1]};
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, p, 1.0f, 0); w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, p, 1.0f, 1);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
            w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, p, 1.0f, 2); w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, p, 1.0f, 3);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
            w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, p, 1.0f, 0); w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, p, 1.0f, 1);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
            w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, p, 1.0f, 2); w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, p, 1.0f, 3);
            asm volatile("" : "+v"(w));
        }
    }
    asm volatile("s_nop 0" ::"v"(c0), "v"(c1), "v"(c2), "v"(c3));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
    sink[threadIdx.x] = c0[0] + c1[0] + c2[0] + c3[0] + (float)y[0] + (float)w;
}

// e2m3: s eem mm : e = 0 -> m / 8; else 2^(e-1) (1 + m / 8).   e2m1: s ee m : e = 0 -> m / 2; else 2^(e-1) (1 + m / 2)
static double dec6(unsigned v) { const int s = v >> 5 & 1, e = v >> 3 & 3, m = v & 7; const double x = e == 0 ? m / 8.0 : std::ldexp(1.0 + m / 8.0, e - 1); return s ? -x : x; }
static double dec4(unsigned v) { const int s = v >> 3 & 1, e = v >> 1 & 3, m = v & 1; const double x = e == 0 ? m / 2.0 : std::ldexp(1.0 + m / 2.0, e - 1); return s ? -x : x; }
static unsigned field(const unsigned* w, int bit, int n) {
    unsigned long long lo = w[bit >> 5];
    if ((bit & 31) + n > 32) lo |= (unsigned long long)w[(bit >> 5) + 1] << 32;
    return (unsigned)(lo >> (bit & 31)) & ((1u << n) - 1);
}
static void put(unsigned* w, int bit, int n, unsigned v) {
    for (int i = 0; i < n; ++i) if (v >> i & 1) w[(bit + i) >> 5] |= 1u << ((bit + i) & 31);
}

int main() {
    unsigned hA[64 * 6] = {}, hB6[64 * 6] = {}, hB4[64 * 4] = {}, hraw[64 * 6];
    double fA[64][32], fB6[64][32], fB4[64][32], fBh6[64][32], fBh4[64][32];
    _Float16 hBh6[64 * 32], hBh4[64 * 32];
    int hsa[64], hsb[64];
    srand(3);
    for (int l = 0; l < 64; ++l) {
        for (int i = 0; i < 32; ++i) {
            unsigned a = rand() & 63, b = rand() & 63, c = rand() & 15;
            put(hA + l * 6, 6 * i, 6, a); fA[l][i] = dec6(a);
            put(hB6 + l * 6, 6 * i, 6, b); fB6[l][i] = dec6(b);
            put(hB4 + l * 4, 4 * i, 4, c); fB4[l][i] = dec4(c);
            // half inputs: exactly representable after the conversion (fp6 at scale 2: 2 x grid; fp4 at scale 1: the grid)
            const unsigned d = rand() & 63, e = rand() & 15;
            fBh6[l][i] = 2.0 * dec6(d); hBh6[l * 32 + i] = (_Float16)(float)fBh6[l][i];
            fBh4[l][i] = dec4(e); hBh4[l * 32 + i] = (_Float16)(float)fBh4[l][i];
        }
        hsa[l] = 126 + rand() % 3; hsb[l] = 126 + rand() % 3;
    }
    unsigned *dA, *dB6, *dB4, *draw; _Float16 *dBh6, *dBh4; int *dsa, *dsb; float* dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB6, sizeof hB6); hipMalloc(&dB4, sizeof hB4); hipMalloc(&draw, sizeof hraw);
    hipMalloc(&dBh6, sizeof hBh6); hipMalloc(&dBh4, sizeof hBh4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dC, 1024);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB6, hB6, sizeof hB6, hipMemcpyHostToDevice);
    hipMemcpy(dB4, hB4, sizeof hB4, hipMemcpyHostToDevice); hipMemcpy(dBh6, hBh6, sizeof hBh6, hipMemcpyHostToDevice);
    hipMemcpy(dBh4, hBh4, sizeof hBh4, hipMemcpyHostToDevice); hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    int rc = 0;
    for (int mode = 0; mode < 5; ++mode) {
        const int m = mode == 4 ? 0 : mode, opsel = mode == 4;
        hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, dA, m == 1 ? dB4 : dB6, m == 3 ? dBh4 : dBh6, dsa, dsb, dC, draw, m, opsel);
        float hC[256];
        hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost); hipMemcpy(hraw, draw, sizeof hraw, hipMemcpyDeviceToHost);
        double maxd = 0, maxr = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double acc = 0;
                for (int g = 0; g < 4; ++g) {
                    double s = 0;
                    for (int k = 0; k < 32; ++k) s += fA[16 * g + i][k] * (m == 0 ? fB6[16 * g + j][k] : m == 1 ? fB4[16 * g + j][k] : m == 2 ? fBh6[16 * g + j][k] : fBh4[16 * g + j][k]);
                    acc += s * std::ldexp(1.0, hsa[16 * g + i] - 127) * (m < 2 ? std::ldexp(1.0, hsb[16 * g + j] - 127) : 1.0);
                }
                maxd = std::fmax(maxd, std::fabs(acc - hC[i * 16 + j])); maxr = std::fmax(maxr, std::fabs(acc));
            }
        printf("mode %d (%s): max |C - model| = %g (max |C| %g)\n", mode,
               mode == 0 ? "fp6 x fp6, raw" : mode == 1 ? "fp6 x fp4, raw" : mode == 2 ? "fp6 x cvt_pk32_fp6_f16(scale 2)" : mode == 3 ? "fp6 x cvt_pk_fp4_f16" : "fp6 x fp6, scale bytes by OPSEL 2 / 1", maxd, maxr);
        if (maxd != 0) rc = 1;
        if (m == 2 || m == 3) {
            int bad = 0;
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 32; ++i) {
                    const double v = m == 2 ? 2.0 * dec6(field(hraw + l * 6, 6 * i, 6)) : dec4(field(hraw + l * 6, 4 * i, 4));
                    if (v != (m == 2 ? fBh6[l][i] : fBh4[l][i])) ++bad;
                }
            printf("   conversion field order / encoding: %d of 2048 fields differ from the model\n", bad);
            if (bad) rc = 1;
        }
    }
    // rounding / saturation behaviour of the conversions (informational)
    {
        const float probe[8] = {0.06f, 0.19f, 0.9f, 1.06f, 7.6f, 9.0f, 100.0f, -0.3f};
        _Float16 hx[64 * 32] = {};
        for (int i = 0; i < 8; ++i) hx[i] = (_Float16)probe[i];
        hipMemcpy(dBh6, hx, sizeof hx, hipMemcpyHostToDevice);
        for (int m = 2; m < 4; ++m) {
            hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, dA, dB6, dBh6, dsa, dsb, dC, draw, m, 0);
            hipMemcpy(hraw, draw, sizeof hraw, hipMemcpyDeviceToHost);
            printf("%s:", m == 2 ? "cvt_pk32_fp6_f16 scale 2 (value = 2 x field)" : "cvt_pk_fp4_f16 scale 1");
            for (int i = 0; i < 8; ++i) printf("  %g -> %g", probe[i], m == 2 ? 2.0 * dec6(field(hraw, 6 * i, 6)) : dec4(field(hraw, 4 * i, 4)));
            printf("\n");
        }
    }
    unsigned long long* dt; float* dsink; unsigned long long ht[8];
    hipMalloc(&dt, 64); hipMalloc(&dsink, 4096);
    const char* names[9] = {"mfma f16 16x16x32", "mfma fp6 x fp6 16x16x128", "mfma fp6 x fp4 16x16x128", "mfma fp8 x fp8 16x16x128", "4 f16 + 1 fp6 chained (per 5)",
                            "cvt_pk32_fp6_f16", "cvt_pk_fp4_f16", "f16 mfma + cvt_pk32 (per pair)", "f16 mfma + 2 cvt_pk_fp4 (per triple)"};
    const int per_iter[9] = {4, 4, 4, 4, 2, 4, 4, 4, 4};
    for (int waves = 4; waves <= 8; waves += 4)
        for (int kind = 0; kind < 9; ++kind) {
            const int n = 512;
            for (int rep = 0; rep < 2; ++rep) {
                switch (kind) {
                    case 0: hipLaunchKernelGGL(rate<0>, dim3(1), dim3(64 * waves), 0, 0, dt, dsink, n); break;
                    case 1: hipLaunchKernelGGL(rate<1>, dim3(1), dim3(64 * waves), 0, 0, dt, dsink, n); break;
                    case 2: hipLaunchKernelGGL(rate<2>, dim3(1), dim3(64 * waves), 0, 0, dt, dsink, n); break;
                    case 3: hipLaunchKernelGGL(rate<3>, dim3(1), dim3(64 * waves), 0, 0, dt, dsink, n); break;
                    case 4: hipLaunchKernelGGL(rate<4>, dim3(1), dim3(64 * waves), 0, 0, dt, dsink, n); break;
                    case 5: hipLaunchKernelGGL(rate<5>, dim3(1), dim3(64 * waves), 0, 0, dt, dsink, n); break;
                    case 6: hipLaunchKernelGGL(rate<6>, dim3(1), dim3(64 * waves), 0, 0, dt, dsink, n); break;
                    case 7: hipLaunchKernelGGL(rate<7>, dim3(1), dim3(64 * waves), 0, 0, dt, dsink, n); break;
                    default: hipLaunchKernelGGL(rate<8>, dim3(1), dim3(64 * waves), 0, 0, dt, dsink, n); break;
                }
                hipDeviceSynchronize();
            }
            hipMemcpy(ht, dt, 8 * waves, hipMemcpyDeviceToHost);
            unsigned long long mx = 0;
            for (int i = 0; i < waves; ++i) mx = ht[i] > mx ? ht[i] : mx;
            // s_memtime counts shader clocks
            printf("%d waves/CU  %-40s %8.3f cycles per unit (%d units per wave)\n", waves, names[kind], (double)mx / (n * per_iter[kind]), n * per_iter[kind]);
        }
    return rc;
}
