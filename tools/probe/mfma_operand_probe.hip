// Does the back-to-back rate of v_mfma_f32_16x16x32_f16 depend on which registers feed it?  (gfx950)
// Compiler-allocated registers, MFMA order pinned by a sched_barrier after each one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define SB __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ __launch_bounds__(512) void k(int n, unsigned long long* out, float* sink, const h8* src) {
    const int w = threadIdx.x >> 6;
    h8 A[8], X[4][3], HX[3];
    for (int i = 0; i < 8; ++i) A[i] = src[threadIdx.x + 64 * i];
    for (int i = 0; i < 12; ++i) X[i / 3][i % 3] = src[threadIdx.x + 64 * (8 + i)];
    for (int i = 0; i < 3; ++i) HX[i] = src[threadIdx.x + 64 * (20 + i)];
    f4 R[8][3], H[2][3];
    for (int i = 0; i < 24; ++i) R[i / 3][i % 3] = f4{0, 0, 0, 0};
    for (int i = 0; i < 6; ++i) H[i / 3][i % 3] = f4{0, 0, 0, 0};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
        if (MODE == 0) {            // 48 MFMAs, all reading the same A/B registers, 24 accumulators
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int ft = 0; ft < 8; ++ft)
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) { R[ft][ct] = MF(A[0], HX[0], R[ft][ct]); SB; }
        } else if (MODE == 1) {     // fc2 shape: R[ft][ct] += A[ft] * HX[ct]   (24 MFMAs) twice
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int ft = 0; ft < 8; ++ft)
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) { R[ft][ct] = MF(A[ft], HX[ct], R[ft][ct]); SB; }
        } else if (MODE == 2) {     // fc1 shape: 6 accumulators, chains of 4 over s  (24 MFMAs) twice
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                        for (int s = 0; s < 4; ++s) { H[j][ct] = MF(A[4 * j + s], X[s][ct], H[j][ct]); SB; }
        } else {                    // the MLP chunk as the kernel orders it: fc2 (24) then fc1 (24)
#pragma unroll
            for (int ft = 0; ft < 8; ++ft)
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) { R[ft][ct] = MF(A[ft], HX[ct], R[ft][ct]); SB; }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                    for (int s = 0; s < 4; ++s) { H[j][ct] = MF(A[4 * j + s], X[s][ct], H[j][ct]); SB; }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < 24; ++i) r += R[i / 3][i % 3][0];
    for (int i = 0; i < 6; ++i) r += H[i / 3][i % 3][1];
    if (r == 12345.678f) sink[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + w] = t1 - t0;
}

static unsigned long long* d_out; static float* d_sink; static h8* d_src; static std::vector<unsigned long long> h(256 * 8);
static double med(int nw) { std::vector<double> v; for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) v.push_back((double)h[b * 8 + w]); std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
template <int MODE> void run(const char* name) {
    const int n = 512;
    double r[2];
    for (int two = 0; two < 2; ++two) {
        const int thr = two ? 512 : 256;
        for (int rep = 0; rep < 2; ++rep) { k<MODE><<<256, thr>>>(n, d_out, d_sink, d_src); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        r[two] = med(thr / 64) / (n * 48.0);
    }
    printf("%-34s 1w/SIMD %6.2f cyc/MFMA   2w/SIMD %6.2f per SIMD-MFMA\n", name, r[0], r[1] / 2);
}
int main() {
    hipMalloc(&d_out, 256 * 8 * 8); hipMalloc(&d_sink, 4096); hipMalloc(&d_src, 512 * 32 * 16); hipMemset(d_src, 0x3c, 512 * 32 * 16);
    run<0>("same A/B registers"); run<1>("fc2 shape (8 A x 3 B, 24 acc)"); run<2>("fc1 shape (chains of 4)"); run<3>("MLP chunk order (fc2 then fc1)");
    return 0;
}
