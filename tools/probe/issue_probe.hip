// gfx950 issue model probe: cycles per MFMA gap when K vector ops of one kind sit beside each 16x16x32 f16 MFMA,
// in one wave (1 or 2 waves per SIMD), and rates of an MFMA-only wave and a VALU-only wave sharing a SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int VT> __device__ __forceinline__ void vop(float& x, float& y) {
    if (VT == 0) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    else if (VT == 1) x = __builtin_amdgcn_exp2f(x);
    else if (VT == 2) { h2 p = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x, y)); x = __builtin_bit_cast(float, p); }
    else if (VT == 3) { f2 v = {x, y}; v = v * f2{1.0001f, 0.9999f}; x = v[0]; y = v[1]; }
    else if (VT == 4) x = __builtin_fmaxf(x, y);
    else if (VT == 5) { f2 v = {x, y}; f2 c = {0.5f, 0.25f}; v = __builtin_elementwise_fma(v, f2{1.0001f, 0.9999f}, c); x = v[0]; y = v[1]; }
}

template <int K, int VT>
__global__ __launch_bounds__(512) void inwave(int n, unsigned long long* out, float* sink) {
    f4 acc[8]; for (int u = 0; u < 8; ++u) acc[u] = f4{0, 0, 0, 0};
    h8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
    float s[8], t[8]; for (int i = 0; i < 8; ++i) { s[i] = threadIdx.x * 1e-3f + i; t[i] = s[i] * 0.5f; }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < K; ++q) vop<VT>(s[(u * K + q) & 7], t[(u * K + q) & 7]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0; for (int i = 0; i < 8; ++i) r += s[i] + t[i];
    for (int u = 0; u < 8; ++u) r += acc[u][0] + acc[u][3];
    if (r == 12345.678f) sink[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

// lo waves: MFMA only, hi waves: VALU only.  which=0: measure MFMA wave (partner runs 4x longer); 1: measure VALU wave
template <int VT, int PRIO, int SWAP = 0, int NOP = -1>
__global__ __launch_bounds__(512) void split(int nm, int nv, unsigned long long* out, float* sink) {
    int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); bool hi = SWAP ? (w < 4) : (w >= 4);
    f4 acc[8]; for (int u = 0; u < 8; ++u) acc[u] = f4{0, 0, 0, 0};
    h8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
    float s[8], t[8]; for (int i = 0; i < 8; ++i) { s[i] = threadIdx.x * 1e-3f + i; t[i] = s[i] * 0.5f; }
    __syncthreads();
    if (PRIO == 1 && hi) __builtin_amdgcn_s_setprio(3);
    if (PRIO == 2 && !hi) __builtin_amdgcn_s_setprio(3);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (!hi) {
        for (int i = 0; i < nm; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
                if (NOP >= 0) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop %0" ::"n"(NOP)); __builtin_amdgcn_sched_barrier(0); }
            }
        }
    } else {
        for (int i = 0; i < nv; ++i) {
#pragma unroll
            for (int q = 0; q < 16; ++q) vop<VT>(s[q & 7], t[q & 7]);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0; for (int i = 0; i < 8; ++i) r += s[i] + t[i];
    for (int u = 0; u < 8; ++u) r += acc[u][0] + acc[u][3];
    if (r == 12345.678f) sink[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + w] = t1 - t0;
}

static unsigned long long* d_out; static float* d_sink; static std::vector<unsigned long long> h(256 * 8);
static double med(int wave_lo, int wave_hi, int nw) {
    std::vector<double> v;
    for (int b = 0; b < 256; ++b) for (int w = wave_lo; w < wave_hi && w < nw; ++w) v.push_back((double)h[b * 8 + w]);
    std::sort(v.begin(), v.end()); return v[v.size() / 2];
}
template <int K, int VT> void run_inwave(const char* name) {
    const int n = 512;
    double r[2];
    for (int two = 0; two < 2; ++two) {
        int thr = two ? 512 : 256;
        for (int rep = 0; rep < 3; ++rep) { inwave<K, VT><<<256, thr>>>(n, d_out, d_sink); }
        hipDeviceSynchronize(); hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        r[two] = med(0, 8, thr / 64) / (n * 8.0);
    }
    printf("inwave %-6s K=%d  1w/SIMD %6.1f cyc/gap   2w/SIMD %6.1f cyc/gap (= %5.1f per SIMD-gap)\n", name, K, r[0], r[1], r[1] / 2);
}
template <int VT, int PRIO, int SWAP = 0, int NOP = -1> void run_split(const char* name, double mf_alone, double v_alone) {
    const int ml = SWAP ? 4 : 0, vl = SWAP ? 0 : 4;
    // measure MFMA waves with a long VALU partner, then VALU waves with a long MFMA partner
    int nm = 2048, nv = 16384;          // long partner
    split<VT, PRIO, SWAP, NOP><<<256, 512>>>(512, nv, d_out, d_sink); hipDeviceSynchronize();
    split<VT, PRIO, SWAP, NOP><<<256, 512>>>(512, nv, d_out, d_sink); hipDeviceSynchronize();
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    double m = med(ml, ml + 4, 8) / (512 * 8.0);
    double vlong = med(vl, vl + 4, 8) / (nv * 16.0);
    split<VT, PRIO, SWAP, NOP><<<256, 512>>>(nm * 8, 512, d_out, d_sink); hipDeviceSynchronize();
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    double v = med(vl, vl + 4, 8) / (512 * 16.0);
    printf("split  %-6s nop=%2d swap=%d prio=%d  MFMA gap beside VALU wave %6.1f cyc (alone %5.1f)   VALU op beside MFMA wave %6.2f cyc (alone %5.2f, mostly-alone %5.2f)\n",
           name, NOP, SWAP, PRIO, m, mf_alone, v, v_alone, vlong);
}
template <int VT> double valu_alone() {
    split<VT, 0><<<256, 512>>>(0, 2048, d_out, d_sink); hipDeviceSynchronize();
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    return med(4, 8, 8) / (2048 * 16.0);
}
int main() {
    hipMalloc(&d_out, 256 * 8 * 8); hipMalloc(&d_sink, 4096);
    run_inwave<0, 0>("none");
#define ROW(VT, NAME) run_inwave<1, VT>(NAME); run_inwave<2, VT>(NAME); run_inwave<3, VT>(NAME); run_inwave<4, VT>(NAME); run_inwave<6, VT>(NAME);
    
    split<0, 0><<<256, 512>>>(2048, 0, d_out, d_sink); hipDeviceSynchronize();
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    double mf = med(0, 4, 8) / (2048 * 8.0);
    double va0 = valu_alone<0>(), va1 = valu_alone<1>(), va2 = valu_alone<2>();
    run_split<0, 0, 0, -1>("fma", mf, va0);
    run_split<0, 0, 0, 0>("fma", mf, va0); run_split<0, 0, 0, 1>("fma", mf, va0); run_split<0, 0, 0, 3>("fma", mf, va0);
    run_split<0, 0, 0, 5>("fma", mf, va0); run_split<0, 0, 0, 7>("fma", mf, va0); run_split<0, 0, 0, 9>("fma", mf, va0);
    run_split<0, 0, 1, 7>("fma", mf, va0); run_split<0, 1, 0, 7>("fma", mf, va0); run_split<0, 1, 1, 7>("fma", mf, va0);
    run_split<1, 0, 0, 7>("exp2", mf, va1); run_split<1, 0, 0, 5>("exp2", mf, va1);
    run_split<2, 0, 0, 7>("cvtpk", mf, va2); run_split<2, 0, 0, 5>("cvtpk", mf, va2);
    return 0;
}
