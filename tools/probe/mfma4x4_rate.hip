// v_mfma_f32_4x4x1_16b_f32 issue rate vs number of independent accumulation chains (gfx950), one / two waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(512) void k(int n, unsigned long long* out, float* sink) {
    f4 acc[8]; for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 0.01f, b = 1.0f + threadIdx.x * 0.001f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 48; ++u) { acc[u % CH] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u % CH], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0; for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
    if (r == 12345.678f) sink[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
static unsigned long long* d_out; static float* d_sink; static std::vector<unsigned long long> h(256 * 8);
template <int CH> void run() {
    double r[2];
    for (int two = 0; two < 2; ++two) {
        int thr = two ? 512 : 256;
        for (int rep = 0; rep < 2; ++rep) { k<CH><<<256, thr>>>(256, d_out, d_sink); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> v; for (int b = 0; b < 256; ++b) for (int w = 0; w < thr / 64; ++w) v.push_back((double)h[b * 8 + w]);
        std::sort(v.begin(), v.end()); r[two] = v[v.size() / 2] / (256.0 * 48);
    }
    printf("4x4x1 f32, %d chain(s): 1w/SIMD %5.1f cyc/MFMA   2w/SIMD %5.1f per SIMD-MFMA\n", CH, r[0], r[1] / 2);
}
int main() { hipMalloc(&d_out, 256 * 64); hipMalloc(&d_sink, 4096); run<1>(); run<2>(); run<4>(); run<8>(); return 0; }
