// valu_dep_probe.hip -- dependent-issue cost of the packed-f32 VALU instructions the front-end is built from (gfx950).
// One wave per SIMD (workgroup = 4 waves, one workgroup per CU, pinned inline-asm streams): NCH independent chains of the
// same instruction, round-robin, 2048 instructions; prints cycles per instruction.  NCH = 1 is a pure dependent chain.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/valu_dep_probe tools/probe/valu_dep_probe.hip && tools/probe/valu_dep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));

#define CHAIN_KERNEL(NAME, INSTR)                                                                              \
template <int NCH> __global__ __launch_bounds__(256) void NAME(unsigned long long* out, float* sink, int n) { \
    f2 r[8]; f2 c = {1.0001f, 0.9999f}; f2 d = {1e-3f, -1e-3f};                                                \
    for (int i = 0; i < 8; ++i) r[i] = f2{threadIdx.x * 1e-3f + i, 1.f + i};                                   \
    __syncthreads();                                                                                           \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                     \
    for (int it = 0; it < n; ++it) {                                                                           \
        _Pragma("unroll") for (int u = 0; u < 64; ++u) {                                                       \
            asm volatile(INSTR : "+v"(r[u % NCH]) : "v"(c), "v"(d));                                           \
        }                                                                                                      \
    }                                                                                                          \
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                                                        \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                     \
    float s = 0; for (int i = 0; i < 8; ++i) s += r[i][0] + r[i][1];                                           \
    if (s == 12345.678f) sink[threadIdx.x] = s;                                                                \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                          \
}
CHAIN_KERNEL(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
CHAIN_KERNEL(k_pk_add, "v_pk_add_f32 %0, %0, %2")
CHAIN_KERNEL(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
CHAIN_KERNEL(k_pk_add_sw, "v_pk_add_f32 %0, %0, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")

// scalar f32 on the low half of the pair
#define CHAIN_KERNEL1(NAME, INSTR)                                                                             \
template <int NCH> __global__ __launch_bounds__(256) void NAME(unsigned long long* out, float* sink, int n) { \
    float r[8]; float c = 1.0001f, d = 1e-3f;                                                                  \
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 1e-3f + i;                                                \
    __syncthreads();                                                                                           \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                     \
    for (int it = 0; it < n; ++it) {                                                                           \
        _Pragma("unroll") for (int u = 0; u < 64; ++u) {                                                       \
            asm volatile(INSTR : "+v"(r[u % NCH]) : "v"(c), "v"(d));                                           \
        }                                                                                                      \
    }                                                                                                          \
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                                                        \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                     \
    float s = 0; for (int i = 0; i < 8; ++i) s += r[i];                                                        \
    if (s == 12345.678f) sink[threadIdx.x] = s;                                                                \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                          \
}
CHAIN_KERNEL1(k_fma, "v_fma_f32 %0, %0, %1, %2")
CHAIN_KERNEL1(k_add, "v_add_f32 %0, %0, %2")

template <typename K> static double run(K k, int nwg) {
    unsigned long long* out; float* sink;
    hipMalloc(&out, nwg * 4 * sizeof(unsigned long long)); hipMalloc(&sink, 1024);
    const int n = 32;
    hipLaunchKernelGGL(k, dim3(nwg), dim3(256), 0, 0, out, sink, n);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(256), 0, 0, out, sink, n);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nwg * 4);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    hipFree(out); hipFree(sink);
    // s_memtime ticks at 100 MHz on gfx9-family parts: convert with the shader clock measured by the caller; report raw ticks per instr too
    return (double)h[h.size() / 2] / (n * 64.0);
}
#define ROW(NAME) printf("%-14s  chains 1: %7.4f   2: %7.4f   4: %7.4f   8: %7.4f   ticks/instr\n", #NAME, \
    run(NAME<1>, nwg), run(NAME<2>, nwg), run(NAME<4>, nwg), run(NAME<8>, nwg));
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int nwg = p.multiProcessorCount;      // one workgroup per CU
    printf("device %s, %d CUs, clock %d kHz; s_memtime ticks (convert: cycles = ticks * f_shader / f_memtime)\n", p.name, nwg, p.clockRate);
    ROW(k_fma) ROW(k_add) ROW(k_pk_fma) ROW(k_pk_add) ROW(k_pk_mul) ROW(k_pk_add_sw)
    return 0;
}
