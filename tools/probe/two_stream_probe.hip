// VERDICT r2 item 4, priced before it is built: ONE wave per SIMD on 512 registers carrying TWO independent 2-clip streams, against the
// product's TWO waves per SIMD with one stream each.  The probe runs the encoder's MLP phase as the kernel orders it -- per chunk: a
// workgroup barrier (the ring step), sixteen 1 KiB fragments read from LDS, fc1 (24 MFMAs per stream, chains of 4), ReLU + pack
// (compiler-visible packed max: no inline asm on MFMA results), fc2 (24 MFMAs per stream) -- and, in the "block" modes, the VALU work
// of LayerNorm + attention (~1 150 VALU instructions per block and stream, here dependent v_fma chains over 48 registers) either as
// a separate phase (product order) or interleaved by the compiler into the other stream's MLP chunks (the two-stream proposal).
//   mode 0: 8 waves x 1 stream, MLP only          mode 1: 4 waves x 2 streams, MLP only (a fragment read feeds 6 MFMAs instead of 3)
//   mode 2: 8 waves x 1 stream, MLP + VALU phase  mode 3: 4 waves x 2 streams, stream A's VALU phase inside stream B's MLP and v.v.
//   mode 4: two 4-wave workgroups per CU (own barriers), VALU phase then MLP, in phase      mode 6: the second one half a block off
// Output: cycles per block and stream-pair on one SIMD (lower is better), and cycles per MFMA of the SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o two_stream_probe two_stream_probe.hip && ./two_stream_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
constexpr int CHUNKS = 12, VALU_PER_BLOCK = 1152;

__device__ __forceinline__ h4 pack4(f4 x) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 a = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x[0], x[1])), b = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x[2], x[3]));
    return h4{a[0], a[1], b[0], b[1]};
}
__device__ __forceinline__ h8 relu_pack8(f4 a, f4 b) {
    const h4 z{0, 0, 0, 0};
    const h4 p = __builtin_elementwise_max(pack4(a), z), q = __builtin_elementwise_max(pack4(b), z);
    return __builtin_shufflevector(p, q, 0, 1, 2, 3, 4, 5, 6, 7);
}

// one MLP chunk for NS streams sharing the fragments; VW = VALU filler instructions woven in (per chunk), acting on F (another stream's
// LayerNorm / attention registers)
template <int NT, int S0, int NS, int VW>                   // streams S0 .. S0+NS-1 of the wave's NT
__device__ __forceinline__ void mlp_chunk(const char* slot, int lane, f4 (&R)[NT][8][3], h8 (&X)[NT][4][3], float (&F)[48], float fk) {
    __builtin_amdgcn_s_barrier();
    h8 A[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) A[f] = *(const h8*)(slot + f * 1024 + lane * 16);
    h8 HX[NT][3];
    constexpr int PER = VW / 48;                             // woven VALU instructions behind every MFMA (pinned: sched_barrier)
    int vi = 0;
    auto weave = [&]() {
        if constexpr (PER > 0) {
#pragma unroll
            for (int q = 0; q < PER; ++q, ++vi) F[vi % 48] = __builtin_fmaf(F[vi % 48], fk, F[(vi + 7) % 48]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // fc1 k-step-major: 6 independent chains per stream interleave (as the product kernel writes it)
    f4 H[NT][2][3];
#pragma unroll
    for (int s = S0; s < S0 + NS; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) H[s][t][ct] = f4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = S0; s < S0 + NS; ++s)
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) { H[s][t][ct] = MF(A[4 * t + k], X[s][k][ct], H[s][t][ct]); weave(); }
    h8 A2[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) A2[f] = *(const h8*)(slot + (8 + f) * 1024 + lane * 16);
#pragma unroll
    for (int s = S0; s < S0 + NS; ++s)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) HX[s][ct] = relu_pack8(H[s][0][ct], H[s][1][ct]);
#pragma unroll
    for (int ft = 0; ft < 8; ++ft)
#pragma unroll
        for (int s = S0; s < S0 + NS; ++s)
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) { R[s][ft][ct] = MF(A2[ft], HX[s][ct], R[s][ft][ct]); weave(); }
}

template <int MODE>
__global__ __launch_bounds__((MODE & 1) || MODE >= 4 ? 256 : 512, MODE >= 4 ? 2 : 1) void k(int nblk, unsigned long long* out, float* sink, const h8* src, float fk) {
    extern __shared__ char smem[];                         // 4 slots x 16 KiB of fragments
    constexpr int NS = (MODE & 1) ? 2 : 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((h8*)smem)[i] = src[i & 1023];
    f4 R[NS][8][3];
    h8 X[NS][4][3];
    float F[48];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int i = 0; i < 24; ++i) R[s][i / 3][i % 3] = f4{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 12; ++i) X[s][i / 3][i % 3] = src[(threadIdx.x + 64 * (i + 12 * s)) & 1023];
    }
#pragma unroll
    for (int i = 0; i < 48; ++i) F[i] = (float)src[(lane + i) & 1023][0];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int b = 0; b < nblk; ++b) {
        if constexpr (MODE == 0 || MODE == 1) {
#pragma unroll 1
            for (int ch = 0; ch < CHUNKS; ++ch) mlp_chunk<NS, 0, NS, 0>(smem + (ch & 3) * 16384, lane, R, X, F, fk);
        } else if constexpr (MODE == 2) {                  // product order: the VALU phase, then the MLP
#pragma unroll 1
            for (int o = 0; o < VALU_PER_BLOCK / 48; ++o)
#pragma unroll
                for (int i = 0; i < 48; ++i) F[i] = __builtin_fmaf(F[i], fk, F[(i + 7) % 48]);
#pragma unroll 1
            for (int ch = 0; ch < CHUNKS; ++ch) mlp_chunk<1, 0, 1, 0>(smem + (ch & 3) * 16384, lane, R, X, F, fk);
        } else if constexpr (MODE == 4 || MODE == 6) {     // two 4-wave workgroups per CU (own barriers): VALU phase + MLP, in phase (4) or the
            const bool mlp_first = MODE == 6 && blockIdx.x >= gridDim.x / 2;       // second-dispatched workgroup of a CU half a block off (6)
            if (!mlp_first) {
#pragma unroll 1
                for (int o = 0; o < VALU_PER_BLOCK / 48; ++o)
#pragma unroll
                    for (int i = 0; i < 48; ++i) F[i] = __builtin_fmaf(F[i], fk, F[(i + 7) % 48]);
            }
#pragma unroll 1
            for (int ch = 0; ch < CHUNKS; ++ch) mlp_chunk<1, 0, 1, 0>(smem + (ch & 3) * 16384, lane, R, X, F, fk);
            if (mlp_first) {
#pragma unroll 1
                for (int o = 0; o < VALU_PER_BLOCK / 48; ++o)
#pragma unroll
                    for (int i = 0; i < 48; ++i) F[i] = __builtin_fmaf(F[i], fk, F[(i + 7) % 48]);
            }
        } else {                                           // two streams half a block apart: each stream's MLP chunk carries 1/12 of the other's VALU phase
#pragma unroll 1
            for (int ch = 0; ch < CHUNKS; ++ch) mlp_chunk<2, 0, 1, VALU_PER_BLOCK / CHUNKS>(smem + (ch & 3) * 16384, lane, R, X, F, fk);
#pragma unroll 1
            for (int ch = 0; ch < CHUNKS; ++ch) mlp_chunk<2, 1, 1, VALU_PER_BLOCK / CHUNKS>(smem + (ch & 3) * 16384, lane, R, X, F, fk);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int i = 0; i < 24; ++i) r += R[s][i / 3][i % 3][0];
#pragma unroll
    for (int i = 0; i < 48; ++i) r += F[i];
    if (r == 12345.678f) sink[threadIdx.x] = r;
    if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
}

static unsigned long long* d_out; static float* d_sink; static h8* d_src; static std::vector<unsigned long long> h(512 * 8);
template <int MODE> double run(const char* name) {
    const int nblk = 48, waves = (MODE & 1) || MODE >= 4 ? 4 : 8, grid = MODE >= 4 ? 512 : 256;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 3; ++rep) { k<MODE><<<grid, waves * 64, 65536>>>(nblk, d_out, d_sink, d_src, 0.999f); hipDeviceSynchronize(); }
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> v;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < waves; ++w) v.push_back((double)h[b * 8 + w]);
    std::sort(v.begin(), v.end());
    // s_memtime ticks at 100 MHz on gfx950; report in shader cycles via a measured ratio is not needed for an A/B: ticks per block + pair
    // streams per SIMD: both layouts carry 2 streams per SIMD; MFMAs per SIMD per block = 2 x 12 x 48
    const double ticks = v[v.size() / 2];
    // mode 3 runs TWO stream-blocks per loop iteration and wave (one per stream), like the others (2 waves x 1 or 1 wave x 2)
    const double per_block_pair = ticks / nblk;
    printf("%-64s %9.1f ticks per block and SIMD stream-pair   (%.3f ticks per MFMA)\n", name, per_block_pair, per_block_pair / (2.0 * CHUNKS * 48));
    return per_block_pair;
}
int main() {
    hipMalloc(&d_out, 512 * 8 * 8); hipMalloc(&d_sink, 4096); hipMalloc(&d_src, 1024 * 16);
    std::vector<_Float16> init(1024 * 8);
    for (size_t i = 0; i < init.size(); ++i) init[i] = (_Float16)(((int)(i * 37 % 17) - 8) * 0.01f);
    hipMemcpy(d_src, init.data(), init.size() * 2, hipMemcpyHostToDevice);
    const double a = run<0>("0: 2 waves/SIMD x 1 stream, MLP phase only");
    const double b = run<1>("1: 1 wave/SIMD x 2 streams (512 regs), MLP phase only");
    const double c = run<2>("2: 2 waves/SIMD x 1 stream, VALU phase then MLP (product order)");
    const double d = run<3>("3: 1 wave/SIMD x 2 streams, each stream's VALU inside the other's MLP");
    printf("two-stream / product: MLP only %.3f, whole block %.3f\n", b / a, d / c);
    const double e = run<4>("4: two 4-wave workgroups per CU, VALU phase then MLP, in phase");
    const double f = run<6>("6: the same, the second workgroup of a CU half a block off (anti-phase)");
    printf("anti-phase / in-phase (two workgroups per CU): %.3f;  in-phase two workgroups / one 8-wave workgroup: %.3f\n", f / e, e / c);
    return 0;
}
