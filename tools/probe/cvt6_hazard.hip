// Probe (GPU box): is the result of v_cvt_scalef32_pk32_fp6_f16 safe to consume by the next instructions?  Eight waves per CU (two per
// SIMD, MFMA-heavy neighbours), every wave repeats: [a few f16 MFMAs] -> three back-to-back conversions -> NOPS wait states -> an MX
// MFMA per conversion that consumes the result, and compares the accumulators with those of a reference pass that puts 64 wait states
// behind the conversions.  Prints the number of mismatching iterations per NOPS setting.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe/cvt6_hazard tools/probe/cvt6_hazard.hip && tools/probe/cvt6_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned u6 __attribute__((ext_vector_type(6)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ v8i pad8(u6 x) { return v8i{(int)x[0], (int)x[1], (int)x[2], (int)x[3], (int)x[4], (int)x[5], 0, 0}; }

template <int NOPS>
__global__ void k(unsigned* mism, int iters) {
    const int l = threadIdx.x & 63;
    h32 x0, x1, x2;
    for (int i = 0; i < 32; ++i) { x0[i] = (_Float16)(float)((l + i) % 13 - 6); x1[i] = (_Float16)(float)((l * 3 + i) % 11 - 5); x2[i] = (_Float16)(float)((l + 2 * i) % 9 - 4); }
    v8i w = {0x11111111, 0x22222222, 0x09090909, 0x11111111, 0x12121212, 0x21212121, 0, 0};
    h8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1};
    unsigned bad = 0;
    f4 ref0, ref1, ref2;
    for (int it = -1; it < iters; ++it) {
        f4 c = {0, 0, 0, 0};
        // some MFMA traffic in front (as in the encoder: the conversions follow a LayerNorm that follows MFMAs)
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c, 0, 0, 0);
        // perturb the inputs a little per iteration so that a stale destination from the previous iteration is visible
        x0[0] = (_Float16)(float)((it & 7) - 3); x1[5] = (_Float16)(float)((it & 3) - 1); x2[9] = (_Float16)(float)((it & 5) - 2);
        u6 y0, y1, y2;
        asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %3, 2.0\n\tv_cvt_scalef32_pk32_fp6_f16 %1, %4, 2.0\n\tv_cvt_scalef32_pk32_fp6_f16 %2, %5, 2.0"
                     : "=&v"(y0), "=&v"(y1), "=&v"(y2) : "v"(x0), "v"(x1), "v"(x2));
        if (it < 0) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(y0), "+v"(y1), "+v"(y2));
        else if (NOPS >= 16) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(y0), "+v"(y1), "+v"(y2));
        else if (NOPS >= 8) asm volatile("s_nop 7" : "+v"(y0), "+v"(y1), "+v"(y2));
        else if (NOPS >= 4) asm volatile("s_nop 3" : "+v"(y0), "+v"(y1), "+v"(y2));
        else if (NOPS >= 1) asm volatile("s_nop 0" : "+v"(y0), "+v"(y1), "+v"(y2));
        f4 z = {0, 0, 0, 0};
        f4 a0, a1, a2;
        // consumers written as asm so that the compiler inserts nothing of its own between producer and consumer
        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %3, %4, %7, %8, %8 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
                     "v_mfma_scale_f32_16x16x128_f8f6f4 %1, %3, %5, %7, %8, %8 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
                     "v_mfma_scale_f32_16x16x128_f8f6f4 %2, %3, %6, %7, %8, %8 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\ts_nop 7\n\ts_nop 7"
                     : "=&v"(a0), "=&v"(a1), "=&v"(a2) : "v"(__builtin_shufflevector(w, w, 0, 1, 2, 3, 4, 5)), "v"(y0), "v"(y1), "v"(y2), "v"(z), "v"(127));
        // per-iteration reference: the same conversions consumed after 64 wait states
        u6 r0, r1, r2;
        asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %3, 2.0\n\tv_cvt_scalef32_pk32_fp6_f16 %1, %4, 2.0\n\tv_cvt_scalef32_pk32_fp6_f16 %2, %5, 2.0\n\t"
                     "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(x0), "v"(x1), "v"(x2));
        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %3, %4, %7, %8, %8 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
                     "v_mfma_scale_f32_16x16x128_f8f6f4 %1, %3, %5, %7, %8, %8 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
                     "v_mfma_scale_f32_16x16x128_f8f6f4 %2, %3, %6, %7, %8, %8 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\ts_nop 7\n\ts_nop 7"
                     : "=&v"(ref0), "=&v"(ref1), "=&v"(ref2) : "v"(__builtin_shufflevector(w, w, 0, 1, 2, 3, 4, 5)), "v"(r0), "v"(r1), "v"(r2), "v"(z), "v"(127));
        if (it >= 0) {
            bool ne = false;
            for (int i = 0; i < 4; ++i) ne |= a0[i] != ref0[i] || a1[i] != ref1[i] || a2[i] != ref2[i];
            bool e0 = false; for (int i = 0; i < 4; ++i) e0 |= a0[i] != ref0[i];
            if (ne) bad += 1 + (e0 ? 0x10000 : 0);
        }
        asm volatile("" :: "v"(c));
    }
    atomicAdd(mism + (NOPS >= 16 ? 4 : NOPS >= 8 ? 3 : NOPS >= 4 ? 2 : NOPS >= 1 ? 1 : 0), bad & 0xffff);
    atomicAdd(mism + 8 + (NOPS >= 16 ? 4 : NOPS >= 8 ? 3 : NOPS >= 4 ? 2 : NOPS >= 1 ? 1 : 0), bad >> 16);
}

int main() {
    unsigned* d; unsigned h[16] = {};
    hipMalloc(&d, 64); hipMemset(d, 0, 64);
    const int iters = 2000, grid = 256;
    hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, d, iters);
    hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, d, iters);
    hipLaunchKernelGGL(k<4>, dim3(grid), dim3(512), 0, 0, d, iters);
    hipLaunchKernelGGL(k<8>, dim3(grid), dim3(512), 0, 0, d, iters);
    hipLaunchKernelGGL(k<16>, dim3(grid), dim3(512), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    const char* names[5] = {"0 wait states", "s_nop 0 (1)", "s_nop 3 (4)", "s_nop 7 (8)", "2 x s_nop 7 (16)"};
    for (int i = 0; i < 5; ++i)
        printf("consumer after %-18s: %u lane-iterations of %d differ from the 64-wait-state reference (%u of them in the FIRST conversion's consumer)\n",
               names[i], h[i], grid * 512 * iters, h[8 + i]);
    return 0;
}
