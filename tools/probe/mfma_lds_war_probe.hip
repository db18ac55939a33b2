// GPU probe (round 6): can the data of a ds_read land in a VGPR that an MFMA issued just BEFORE it still has to read (A operand:
// WAR) or still has to WRITE (its destination: WAW)?  hipcc re-uses the registers of a finished-in-program-order MFMA for the next
// fragment load right behind it (the chained lab build of the f16ax BNeckAttention kernel: `v_mfma v[132:135], .., v[136:139]` ;
// `ds_read_b128 v[136:139]` two instructions later, profiles/round6/f16ax_race.log) and pads nothing there: an LDS round trip is
// assumed to outlast any MFMA in flight.  This probe pins both patterns with inline asm and counts violations, with and without a
// sibling wave streaming MFMAs on the same SIMD.
//
//   hipcc --offload-arch=gfx950 -O2 -o mfma_lds_war_probe mfma_lds_war_probe.hip && ./mfma_lds_war_probe
//
// mode 0 (WAR on A): acc += A.B in a dependent chain of K MFMAs, then `ds_read_b128 A <- poison`.  A = B = 1.0: every MFMA adds 32;
//   an MFMA that read the poisoned A (1024.0) adds 32768.  Expected acc = 32 K per round, exactly.
// (every asm statement opens with s_nop 7: the compiler pads ONE state between its own v_mov of `a` and an asm MFMA that reads it -- the
//  first version of this probe measured exactly that hazard: the first MFMA of a round read the previous round's poison)
// mode 1 (WAW on D): P: D1 = A.B + D0 ; C: D2 = A.B + D1 (dependent on P) ; `ds_read_b128 D1 <- poison`.  After lgkmcnt(0) and a drain
//   D1 must hold the poison (the load is the last writer in program order); if P's write-back came later it holds 32 (K+1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define PAD "s_nop 7\n\t"
#define CHAIN1 "v_mfma_f32_16x16x32_f16 %[acc], %[A], %[B], %[acc]\n\t"
#define CHAIN4 CHAIN1 CHAIN1 CHAIN1 CHAIN1

template <int MODE, int K>
__global__ __launch_bounds__(512) void probe(int rounds, int sibling_streams, unsigned* bad, float* firstbad) {
    __shared__ __attribute__((aligned(16))) _Float16 poison[8 * 64 * 8];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8 * 64 * 8; i += 512) poison[i] = (_Float16)1024.f;
    __syncthreads();
    h8 one;
    for (int i = 0; i < 8; ++i) one[i] = (_Float16)1.f;
    h8 A = one, B = one;
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(poison + (w * 64 + lane) * 8);
    unsigned nbad = 0;
    float fb = 0.f;
    if (sibling_streams && w >= 4) {                                        // waves 4..7 (the second wave of every SIMD): MFMA traffic only
        f4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
        for (int r = 0; r < rounds * 8; ++r)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %5, %1\n\t"
                         "v_mfma_f32_16x16x32_f16 %2, %4, %5, %2\n\tv_mfma_f32_16x16x32_f16 %3, %4, %5, %3"
                         : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(A), "v"(B));
        if (s0[0] + s1[0] + s2[0] + s3[0] == 12345.f) firstbad[threadIdx.x] = s0[0];
        return;
    }
    for (int r = 0; r < rounds; ++r) {
        if constexpr (MODE == 0) {
            f4 acc = {0, 0, 0, 0};
            h8 a = one;
            // K dependent MFMAs, then the load that overwrites their A operand; the wait + drain; result checked in C++
            if constexpr (K == 1) asm volatile(PAD CHAIN1 "ds_read_b128 %[A], %[ad]\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" : [acc] "+v"(acc), [A] "+v"(a) : [B] "v"(B), [ad] "v"(addr) : "memory");
            if constexpr (K == 4) asm volatile(PAD CHAIN4 "ds_read_b128 %[A], %[ad]\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" : [acc] "+v"(acc), [A] "+v"(a) : [B] "v"(B), [ad] "v"(addr) : "memory");
            if constexpr (K == 12) asm volatile(PAD CHAIN4 CHAIN4 CHAIN4 "ds_read_b128 %[A], %[ad]\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" : [acc] "+v"(acc), [A] "+v"(a) : [B] "v"(B), [ad] "v"(addr) : "memory");
            const bool ok = acc[0] == 32.f * K && acc[1] == 32.f * K && acc[2] == 32.f * K && acc[3] == 32.f * K && (float)a[0] == 1024.f;
            if (!ok) { if (!nbad) fb = acc[0]; ++nbad; }
        } else {
            f4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0}, d2 = {0, 0, 0, 0};
            // a chain of K on d0 (keeps the pipe busy and P late), P: d1 = A.B + d0, C: d2 = A.B + d1, then the load INTO d1
#define TAIL "v_mfma_f32_16x16x32_f16 %[d1], %[A], %[B], %[d0]\n\tv_mfma_f32_16x16x32_f16 %[d2], %[A], %[B], %[d1]\n\tds_read_b128 %[d1], %[ad]\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15"
#define CH1 "v_mfma_f32_16x16x32_f16 %[d0], %[A], %[B], %[d0]\n\t"
            if constexpr (K == 1) asm volatile(PAD CH1 TAIL : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2) : [A] "v"(A), [B] "v"(B), [ad] "v"(addr) : "memory");
            if constexpr (K == 4) asm volatile(PAD CH1 CH1 CH1 CH1 TAIL : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2) : [A] "v"(A), [B] "v"(B), [ad] "v"(addr) : "memory");
            if constexpr (K == 12) asm volatile(PAD CH1 CH1 CH1 CH1 CH1 CH1 CH1 CH1 CH1 CH1 CH1 CH1 TAIL : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2) : [A] "v"(A), [B] "v"(B), [ad] "v"(addr) : "memory");
            const h8 got = __builtin_bit_cast(h8, d1);
            const bool ok = (float)got[0] == 1024.f && (float)got[7] == 1024.f && d2[0] == 32.f * (K + 2);
            if (!ok) { if (!nbad) fb = d1[0]; ++nbad; }
        }
    }
    if (nbad) { atomicAdd(bad, nbad); firstbad[blockIdx.x * 512 + threadIdx.x] = fb; }
}

template <int MODE, int K>
static void run(const char* what, int sibling) {
    unsigned* bad; float* fb;
    const int grid = 512, rounds = 20000;
    hipMalloc(&bad, 4); hipMalloc(&fb, grid * 512 * 4);
    hipMemset(bad, 0, 4); hipMemset(fb, 0, grid * 512 * 4);
    hipLaunchKernelGGL((probe<MODE, K>), dim3(grid), dim3(512), 0, 0, rounds, sibling, bad, fb);
    hipDeviceSynchronize();
    unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    std::vector<float> f(grid * 512); hipMemcpy(f.data(), fb, f.size() * 4, hipMemcpyDeviceToHost);
    float ex = 0.f; for (float v : f) if (v != 0.f) { ex = v; break; }
    const long total = (long)grid * (sibling ? 256 : 512) * rounds;
    printf("%-52s K=%2d sibling MFMA stream %d: %u of %ld lane-rounds wrong%s", what, K, sibling, h, total, h ? "" : "\n");
    if (h) printf(" (e.g. %g)\n", ex);
    hipFree(bad); hipFree(fb);
}

int main() {
    for (int sib = 0; sib < 2; ++sib) {
        run<0, 1>("WAR: ds_read overwrites the A operand of an MFMA", sib);
        run<0, 4>("WAR: ds_read overwrites the A operand of an MFMA", sib);
        run<0, 12>("WAR: ds_read overwrites the A operand of an MFMA", sib);
        run<1, 1>("WAW: ds_read into the destination of an MFMA", sib);
        run<1, 4>("WAW: ds_read into the destination of an MFMA", sib);
        run<1, 12>("WAW: ds_read into the destination of an MFMA", sib);
    }
    return 0;
}
