// Does VALU work of one wave hide under the MFMAs of the other wave on the same SIMD (gfx950)?
// 8-wave workgroups, one per CU; waves w and w+4 share a SIMD.  Roles per mode, see main().
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int VK>
__device__ __forceinline__ void valu_phase(int n, float& s0, float& s1, float& s2, float& s3) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (VK == 0) {            // plain fma chain x4
                s0 = __builtin_fmaf(s0, 1.0001f, 0.5f); s1 = __builtin_fmaf(s1, 1.0001f, 0.5f);
                s2 = __builtin_fmaf(s2, 1.0001f, 0.5f); s3 = __builtin_fmaf(s3, 1.0001f, 0.5f);
            } else if (VK == 1) {     // exp2
                s0 = __builtin_amdgcn_exp2f(s0); s1 = __builtin_amdgcn_exp2f(s1);
                s2 = __builtin_amdgcn_exp2f(s2); s3 = __builtin_amdgcn_exp2f(s3);
            } else {                  // cvt pack + unpack-ish
                auto p = __builtin_amdgcn_cvt_pkrtz(s0, s1); auto q = __builtin_amdgcn_cvt_pkrtz(s2, s3);
                s0 += (float)p[0]; s1 += (float)q[1]; s2 += (float)p[1]; s3 += (float)q[0];
            }
        }
    }
}

__device__ __forceinline__ void mfma_phase(int n, f4 (&acc)[8], h8 a, h8 b) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
    }
}

// mode: 0 = waves0-3 MFMA only (4-7 exit); 1 = waves 4-7 VALU only; 2 = 0-3 MFMA + 4-7 VALU concurrently;
// 3 = all waves: R rounds of (MFMA n, VALU m) in phase; 4 = same, waves 4-7 start with VALU (anti-phase)
template <int VK>
__global__ __launch_bounds__(512) void probe(int mode, int nm, int nv, int rounds, float* out) {
    int w = threadIdx.x >> 6;
    bool hi = w >= 4;
    f4 acc[8]; for (int u = 0; u < 8; ++u) acc[u] = f4{0, 0, 0, 0};
    h8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
    float s0 = threadIdx.x * 1e-3f, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    if (mode == 0) { if (hi) return; mfma_phase(nm * rounds, acc, a, b); }
    else if (mode == 1) { if (!hi) return; valu_phase<VK>(nv * rounds, s0, s1, s2, s3); }
    else if (mode == 2) { if (hi) valu_phase<VK>(nv * rounds, s0, s1, s2, s3); else mfma_phase(nm * rounds, acc, a, b); }
    else if (mode == 5) { if (hi) { __builtin_amdgcn_s_setprio(3); valu_phase<VK>(nv * rounds, s0, s1, s2, s3); } else mfma_phase(nm * rounds, acc, a, b); }
    else if (mode == 6) { if (!hi) valu_phase<VK>(nv * rounds, s0, s1, s2, s3); else mfma_phase(nm * rounds, acc, a, b); }
    else if (mode == 7) {       // anti-phase + priority follows the phase (VALU phase high)
        if (hi) { __builtin_amdgcn_s_setprio(3); valu_phase<VK>(nv, s0, s1, s2, s3); }
        for (int r = 0; r < rounds; ++r) {
            __builtin_amdgcn_s_setprio(0);
            mfma_phase(nm, acc, a, b);
            __builtin_amdgcn_s_setprio(3);
            if (!(hi && r == rounds - 1)) valu_phase<VK>(nv, s0, s1, s2, s3);
        }
    }
    else if (mode == 8) {       // in-wave interleave, same totals as mode 3: per MFMA nv*32/(nm*8) VALU ops
        int per = (nv * 4) / nm;            // VALU ops (x1) per MFMA / 1
        for (int i = 0; i < nm * rounds; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
                for (int q = 0; q < per; q += 4) {
                    if (VK == 0) { s0 = __builtin_fmaf(s0, 1.0001f, 0.5f); s1 = __builtin_fmaf(s1, 1.0001f, 0.5f);
                                   s2 = __builtin_fmaf(s2, 1.0001f, 0.5f); s3 = __builtin_fmaf(s3, 1.0001f, 0.5f); }
                    else if (VK == 1) { s0 = __builtin_amdgcn_exp2f(s0); s1 = __builtin_amdgcn_exp2f(s1);
                                        s2 = __builtin_amdgcn_exp2f(s2); s3 = __builtin_amdgcn_exp2f(s3); }
                    else { auto p = __builtin_amdgcn_cvt_pkrtz(s0, s1); auto q2 = __builtin_amdgcn_cvt_pkrtz(s2, s3);
                           s0 += (float)p[0]; s1 += (float)q2[1]; s2 += (float)p[1]; s3 += (float)q2[0]; }
                }
            }
        }
    }
    else {
        bool anti = (mode == 4) && hi;
        if (anti) valu_phase<VK>(nv, s0, s1, s2, s3);
        for (int r = 0; r < rounds; ++r) {
            mfma_phase(nm, acc, a, b);
            if (!(anti && r == rounds - 1)) valu_phase<VK>(nv, s0, s1, s2, s3);
        }
    }
    float t = s0 + s1 + s2 + s3;
    for (int u = 0; u < 8; ++u) t += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    if (t == 12345.678f) out[threadIdx.x] = t;
}

template <int VK> float run(int mode, int nm, int nv, int rounds, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<VK><<<256, 512>>>(mode, nm, nv, rounds, d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) probe<VK><<<256, 512>>>(mode, nm, nv, rounds, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5 * 1000;
}

int main() {
    float* d; hipMalloc(&d, 4096);
    const char* names[3] = {"fma", "exp2", "cvt"};
    for (int vk = 0; vk < 3; ++vk) {
        // nm MFMA-iterations (x8 MFMAs x16 cyc = 128 cyc each); nv VALU iterations (x32 ops)
        int nm = 64, rounds = 16;
        for (int nv : {32, 64, 128}) {
            float t[9];
            for (int m = 0; m < 9; ++m)
                t[m] = vk == 0 ? run<0>(m, nm, nv, rounds, d) : vk == 1 ? run<1>(m, nm, nv, rounds, d) : run<2>(m, nm, nv, rounds, d);
            printf("%-5s nv=%3d  mfma_only %7.1f us  valu_only %7.1f  concurrent %7.1f  inphase(2w) %7.1f  antiphase(2w) %7.1f | conc+prio %7.1f  conc_oldVALU %7.1f  anti+prio %7.1f  inwave %7.1f\n",
                   names[vk], nv, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8]);
        }
    }
    return 0;
}
