// Verifies the operand / result lane maps of v_mfma_f32_4x4x1_16b_f32 on gfx950 that frontend.hip relies on:
//   A: lane l -> block l/4, row l%4;  B: lane l -> block l/4, col l%4;  D: lane l -> (block l/4, col l%4), VGPR r = row r.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
    f4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[64 + threadIdx.x], b[64 + threadIdx.x], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[threadIdx.x * 4 + r] = acc[r];
}
int main() {
    float ha[128], hb[128], hd[256];
    for (int i = 0; i < 128; ++i) { ha[i] = 1.0f + 0.37f * i + (i % 7); hb[i] = 2.0f - 0.11f * i + (i % 5); }
    float *a, *b, *d; hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(a, b, d); hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int blk = l / 4, j = l % 4;
        const float want = ha[4 * blk + r] * hb[4 * blk + j] + ha[64 + 4 * blk + r] * hb[64 + 4 * blk + j];
        if (fabsf(want - hd[l * 4 + r]) > 1e-3f * fabsf(want)) { if (bad < 5) printf("lane %d r %d got %f want %f\n", l, r, hd[l * 4 + r], want); ++bad; }
    }
    printf(bad ? "LAYOUT MISMATCH (%d)\n" : "layout ok (%d mismatches)\n", bad);
    return bad != 0;
}
