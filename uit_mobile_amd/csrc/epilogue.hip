// epilogue.hip -- device-side post-processing of the [B, C] probability rows (gfx950, wave64).
//
// The step AFTER the hot path: the reference ships [B, 537] to the host and runs
//   * output.topk(k)                               (inference.py:57)
//   * _output_transform_xiaoai, a per-sample Python loop   (evaluate.py:212-229)
// there.  Here one wave owns one row: the row is read once, coalesced, into <= 16 registers per lane, and only the
// k winners / the transformed row leave the chip.  HBM-bound streaming work (2 148 B per clip), no LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "uitk_dev.h"

namespace uitk {

constexpr int EP_WAVES = 4;             // waves (rows in flight) per workgroup
constexpr int EP_MAXJ = 16;             // registers per lane: rows of up to 64 * 16 = 1024 classes

struct Best { float v; int i; };

// (value, index) ordering of the reductions: larger value first, lower index on equal values (torch's CPU kernels
// return the first maximal element; topk ties are therefore "lowest index first" here)
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__device__ __forceinline__ Best wave_best(Best b) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(b.v, off, 64);
        const int oi = __shfl_xor(b.i, off, 64);
        if (better(ov, oi, b.v, b.i)) { b.v = ov; b.i = oi; }
    }
    return b;
}

// values[b][0..k) / indices[b][0..k) = the k largest entries of probs[b][:], descending
__global__ __launch_bounds__(EP_WAVES * 64) void topk_kernel(const float* __restrict__ probs, int64_t B, int C, int k,
                                                              float* __restrict__ values, int32_t* __restrict__ indices) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nj = (C + 63) >> 6;
    for (int64_t row = (int64_t)blockIdx.x * EP_WAVES + wave; row < B; row += (int64_t)gridDim.x * EP_WAVES) {
        const float* src = probs + row * C;
        float v[EP_MAXJ];
#pragma unroll
        for (int j = 0; j < EP_MAXJ; ++j) {
            const int c = lane + 64 * j;
            v[j] = (j < nj && c < C) ? src[c] : -INFINITY;
        }
        for (int t = 0; t < k; ++t) {
            Best b{-INFINITY, 0x7fffffff};
#pragma unroll
            for (int j = 0; j < EP_MAXJ; ++j)
                if (j < nj && better(v[j], lane + 64 * j, b.v, b.i)) { b.v = v[j]; b.i = lane + 64 * j; }
            b = wave_best(b);
#pragma unroll
            for (int j = 0; j < EP_MAXJ; ++j)
                if (lane + 64 * j == b.i) v[j] = -INFINITY;          // the owner retires the winner
            if (lane == 0) { values[row * k + t] = b.v; indices[row * k + t] = b.i; }
        }
    }
}

// In place on y_pred[b][:] (evaluate.py:212-229): within the first F ("filler" = AudioSet) classes keep only the
// maximal score(s); if any keyword score (classes >= F) reaches the threshold, also clear the first maximal filler so
// that the arg-max of the row is the keyword.  y_index[b] = argmax of the target row, a filler target being mapped to
// the predicted filler.
__global__ __launch_bounds__(EP_WAVES * 64) void kws_transform_kernel(float* __restrict__ y_pred, const float* __restrict__ y,
                                                                       int64_t B, int C, int F, float threshold,
                                                                       int64_t* __restrict__ y_index) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nj = (C + 63) >> 6;
    for (int64_t row = (int64_t)blockIdx.x * EP_WAVES + wave; row < B; row += (int64_t)gridDim.x * EP_WAVES) {
        float* p = y_pred + row * C;
        float v[EP_MAXJ];
        Best f{-INFINITY, 0x7fffffff};
        bool fire = false;
#pragma unroll
        for (int j = 0; j < EP_MAXJ; ++j) {
            const int c = lane + 64 * j;
            v[j] = (j < nj && c < C) ? p[c] : -INFINITY;
            if (j < nj && c < F && better(v[j], c, f.v, f.i)) { f.v = v[j]; f.i = c; }
            if (j < nj && c >= F && c < C && v[j] >= threshold) fire = true;
        }
        f = wave_best(f);
        fire = __any(fire);
#pragma unroll
        for (int j = 0; j < EP_MAXJ; ++j) {
            const int c = lane + 64 * j;
            if (j < nj && c < F) {
                // y_pred * mask: non-maximal fillers become +0 (probabilities are >= 0), maximal ones keep their value
                float o = v[j] == f.v ? v[j] : v[j] * 0.0f;
                if (fire && c == f.i) o = 0.0f;
                p[c] = o;
            }
        }
        if (y_index != nullptr && y != nullptr) {
            const float* t = y + row * C;
            Best yb{-INFINITY, 0x7fffffff};
#pragma unroll
            for (int j = 0; j < EP_MAXJ; ++j) {
                const int c = lane + 64 * j;
                if (j < nj && c < C) { const float tv = t[c]; if (better(tv, c, yb.v, yb.i)) { yb.v = tv; yb.i = c; } }
            }
            yb = wave_best(yb);
            if (lane == 0) y_index[row] = yb.i < F ? f.i : yb.i;
        }
    }
}

static int ep_grid(int64_t B) {
    const int64_t g = (B + EP_WAVES - 1) / EP_WAVES;
    return (int)(g < 1 ? 1 : g > 256 * 32 ? 256 * 32 : g);
}

hipError_t launch_topk(const float* probs, int64_t B, int C, int k, float* values, int32_t* indices, hipStream_t stream) {
    hipLaunchKernelGGL(topk_kernel, dim3(ep_grid(B)), dim3(EP_WAVES * 64), 0, stream, probs, B, C, k, values, indices);
    return hipGetLastError();
}

hipError_t launch_kws_transform(float* y_pred, const float* y, int64_t B, int C, int F, float threshold, int64_t* y_index,
                                hipStream_t stream) {
    hipLaunchKernelGGL(kws_transform_kernel, dim3(ep_grid(B)), dim3(EP_WAVES * 64), 0, stream, y_pred, y, B, C, F, threshold,
                       y_index);
    return hipGetLastError();
}

}  // namespace uitk
