// naive_forward.h -- the plain fp32 forward of calibrate.hip as an inference path: precision 'f32' (UITK_PREC_F32), the last rung of the
// parity ladder (round 5).  Not part of the MFMA kernels' sources (uitk_dev.h): nothing in here is on the default path.
#pragma once
#include <hip/hip_runtime.h>

#include "uitk_dev.h"

namespace uitk {

struct NaiveHeadArgs {
    const float* norm_w; const float* norm_b;        // final LayerNorm            models/uit.py:357, 395
    const float* hln_w; const float* hln_b;          // outputlayer.0 (LayerNorm)  :358-360
    const float* head_w; const float* head_b;        // outputlayer.1 (Linear) [C][128], [C]
    int C, pooling;                                  // pooling: 0 'mean', 1 'dm', 2 'token'
    float* rowprobs;                                 // [rows][C]
    float* rowlogits;                                // [rows][C] or nullptr (never written for pooling 'dm')
};

// precision 'f32' reads the front-end's fp32 dB FRAMES ([B][nF][64], the layout of uitk_mel_db) instead of its 16-bit patch operands:
// on checkpoints with strong attention sinks the fp16 rounding of the mel operands alone is 0.8 .. 2e-3 on logits
struct NaiveFramesArgs {
    const float* frames;                             // [B][nF][64] dB, no top_db clamp yet
    int nF, ncrops, tl;                              // frames per clip, crops per clip, target_length (models/uit.py:468-481)
    int64_t row0;                                    // first clip-crop of this launch (row = clip * ncrops + crop)
};

// a: work buffers bound for a.rows clip-crops (a.stats = nullptr: forward only); nt: valid time patches per crop (GT = full clips);
// head = nullptr: stop at the last block's residual stream (the calibration / held-out uses); frames = nullptr: the patch operands a.mel
hipError_t launch_naive_forward(const CalibArgs& a, int nt, const NaiveHeadArgs* head, hipStream_t stream, const NaiveFramesArgs* frames = nullptr);

// Y[m][o] = (accumulate ? Y[m][o] : 0) + act(b[o] + sum_k X[m][k] W[o][k]) on the matrix pipe's exact-fp32 mode (calibrate.hip:
// naive_linear_mfma_kernel); K a multiple of 16; act: 0 none, 1 ReLU, 2 exact GELU; b may be null.  Used by the general engine (general.hip).
hipError_t launch_naive_linear(const float* X, const float* W, const float* b, float* Y, int M, int K, int O, int act, int accumulate, hipStream_t s);

}  // namespace uitk
