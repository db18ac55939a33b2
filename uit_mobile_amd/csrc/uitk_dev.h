// uitk_dev.h -- kernel argument blocks and layout constants shared by the host code and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uitk {

constexpr int AMAX_SLOTS = 64;          // == UITK_AMAX_SLOTS

// ---- fixed geometry of the uit_* family (models/uit.py:581-635 with target_length 96..111)
constexpr int ENC_TIMING_SLOTS = 20;   // per-wave phase cycle sums of the diagnostic encoder build (uitk_set_timing_buffers)
constexpr int DIM = 128;                // embed_dim
constexpr int NTOK = 24;                // 4 freq patches x 6 time patches, token = f*6 + t (uit.py:388)
constexpr int GT = 6, GF = 4;
constexpr int INNER = 32;               // BNeckAttention inner_dim = dim/4 (uit.py:101), 2 heads x 16
constexpr int HID = 384;                // mlp hidden
constexpr int NMEL = 64;
constexpr int MEL_PROW = 272;           // row stride (floats) of the front-end's zero-padded power rows: every mel bin window ends inside it
constexpr int MEL_BMAX = 52;            // filterbank steps the front-end keeps on chip (its fast path always runs this many)

// ---- weight stream: 16 KiB stages of sixteen 1 KiB MFMA A/B fragments (see pack.cpp / encoder.hip)
constexpr int FRAG_BYTES = 1024;        // 64 lanes x 8 x 16-bit
constexpr int STAGE_FRAGS = 16;
constexpr int STAGE_BYTES = STAGE_FRAGS * FRAG_BYTES;
constexpr int PE_HI_STAGES = 4;         // patch-embed: 8 out tiles x 8 k-steps of 16-bit weights ...
constexpr int PE_STAGES = 8;            // ... plus the same 64 fragments of the weights' low-order terms (W = hi + lo), used only with
                                        // UITK_SPLIT_PE=1: pack.cpp's structured rounding makes them unnecessary
constexpr int BLOCK_STAGES = 14;        // QK, V+proj, 12 MLP chunks
constexpr int TAIL_STAGES = 4;          // zero stages after the last block: the ring prefetch (<= 4 ahead) never branches
// ---- fp8 weight stream (precision fp8: qkv / fc1 / fc2 as OCP e4m3 operands of v_mfma_scale_f32_16x16x128_f8f6f4)
// A "fragment8" is the 2 KiB A/B operand of one 16x16x128 MFMA, stored as two lane-linear 1 KiB halves (bytes 0..15 and
// 16..31 of every lane).  Lane l = (row/col c = l&15, g = l>>4) byte j  <->  W[row][ base + 16*(j>>2) + 4*g + (j&3) ]: the k-order
// a stack of eight C/D tiles has after in-register conversion (4 bytes = one tile's 4 rows).  MX block scales (E8M0): the
// hardware's 32-element scale block beta (0..3) of a row is {lanes g = 2(beta&1), 2(beta&1)+1} x {bytes 16(beta>>1) .. +15}, and
// the scale byte is supplied by lane (row, g' = beta)  (tools/probe/mx_layout.hip).
// Stages of one block: 0 qkv (6 fragment8 + scale dwords at FP8_QKV_SCALES), 1 proj (8 f16 fragments) + fc1/fc2 scale
// dwords, then for q = 0..2: fc1 tiles 8q..8q+7, fc2 fragments (ft = 0..7, k-group q).
// ---- full-width Attention (models/uit.py:124-178): qkv 128->384 = (q|k|v) x (head 0|1) x 64, proj 128->128.
// 20 stages per block: for head h = 0, 1: [q_h: d-tile dt = f/4, k-step s = f%4] [k_h] [v_h, used as the B operand];
// 2 stages proj (fragment ft*4 + s); 12 MLP stages as in the bottleneck layout.  Bias block (exactly BIAS_FLOATS):
constexpr int BLOCK_STAGES_ATT = 20;
constexpr int BIASF_QKV = 0, BIASF_PROJ = 384, BIASF_FC1 = 512, BIASF_FC2 = 896;
constexpr int BLOCK_STAGES_FP8 = 8;
// ---- precision f16x2 (W = hi + lo, both fp16): 28 stages per block; every stage holds the hi AND the lo fragments of the rows
// it covers, so that an accumulator chain (4 or 8 hi MFMAs + as many lo MFMAs) completes inside one stage:
//   0: q h0, q h1 (per row tile: hi s0..3, lo s0..3)   1: k h0, k h1   2: v h0, v h1 (B operand)   3: proj ft 0..7 hi, then lo
//   4 + 2ch: fc1 row tiles 2ch, 2ch+1 (hi s0..3, lo s0..3 each)        5 + 2ch: fc2 k-step ch, ft 0..7 hi, then lo
constexpr int BLOCK_STAGES_X2 = 28;
// f16x2 with full-width Attention (round 3): 40 stages per block.  A q / k / v row group of a head (4 d-tiles) takes TWO stages of two
// tiles each -- tile slot dl: fragments dl*8 + s (hi k-step s) and dl*8 + 4 + s (lo) -- in the order of the f16 layout:
//   2 (3h + which) + sub, which = q, k, v;   12 + i: proj output tiles 2i, 2i+1 (hi s0..3, lo s0..3 each);   16 + 2ch / 17 + 2ch: MLP as above
constexpr int BLOCK_STAGES_X2_ATT = 40;
// ---- precision f16l6 (W = f16 hi + MX-fp6 lo; round 3): the 14 stages of precision f16, the MLP / proj stages followed by the
// low-order fragments of the rows they cover, in a ring of four 32 KiB slots (stage sizes differ; every DMA is static per call site).
// A "fragment6" is the 1536-byte A operand of one v_mfma_scale_f32_16x16x128_f8f6f4 with e2m3 data: lane l = (row c = l&15,
// g = l>>4) holds 32 six-bit fields, field i at bits [6i, 6i+6) of its 192 bits; field i <-> k = kbase + 32 (i>>3) + 16 ((i&7)>>2)
// + 4 g + (i&3): the order v_cvt_scalef32_pk32_fp6_f16 gives the 32 halves of a lane's four k-step operands (and four
// v_cvt_scalef32_pk_fp4_f16 per chunk give the hidden activations of four MLP chunks) -- tools/probe/mx6_layout.hip.  Stored as a
// lane-linear 1 KiB part (dwords 0..3 of every lane) followed by a lane-linear 512-byte part (dwords 4, 5).  One E8M0 scale per
// ROW of a fragment6 (all four lane groups use it), four fragments per dword, in the block's bias table (L6_SCALES).
//   QK : 16 f16 fragments (qkv keeps single fp16 weights)                                         16 KiB
//   VP : 8 V + 8 proj f16 fragments | 8 proj low-order f16 fragments                              24 KiB
//   M  : 8 fc1 + 8 fc2 f16 fragments | fragment6 of the two fc1 row tiles                         19 KiB
//   MB : M | fragment6 ft = 0..7 of fc2's hidden k-group q (every fourth MLP stage)               31 KiB
constexpr int FRAG6_BYTES = 1536;
constexpr int L6_SLOT_BYTES = 32768;
constexpr int L6_QK_BYTES = STAGE_BYTES;
constexpr int L6_VP_BYTES = STAGE_BYTES + 8 * FRAG_BYTES;
constexpr int L6_M_BYTES = STAGE_BYTES + 2 * FRAG6_BYTES;
constexpr int L6_MB_BYTES = L6_M_BYTES + 8 * FRAG6_BYTES;
constexpr int L6_BLOCK_BYTES = L6_QK_BYTES + L6_VP_BYTES + 9 * L6_M_BYTES + 3 * L6_MB_BYTES;
constexpr int L6_TAIL_BYTES = L6_QK_BYTES + L6_VP_BYTES;         // two dummy stages: the prefetch of the last block never branches
// byte offset of MLP stage ch (0..11) inside a block
constexpr int l6_mlp_offset(int ch) { return L6_QK_BYTES + L6_VP_BYTES + ch * L6_M_BYTES + (ch / 4) * (L6_MB_BYTES - L6_M_BYTES); }
constexpr int L6_SCALES = 736;          // bias-table dwords [736, 928): [quad][row] of E8M0 bytes; quads 0..5: fc1 row tiles 4 quad ..;
                                        // quads 6..11: fc2 fragment6 8 q + ft = 4 (quad - 6) ..
constexpr int FRAG8_BYTES = 2048;
constexpr int FP8_QKV_SCALES = 6 * FRAG8_BYTES;                 // 2 dwords per lane: bytes = tiles 0..3 | 4, 5
constexpr int FP8_MLP_SCALES = 8 * FRAG_BYTES;                  // stage 1: 6 dwords per lane for fc1 tiles, then 6 for fc2 (q*8 + ft)
constexpr int BIAS_FLOATS = 1024;       // per block: qkv 96 | proj 128 | fc1 384 | fc2 128 (padded to 4 KiB)
constexpr int BIAS_QKV = 0, BIAS_PROJ = 96, BIAS_FC1 = 224, BIAS_FC2 = 608;
// ---- operand centring of LayerNorm-1 (round 5): instantiations that subtract the calibration mean mu of the LayerNorm-1
// output from it BEFORE the fp16 rounding (the linear is exact for any mu: W (x - mu) + (b + W mu)) find -mu[128] in their block's
// bias table: BNeckAttention at BIAS_NEGMU (the table's spare tail), full-width attention in the K rows' bias slot BIASF_QKV + 128 (with
// the centring the K and V biases are exactly zero: k - E[k] leaves the softmax unchanged, v - E[v] moves into the proj bias).
constexpr int BIAS_NEGMU = 736;
// which (attention, activation, pooling, kernel precision) builds carry the runtime centring: the variants whose fp16 ACTIVATION floor sat
// at the gate in round 4 (full-width attention, GELU, pooling 'token'), on fp16 operands with single or split weights
constexpr bool operand_centring(int att, int act, int pool, int kernel_prec) {
#if defined(UITK_LAB) && defined(UITK_CEN_ALL)
    return kernel_prec == 0 || kernel_prec == 3 || kernel_prec == 5 || kernel_prec == 6;      // lab: the runtime centring for the default family too
#else
    return (kernel_prec == 0 || kernel_prec == 3 || kernel_prec == 5 || kernel_prec == 6) && (att != 0 || act != 0 || pool == 2);      // (5 / 6: precisions f16a / f16ax)
#endif
}

struct FrontendArgs {
    const void* pcm;        // [B][T] f32 or i16
    int64_t B, T;
    int nF, hop;
    int pad;                // samples of reflect padding in front of frame 0: n_fft/2 (center=True) or 0 (center=False: no frame leaves the clip)
    const float* window;    // [512]
    const float2* tw256;    // [16][16]  W256^(p*k1) = (cos, -sin)(2 pi p k1 / 256)
    const float2* tw512;    // [128]     (cos, sin)(2 pi k / 512)
    const float* mel_b;     // banded filterbank for v_mfma_f32_4x4x1: [64 x int32 first bin of the lane's mel group, a
                            // multiple of 4] then [mel_steps/4][64 lanes][4 steps] B operands (x 1/4)
    int mel_steps;          // S: bins every 4-mel group walks (multiple of 4)
    float* mel;             // PATCH layout: 16-bit MFMA operands [B*ncrops][24 tokens][16 mels][16 frames] holding
                            // init_bn(dB) WITHOUT the top_db clamp (the encoder clamps); FRAMES layout (uitk_mel_db
                            // only): fp32 dB [B][nF][64]
    const float* bn_sc;     // [64] init_bn scale / shift (PATCH layout only)
    const float* bn_sh;
    int op_bf16;            // operand type of the PATCH layout: 0 = fp16, 1 = bf16
    unsigned long long* timing;  // -DUITK_LAB builds: [waves][8] phase cycle sums (null = off; the product build rejects non-null)
    int ncrops, tl, nt;     // crop plan (models/uit.py:468-481): crops per clip, target_length, valid time patches
    float* amax;            // [AMAX_SLOTS] filled by this launch (pre-armed with 0xFFFFFFFF)
    float* amax_rearm;      // [AMAX_SLOTS] the other set: re-armed by this launch for the next one
    int asym_pm;            // per-mille share of the passes for the first-dispatched half of the grid (500 = symmetric, -1 = not calibrated: 590)
    float fb_max;           // largest filterbank entry (> 0): the Parseval bound that lets leftover frames which cannot raise the batch amax
                            // skip their FFT (frontend.hip); 0 disables the skip
};

struct EncoderArgs {
    const unsigned short* mel;  // [rows][24][16][16] patch-major 16-bit operands: init_bn(dB), unclamped
    const float* amax;      // [AMAX_SLOTS]
    float top_db;
    int rows;               // B * ncrops
    int nt;                 // valid time patches per crop (1..6)
    int depth;
    int C;                  // outputdim
    const char* wstream;    // PE_STAGES + depth*BLOCK_STAGES + TAIL_STAGES stages of STAGE_BYTES
    const float* bias;      // [depth][BIAS_FLOATS]
    const float* bn_sc;     // [64]  w / sqrt(var + 1e-5)
    const float* bn_sh;     // [64]  b - mean * sc
    const float* pe_bias;   // [25][128] conv bias + time_pos + freq_pos (rows 0..23), cls_token + token_pos_embed (row 24); centred
    const float* fin;       // [4][128] norm.w, norm.b, head_ln.w, head_ln.b
    const char* head_w;     // [ceil(C/16)][4] fragments
    const float* head_b;    // [ceil(C/16)*16]
    float* probs;           // [rows][C]
    float* logits;          // [rows][C] or null
    float* dbg;             // [rows][depth+1][24][128] or null
    unsigned long long* timing;  // -DUITK_LAB builds: [waves][ENC_TIMING_SLOTS] phase cycle sums (null = off; the product build rejects non-null)
    int attention_full;     // 0 = BNeckAttention, 1 = Attention (full width, 2 heads x 64)
    int act_gelu;           // MLP activation: 0 = ReLU, 1 = exact (erf) GELU
    int pooling;            // 0 = mean over the tokens, 1 = 'dm' (per time patch: mean over frequency patches -> head -> sigmoid; mean over
                            // time), 2 = 'token' (a cls token in column 0 of a ONE-clip wave; its row feeds the head)
    int bn_nonneg;          // 1 = every init_bn scale w/sqrt(var+eps) is >= 0 (the usual case): the top_db clamp is a plain max
    int split_pe;           // 1 = patch embed on split weights (W = hi + lo, 8-wave workgroups only); 0 = high-order term only
};

// calibrate.hip (pack time only): the plain fp32 forward of the folded network (pack.h: Folded) that measures the operand statistics
// of the calibrated rounding.  Every pointer is device memory; `blocks` holds per block the eight folded arrays as floats at the
// o_* offsets; `stats` receives the second moments as floats in the layout of pack.h: calib_offsets (off_* / per_block in floats).
struct CalibArgs {
    const unsigned short* mel;  // [rows][24][256] the front-end's patch operands (full clips: every time patch valid)
    const float* amax;          // [AMAX_SLOTS] of the same front-end launch
    float top_db;
    const float* bn_sc; const float* bn_sh;
    int op_bf16;
    int rows, ntok, depth, inner, act_gelu;       // ntok: 24, or 25 with the cls token of pooling 'token' in front
    const float* pe_w;          // [128][256]
    const float* pe_bias;       // [25][128]
    const float* blocks;        // [depth][block_floats]
    size_t block_floats, o_qkv_w, o_qkv_b, o_proj_w, o_proj_b, o_fc1_w, o_fc1_b, o_fc2_w, o_fc2_b;
    float* X0; float* R; float* Xn; float* QKV; float* O; float* Hd;      // work buffers [rows*ntok][256 | 128 | 128 | 3 inner | inner | 384]
    float* stats;
    size_t off_pe, off_ln1, off_att, off_ln2, off_hid, per_block;
    size_t off_mu1;             // [depth][128] column means of the LayerNorm-1 output (fp32 values, before operand rounding)
    int centre_ln1;             // builds with the runtime operand centring (operand_centring()): the LayerNorm-1 moment is that of the operand the
                                // kernel forms, round16(x - mu), not of round16(x)
};
hipError_t launch_calibration(const CalibArgs& a, hipStream_t stream);

hipError_t launch_frontend(const FrontendArgs& a, int in_i16, int patch_layout, int n_cu, int wg_per_cu, hipStream_t stream);
hipError_t launch_mel_export(const float* mel, const float* amax, float top_db, int64_t B, int nF, float* out,
                             hipStream_t stream);
hipError_t launch_encoder(const EncoderArgs& a, int precision, int waves, hipStream_t stream);
hipError_t launch_crop_reduce(const float* rowprobs, float* probs, int64_t B, int ncrops, int C, int eval_avg,
                              hipStream_t stream);
// epilogue.hip: device-side post-processing of the probability rows (C <= 1024, k <= C)
hipError_t launch_topk(const float* probs, int64_t B, int C, int k, float* values, int32_t* indices, hipStream_t stream);
hipError_t launch_kws_transform(float* y_pred, const float* y, int64_t B, int C, int F, float threshold, int64_t* y_index,
                                hipStream_t stream);

}  // namespace uitk
