// pack.h -- packed parameter image (what uitk_create uploads; produced by pack.cpp, host only).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/uitk.h"

namespace uitk {

constexpr uint32_t IMAGE_MAGIC = 0x4b544955u;   // "UITK"

// All offsets are bytes from the start of the image, 256-byte aligned.
struct ImageHeader {
    uint32_t magic;
    int32_t precision, depth, outputdim, mel_steps;
    int32_t _mel_reserved[12];              // (was: per-16-mel-tile band tables)
    int32_t _pad[3];
    uint64_t total_bytes;
    uint64_t off_wstream, off_bias, off_bn_sc, off_bn_sh, off_pe_bias, off_fin, off_head_w, off_head_b;
    uint64_t off_window, off_tw256, off_tw512, off_mel_b;
};

// The network the kernels compute, in double: every pack-time fold applied (LayerNorm affine and the attention scale in qkv / fc1,
// residual-stream centring of everything added to the stream), nothing rounded yet.  Shared by pack_image (which rounds it into the
// kernels' operand layouts) and by the calibration forward of uitk_calibrate (calibrate.hip), so that both see the same numbers.
struct FoldedBlock {
    std::vector<double> qkv_w, qkv_b;      // [3*inner][128], [3*inner]   (q rows x 0.125 log2 e)
    std::vector<double> proj_w, proj_b;    // [128][inner], [128]         (column / bias means over the 128 outputs removed)
    std::vector<double> fc1_w, fc1_b;      // [384][128], [384]
    std::vector<double> fc2_w, fc2_b;      // [128][384], [128]           (centred like proj)
};
struct Folded {
    int depth = 0, inner = 0;
    std::vector<double> pe_w;              // [128][256], k = mel-in-patch * 16 + frame-in-patch; centred columns
    std::vector<double> pe_bias;           // [25][128]: conv bias + positional embeddings per token, row 24 = cls token; centred rows
    std::vector<FoldedBlock> blocks;
};
void fold_model(const uitk_config* c, const uitk_weights* w, Folded* out);

// Calibration statistics (uitk_calibrate / uitk_pack_weights_calibrated): UNCENTRED second moments E[x x^T] of the operand of every
// single-weight linear, doubles, in this order: patch embed [256][256]; then per block LayerNorm-1 output [128][128] (qkv),
// attention output [inner][inner] (proj), LayerNorm-2 output [128][128] (fc1), activation output [384][384] (fc2).  Round 5: behind
// them the MEAN of the LayerNorm-1 output per block, [depth][128] (operand centring, pack.cpp: centre_attention_operands).
size_t calib_stats_doubles(const uitk_config* c);
struct CalibOffsets { size_t pe, ln1, att, ln2, hid, per_block, mu1; };     // doubles; block b: ln1 + b * per_block, ...; mu1 + b * 128
CalibOffsets calib_offsets(const uitk_config* c);

int validate_config(const uitk_config* c, const char** why);
// calib: nullptr (data-free rounding: nearest, structured for the conv / fc2 weights) or calib_stats_doubles(c) doubles
int pack_image(const uitk_config* c, const uitk_weights* w, const double* calib, std::vector<char>* out, ImageHeader* hdr_out);

}  // namespace uitk
