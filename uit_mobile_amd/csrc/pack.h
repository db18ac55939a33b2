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

int validate_config(const uitk_config* c, const char** why);
int pack_image(const uitk_config* c, const uitk_weights* w, std::vector<char>* out, ImageHeader* hdr_out);

}  // namespace uitk
