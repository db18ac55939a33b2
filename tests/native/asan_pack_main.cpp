// Sanitizer driver for the host-only packer (csrc/pack.cpp): test infrastructure.  Built by tests/test_sanitizers.py with
// g++ -fsanitize=address,undefined (GPU AddressSanitizer is not available on the pool; the packer is plain host C++).
// Packs random weights for every precision / attention layout / activation / pooling the configuration check admits and checks the
// image size against the header; for every precision the calibrated entry (GPTQ on the fp16 / MX e4m3 grid, operand centring from the
// LayerNorm-1 means, worker threads) from random rank-deficient statistics; and the refusal of non-finite statistics.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>
#include "../../uit_mobile_amd/csrc/pack.h"

int main() {
    std::vector<float> pool(1 << 20);
    for (size_t i = 0; i < pool.size(); ++i) pool[i] = (float)((double)rand() / RAND_MAX - 0.5) * 0.2f;
    for (float& v : pool) if (v == 0.f) v = 0.01f;
    const float* p = pool.data();
    int fails = 0, packed = 0, calibrated = 0, refused_cfg = 0;
    auto psd = [&](double* H, int K) {          // H = X^T X / n of n = 48 random rows (rank-deficient for K > 48: the damping's job)
        std::vector<double> X((size_t)48 * K);
        for (double& v : X) v = (double)rand() / RAND_MAX - 0.3;
        for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) { double a = 0; for (int n = 0; n < 48; ++n) a += X[(size_t)n * K + i] * X[(size_t)n * K + j]; H[(size_t)i * K + j] = a / 48; }
    };
    for (int full = 0; full < 2; ++full)
        for (int gelu = 0; gelu < 2; ++gelu)
            for (int pooling = 0; pooling < 3; ++pooling)
                for (int prec = 0; prec <= UITK_PREC_F16AX; ++prec)
                    for (int depth : {1, 2}) {
                        uitk_config c{};
                        c.depth = depth; c.embed_dim = 128; c.num_heads = 2; c.mlp_hidden = 384; c.outputdim = 537; c.target_length = 102;
                        c.n_mels = 64; c.n_fft = 512; c.hop_size = 160; c.win_size = 512; c.patch_size = 16; c.center = 1; c.eval_avg = 0;
                        c.precision = prec; c.top_db = 120.f; c.attention_full = full; c.act_gelu = gelu; c.pooling = pooling;
                        std::vector<uitk_block_weights> blocks(depth);
                        for (auto& b : blocks) {
                            b.norm1_w = p; b.norm1_b = p + 128; b.qkv_w = p + 1000; b.qkv_b = p + 60000; b.proj_w = p + 70000; b.proj_b = p + 90000;
                            b.norm2_w = p + 256; b.norm2_b = p + 384; b.fc1_w = p + 100000; b.fc1_b = p + 160000; b.fc2_w = p + 170000; b.fc2_b = p + 230000;
                        }
                        uitk_weights w{};
                        w.window = p; w.mel_fb = p + 300000; w.bn_weight = p + 512; w.bn_bias = p + 576; w.bn_mean = p + 640;
                        std::vector<float> var(64, 300.f); w.bn_var = var.data();
                        w.pe_w = p + 400000; w.pe_b = p + 700; w.time_pos = p + 2000; w.freq_pos = p + 4000; w.norm_w = p + 5000; w.norm_b = p + 5200;
                        w.head_ln_w = p + 5400; w.head_ln_b = p + 5600; w.head_w = p + 500000; w.head_b = p + 8000; w.blocks = blocks.data(); w.cls_token = p + 9000; w.token_pos = p + 9200;
                        const char* why = "";
                        if (uitk::validate_config(&c, &why)) { ++refused_cfg; continue; }       // (a combination the library does not build)
                        std::vector<char> img; uitk::ImageHeader h{};
                        if (uitk::pack_image(&c, &w, nullptr, &img, &h) || img.size() != h.total_bytes || h.magic != uitk::IMAGE_MAGIC) { std::printf("pack failed: full %d gelu %d pooling %d prec %d depth %d\n", full, gelu, pooling, prec, depth); ++fails; continue; }
                        ++packed;
                        if (depth != 2 || (pooling == 1 && prec != 0)) continue;            // (calibrated: one depth, 'dm' once)
                        std::vector<double> st(uitk::calib_stats_doubles(&c));
                        const uitk::CalibOffsets co = uitk::calib_offsets(&c);
                        const int inner = full ? 128 : 32;
                        psd(st.data() + co.pe, 256);
                        for (int b = 0; b < depth; ++b) {
                            double* s = st.data() + (size_t)b * co.per_block;
                            psd(s + co.ln1, 128); psd(s + co.att, inner); psd(s + co.ln2, 128); psd(s + co.hid, 384);
                            for (int k = 0; k < 128; ++k) st[co.mu1 + (size_t)b * 128 + k] = 0.3 * ((double)rand() / RAND_MAX - 0.5);   // |mean| <= sqrt(E[x^2]) need not hold for random numbers: the packer must cope
                        }
                        std::vector<char> img2; uitk::ImageHeader h2{};
                        if (uitk::pack_image(&c, &w, st.data(), &img2, &h2) || img2.size() != img.size() || img2 == img) { std::printf("calibrated pack failed / unchanged: full %d gelu %d pooling %d prec %d\n", full, gelu, pooling, prec); ++fails; }
                        ++calibrated;
                        if (prec == 0 && pooling == 0) {        // non-finite / non-positive statistics are refused with an error code, not sorted
                            for (double poison : {std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::infinity(), -1.0}) {
                                std::vector<double> bad(st);
                                bad[co.ln2 + 5 * 128 + 5] = poison;
                                std::vector<char> img3; uitk::ImageHeader h3{};
                                if (uitk::pack_image(&c, &w, bad.data(), &img3, &h3) != UITK_ERR_INVALID_ARG) { std::printf("poisoned statistics (%g) accepted: full %d gelu %d\n", poison, full, gelu); ++fails; }
                            }
                        }
                    }
    std::printf("pack sanitizer run: %d images, %d calibrated, %d configurations refused, %d failures\n", packed, calibrated, refused_cfg, fails);
    return fails;
}
