/* A consumer of libuitk.so written in plain C99 (no Python, no torch, no C++): what a maintainer of another host language binds against.
 * tests/test_gpu_parity.py::test_c_abi_consumer_in_plain_c builds it with gcc, hands it a checkpoint + clips as one binary blob and
 * compares what it writes, bit for bit, with the Python wrapper's output for the same weights and clips.
 *
 *   cabi_consumer <blob in> <out> [calib_budget_e6]
 * blob: int32 depth, outputdim, B, T; then float32 tensors in the order of `take` calls below (reference state_dict layouts,
 * include/uitk.h); then int16 PCM [B][T].   out: float32 probs [B][outputdim], logits [B][outputdim], calibration check [5].
 *
 * Build:  gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/native/cabi_consumer.c \
 *             -Luit_mobile_amd -luitk -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/uit_mobile_amd -Wl,-rpath,/opt/rocm/lib
 * (the HIP runtime API is used for device buffers only: hipMalloc / hipMemcpy / hipDeviceSynchronize.) */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "uitk.h"

static const float* cursor;
static const float* take(size_t n) { const float* p = cursor; cursor += n; return p; }

#define CHECK_UITK(call)                                                                         \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ != UITK_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, uitk_last_error()); return 2; } \
    } while (0)
#define CHECK_HIP(call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 3; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 3 && argc != 4) { fprintf(stderr, "usage: %s blob out [calib_budget_e6]\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* blob = (char*)malloc((size_t)size);
    if (!blob || fread(blob, 1, (size_t)size, f) != (size_t)size) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    const int32_t* hdr = (const int32_t*)blob;
    const int depth = hdr[0], C = hdr[1], B = hdr[2], T = hdr[3];
    cursor = (const float*)(blob + 16);

    uitk_weights w;
    memset(&w, 0, sizeof w);
    w.window = take(512);
    w.mel_fb = take(257 * 64);
    w.bn_weight = take(64); w.bn_bias = take(64); w.bn_mean = take(64); w.bn_var = take(64);
    w.pe_w = take(128 * 256); w.pe_b = take(128);
    w.time_pos = take(128 * 6); w.freq_pos = take(128 * 4);
    w.norm_w = take(128); w.norm_b = take(128);
    w.head_ln_w = take(128); w.head_ln_b = take(128);
    w.head_w = take((size_t)C * 128); w.head_b = take((size_t)C);
    w.cls_token = take(128); w.token_pos = take(128);
    uitk_block_weights* blocks = (uitk_block_weights*)calloc((size_t)depth, sizeof *blocks);
    for (int b = 0; b < depth; ++b) {
        blocks[b].norm1_w = take(128); blocks[b].norm1_b = take(128);
        blocks[b].qkv_w = take(96 * 128); blocks[b].qkv_b = take(96);
        blocks[b].proj_w = take(128 * 32); blocks[b].proj_b = take(128);
        blocks[b].norm2_w = take(128); blocks[b].norm2_b = take(128);
        blocks[b].fc1_w = take(384 * 128); blocks[b].fc1_b = take(384);
        blocks[b].fc2_w = take(128 * 384); blocks[b].fc2_b = take(128);
    }
    w.blocks = blocks;
    const int16_t* pcm = (const int16_t*)cursor;
    if ((const char*)(pcm + (size_t)B * T) != blob + size) { fprintf(stderr, "blob size does not match its header\n"); return 1; }

    uitk_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.depth = depth; cfg.embed_dim = 128; cfg.num_heads = 2; cfg.mlp_hidden = 384; cfg.outputdim = C; cfg.target_length = 102;
    cfg.n_mels = 64; cfg.n_fft = 512; cfg.hop_size = 160; cfg.win_size = 512; cfg.patch_size = 16; cfg.center = 1;
    cfg.eval_avg = UITK_AVG_MEAN; cfg.precision = UITK_PREC_F16; cfg.top_db = 120.0f;
    cfg.attention_full = 0; cfg.act_gelu = 0; cfg.pooling = 0; cfg.fe_autotune = 0;

    /* THE PARITY MODE as a C consumer builds it (what the Python wrapper's precision 'auto' does): create in f16, calibrate on the library's
     * own signals, then ask the handle whether the image it kept is inside the absolute budget on the held-out clips -- if not, rebuild in
     * the precision it names and ask again: split weights (f16l6 / f16x2; pooling 'token': f16a, f16ax), and last UITK_PREC_F32, the plain
     * fp32 forward, which has nothing to calibrate and nothing above it.  (The wrapper also measures each rung with uitk_self_check and skips
     * one that is no better; a consumer that simply follows the handle ends on the same last rung.)
     * argv[3] (optional): a calib_budget_e6 value, e.g. 1 = a budget no fp16 mode meets (the whole ladder). */
    uitk_model* m = NULL;
    CHECK_UITK(uitk_create(&cfg, &w, &m));
    if (argc > 3) CHECK_UITK(uitk_set_option(m, "calib_budget_e6", atoi(argv[3])));
    CHECK_UITK(uitk_calibrate_weights(m, &w, NULL, UITK_IN_I16, 0, 0, NULL));
    float check[5];
    CHECK_UITK(uitk_get_calib_check(m, check));
    int escalate_to = -1;
    CHECK_UITK(uitk_get_option(m, "parity_escalate_to", &escalate_to));
    for (int step = 0; step < 3 && escalate_to >= 0; ++step) {
        float self[2];
        const int to = escalate_to;
        CHECK_UITK(uitk_destroy(m));
        cfg.precision = to;
        CHECK_UITK(uitk_create(&cfg, &w, &m));
        escalate_to = -1;
        if (to != UITK_PREC_F32) {
            if (argc > 3) CHECK_UITK(uitk_set_option(m, "calib_budget_e6", atoi(argv[3])));
            CHECK_UITK(uitk_calibrate_weights(m, &w, NULL, UITK_IN_I16, 0, 0, NULL));      /* split weights: the operand centring + the verdict on this build */
            CHECK_UITK(uitk_get_option(m, "parity_escalate_to", &escalate_to));
        }
        CHECK_UITK(uitk_self_check(m, &w, NULL, self));
        printf("over budget (or refused): rebuilt in precision %d, self check max %.2e rms %.2e\n", to, self[0], self[1]);
    }

    void *d_pcm = NULL, *d_probs = NULL, *d_logits = NULL;
    const size_t n_out = (size_t)B * C;
    CHECK_HIP(hipMalloc(&d_pcm, (size_t)B * T * 2));
    CHECK_HIP(hipMalloc(&d_probs, n_out * 4));
    CHECK_HIP(hipMalloc(&d_logits, n_out * (size_t)uitk_num_crops(m, T) * 4));
    CHECK_HIP(hipMemcpy(d_pcm, pcm, (size_t)B * T * 2, hipMemcpyHostToDevice));
    CHECK_UITK(uitk_forward(m, d_pcm, UITK_IN_I16, B, T, (float*)d_probs, (float*)d_logits, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float* out = (float*)malloc((2 * n_out + 5) * 4);
    CHECK_HIP(hipMemcpy(out, d_probs, n_out * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + n_out, d_logits, n_out * 4, hipMemcpyDeviceToHost));
    memcpy(out + 2 * n_out, check, sizeof check);
    f = fopen(argv[2], "wb");
    if (!f || fwrite(out, 4, 2 * n_out + 5, f) != 2 * n_out + 5) { perror(argv[2]); return 1; }
    fclose(f);
    printf("%s: %d clips x %d samples -> probs[0][0..2] = %.6f %.6f %.6f; calibration check: nearest %.2e, calibrated %.2e, accepted %d\n",
           uitk_version(), B, T, out[0], out[1], out[2], check[0], check[1], (int)check[4]);
    CHECK_UITK(uitk_destroy(m));
    (void)hipFree(d_pcm); (void)hipFree(d_probs); (void)hipFree(d_logits);
    free(blocks); free(out); free(blob);
    return 0;
}
