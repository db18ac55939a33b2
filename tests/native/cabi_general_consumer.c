/* The GENERAL engine of libuitk.so (include/uitk.h: uitk_general_*) from plain C99 -- a UITBase geometry the MFMA kernels are not built
 * for, as a maintainer of another host language would bind it.  tests/test_gpu_general.py::test_general_engine_from_plain_c builds it
 * with gcc, hands it a config + checkpoint + clips as one binary blob and compares what it writes, bit for bit, with the Python wrapper.
 *
 *   cabi_general_consumer <blob in> <out>
 * blob: int32[16] = depth, embed_dim, num_heads, mlp_hidden, outputdim, target_length, n_mels, n_fft, hop_size, win_size, patch_size,
 *       center, attention_full, act_gelu, pooling, B; int32 T; then float32 tensors in the order of the `take` calls below (reference
 *       state_dict layouts); then int16 PCM [B][T].    out: float32 probs [B][outputdim], then mel dB [B][n_mels][frames].
 * Build: as tests/native/cabi_consumer.c. */
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
    if (argc != 3) { fprintf(stderr, "usage: %s blob out\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* blob = (char*)malloc((size_t)size);
    if (!blob || fread(blob, 1, (size_t)size, f) != (size_t)size) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    const int32_t* h = (const int32_t*)blob;
    uitk_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.depth = h[0]; cfg.embed_dim = h[1]; cfg.num_heads = h[2]; cfg.mlp_hidden = h[3]; cfg.outputdim = h[4]; cfg.target_length = h[5];
    cfg.n_mels = h[6]; cfg.n_fft = h[7]; cfg.hop_size = h[8]; cfg.win_size = h[9]; cfg.patch_size = h[10]; cfg.center = h[11];
    cfg.attention_full = h[12]; cfg.act_gelu = h[13]; cfg.pooling = h[14];
    cfg.eval_avg = UITK_AVG_MEAN; cfg.precision = UITK_PREC_F32; cfg.top_db = 120.0f;
    const int B = h[15], T = h[16];
    const size_t D = (size_t)cfg.embed_dim, P = (size_t)cfg.patch_size, inner = cfg.attention_full ? D : D / 4, H = (size_t)cfg.mlp_hidden, C = (size_t)cfg.outputdim;
    cursor = (const float*)(blob + 17 * 4);

    uitk_weights w;
    memset(&w, 0, sizeof w);
    w.window = take((size_t)cfg.win_size);
    w.mel_fb = take((size_t)(cfg.n_fft / 2 + 1) * cfg.n_mels);
    w.bn_weight = take(cfg.n_mels); w.bn_bias = take(cfg.n_mels); w.bn_mean = take(cfg.n_mels); w.bn_var = take(cfg.n_mels);
    w.pe_w = take(D * P * P); w.pe_b = take(D);
    w.time_pos = take(D * (cfg.target_length / P)); w.freq_pos = take(D * (cfg.n_mels / P));
    w.norm_w = take(D); w.norm_b = take(D);
    w.head_ln_w = take(D); w.head_ln_b = take(D);
    w.head_w = take(C * D); w.head_b = take(C);
    w.cls_token = take(D); w.token_pos = take(D);
    uitk_block_weights* blocks = (uitk_block_weights*)calloc((size_t)cfg.depth, sizeof *blocks);
    for (int b = 0; b < cfg.depth; ++b) {
        blocks[b].norm1_w = take(D); blocks[b].norm1_b = take(D);
        blocks[b].qkv_w = take(3 * inner * D); blocks[b].qkv_b = take(3 * inner);
        blocks[b].proj_w = take(D * inner); blocks[b].proj_b = take(D);
        blocks[b].norm2_w = take(D); blocks[b].norm2_b = take(D);
        blocks[b].fc1_w = take(H * D); blocks[b].fc1_b = take(H);
        blocks[b].fc2_w = take(D * H); blocks[b].fc2_b = take(D);
    }
    w.blocks = blocks;
    const int16_t* pcm = (const int16_t*)cursor;
    if ((const char*)(pcm + (size_t)B * T) != blob + size) { fprintf(stderr, "blob size does not match its header\n"); return 1; }

    /* the MFMA entry point refuses this geometry by name; the general engine takes it */
    uitk_model* mfma = NULL;
    uitk_config c16 = cfg;
    c16.precision = UITK_PREC_F16;
    if (uitk_create(&c16, &w, &mfma) != UITK_ERR_UNSUPPORTED) { fprintf(stderr, "uitk_create accepted a geometry it is not built for\n"); return 4; }
    printf("uitk_create: %s\n", uitk_last_error());
    uitk_general* m = NULL;
    CHECK_UITK(uitk_general_create(&cfg, &w, &m));
    const int nF = uitk_general_num_frames(m, T);
    void* d_pcm = NULL; float* d_probs = NULL; float* d_mel = NULL;
    const size_t n_probs = (size_t)B * C, n_mel = (size_t)B * cfg.n_mels * nF;
    CHECK_HIP(hipMalloc(&d_pcm, (size_t)B * T * 2));
    CHECK_HIP(hipMalloc((void**)&d_probs, n_probs * 4));
    CHECK_HIP(hipMalloc((void**)&d_mel, n_mel * 4));
    CHECK_HIP(hipMemcpy(d_pcm, pcm, (size_t)B * T * 2, hipMemcpyHostToDevice));
    CHECK_UITK(uitk_general_reserve(m, B, T));
    CHECK_UITK(uitk_general_mel_db(m, d_pcm, UITK_IN_I16, B, T, d_mel, NULL));
    CHECK_UITK(uitk_general_forward(m, d_pcm, UITK_IN_I16, B, T, d_probs, NULL, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float* out = (float*)malloc((n_probs + n_mel) * 4);
    CHECK_HIP(hipMemcpy(out, d_probs, n_probs * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + n_probs, d_mel, n_mel * 4, hipMemcpyDeviceToHost));
    f = fopen(argv[2], "wb");
    if (!f || fwrite(out, 4, n_probs + n_mel, f) != n_probs + n_mel) { perror(argv[2]); return 1; }
    fclose(f);
    printf("general engine: %d clips x %d samples -> %d frames, %d crops, probs[0][0] = %.6f\n", B, T, nF, uitk_general_num_crops(m, T), out[0]);
    CHECK_UITK(uitk_general_destroy(m));
    (void)hipFree(d_pcm); (void)hipFree(d_probs); (void)hipFree(d_mel);
    free(out); free(blocks); free(blob);
    return 0;
}
