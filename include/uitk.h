/*
 * uitk.h -- C ABI of libuitk.so: MI355X-native batched inference for the UiT audio models
 *           (log-mel front-end + UiT encoder + head), the hot path of RicherMans/UIT_Mobile.
 *
 * The reference has no FFI: its boundary for this path is the Python nn.Module API of models/uit.py
 * (factories uit_xs/uit_xxs/uit_xxxs :581-635, UITBase.__init__ :253-363, load_state_dict :416-423,
 * forward :452-493), called from inference.py:55 and evaluate.py:65.  The Python wrapper in
 * uit_mobile_amd/models/uit.py keeps that API and calls these entry points through ctypes; nothing else
 * does.  Each entry point names the reference interface it stands in for.
 *
 * Two engines behind one header.  Handles of type uitk_model: the MFMA kernels, specialised to the geometry of the published
 * factories (embed 128, 2 heads, MLP x 3, 64 mels, n_fft 512, 16 x 16 patches, six time patches) -- the fast path every BASELINE
 * config runs on.  Handles of type uitk_general -- the uitk_general_* entry points at the end of this file -- plain fp32 for every
 * OTHER geometry the reference's UITBase constructor admits.  uitk_create refuses what it is not built for by name
 * (UITK_ERR_UNSUPPORTED).
 *
 * Conventions: plain pointers and sizes, int return codes (0 = ok, no exceptions across the boundary),
 * caller owns input/output device buffers, the model owns packed weights + workspace, every call is
 * enqueue-only on the HIP stream passed in (no host synchronisation) once the workspace is reserved.
 * One model handle per device; a handle is not re-entrant.  There is NO CPU implementation behind this
 * ABI: without a HIP device uitk_create() fails with UITK_ERR_NO_DEVICE.
 */
#ifndef UITK_H
#define UITK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UITK_OK 0
#define UITK_ERR_INVALID_ARG (-1)   /* null pointer, bad size */
#define UITK_ERR_UNSUPPORTED (-2)   /* configuration outside the uit_* family this library is built for */
#define UITK_ERR_NO_DEVICE (-3)     /* no HIP device / HIP runtime error at create */
#define UITK_ERR_HIP (-4)           /* HIP runtime error (see uitk_last_error) */
#define UITK_ERR_TOO_SHORT (-5)     /* clip shorter than n_fft/2+1 samples (reflect pad) or than one patch */
#define UITK_ERR_WORKSPACE (-6)     /* workspace too small and growth not allowed (stream capture) */

typedef enum uitk_in_dtype {
    UITK_IN_F32 = 0, /* float32 in [-1,1): what torchaudio.load / the collate hand the model (inference.py:52) */
    UITK_IN_I16 = 1  /* int16 PCM as stored (dataset.py:44-46); the x(1/32768) is fused into the load       */
} uitk_in_dtype;

typedef enum uitk_precision {
    UITK_PREC_F16 = 0,  /* fp16 MFMA operands, fp32 accumulate/residual/LN/softmax.  With uitk_calibrate_weights (calibrated rounding of
                           the weights, round 4) THE PARITY MODE: worst 4.4e-4 on logits over the trained-like sweep; with the weights
                           rounded to nearest (no calibration call): 1e-3 up to |logit| ~ 12, 0.6 .. 1.2e-3 at |logit| 13-15 */
    UITK_PREC_BF16 = 1, /* bf16 MFMA operands, same structure (same MFMA rate, ~8x the operand rounding) */
    UITK_PREC_FP8 = 2,  /* BASELINE config #5: qkv / fc1 / fc2 weights AND their input activations as OCP e4m3 with MX block
                           scales on v_mfma_scale_f32_16x16x128_f8f6f4 (2x the f16 MFMA rate); patch embed, attention
                           internals, proj and the head stay f16.  Far outside the 1e-3 gate: a measured throughput mode */
    UITK_PREC_F16X2 = 3,/* every weight of the path as TWO fp16 terms (W = hi + lo, 22 significant bits, two MFMAs per fragment),
                           activations fp16 as in UITK_PREC_F16: the reference point for what weight splitting can buy, at ~1.7x the
                           encoder time.  Any attention / activation / pooling (round 3): it is the parity mode of the
                           full-width-attention / GELU factories (audio_transformer_*), which have no UITK_PREC_F16L6 build */
    UITK_PREC_FP8W = 5, /* round 5: BASELINE config #5 read literally -- "fp8 WEIGHTS": the qkv / fc1 / fc2 weights on the OCP e4m3 grid with MX
                           block scales (32 consecutive k per scale), expanded to fp16 fragments at pack time: the UITK_PREC_F16 kernel,
                           fp16 activations, any family / pooling.  With uitk_calibrate_weights the grid rounding is GPTQ (as for
                           UITK_PREC_FP8).  A measured mode (weight error ~2^-4 relative), not a parity mode */
    UITK_PREC_F16A = 6, /* round 5, pooling 'token' only: UITK_PREC_F16 (calibrated single fp16 weights) with EVERY ACTIVATION operand of the
                           column tile that holds the cls token as hi + lo fp16 terms: the LayerNorm-1 operand (into q, k, v), q, the attention
                           output, the LayerNorm-2 operand, the ReLU / GELU output.  With pooling 'token' the head reads ONE row -- no mean
                           over 24 tokens averages the fp16 activation rounding away, and split WEIGHTS do not touch it
                           (tools/err_attrib_centring.py) -- so this is the parity mode of pooling 'token' (the wrapper's 'auto' starts there) and
                           what get_option("parity_escalate_to") names for it; 1.2-1.4x the encoder time of UITK_PREC_F16.  q, k and v of EVERY
                           column tile are hi + lo as well (the cls row's scores against all keys: with full-width attention the fp16 rounding
                           of the keys carried the error on heavy-tailed checkpoints, tools/err_attrib_token.py) */
    UITK_PREC_F16AX = 7,/* round 5, pooling 'token' only, the top of its fp16 ladder: UITK_PREC_F16A's split activation operands on UITK_PREC_F16X2's
                           split WEIGHTS (its parameter image and ring schedule) -- what is left is the fp16 rounding of the other tiles'
                           LayerNorm operands; ~2x the encoder time of UITK_PREC_F16.  get_option("parity_escalate_to") of a
                           UITK_PREC_F16A handle over its budget names it */
    UITK_PREC_F32 = 8,  /* round 5: the LAST rung of the parity ladder, any family / pooling: the folded network in fp32 on the vector ALUs (the
                           plain forward of csrc/calibrate.hip: one simple kernel per operation, sums of the head in double) behind the
                           product front-end -- ~13x slower than the MFMA encoder (831 k clips/s for UiT-XS: 70x the 64-core CPU port) and
                           right for every checkpoint, e.g. those with strong attention sinks, whose fp16 activation floor no split
                           mode brings inside the budget (tools/err_attrib_sink.py).  get_option("parity_escalate_to") of a split-weight
                           handle (f16l6 / f16x2 / f16ax) that uitk_calibrate_weights measured over its budget names it */
    UITK_PREC_F16L6 = 4 /* round 3's parity mode (kept built and tested): fp16 operands as UITK_PREC_F16 plus the LOW-ORDER weight term on the cheap pipe:
                           W = f16(W) + 2^e(row) . e2m3(W - f16(W)), the second term as OCP fp6 on
                           v_mfma_scale_f32_16x16x128_f8f6f4 (one MFMA at the cost of one f16 k-step covers K = 128) against fp6
                           (LayerNorm outputs) / fp4 (ReLU outputs) copies of the activations; proj (K = 32) takes a full fp16 low
                           term.  Same accuracy as UITK_PREC_F16X2 (the correction only needs ~3 significant bits) at ~1/4 of its
                           extra MFMA time; BNeckAttention + ReLU (the uit_* family), any pooling */
} uitk_precision;

typedef enum uitk_eval_avg { UITK_AVG_MEAN = 0, UITK_AVG_MAX = 1 } uitk_eval_avg; /* models/uit.py:483-488 */

/* Mirrors the UITBase.__init__ kwargs that matter in eval mode (models/uit.py:253-293, factories :581-635). */
typedef struct uitk_config {
    int32_t depth;         /* 12 / 6 / 4                                   uit.py:585,623,604 */
    int32_t embed_dim;     /* 128                                           uit.py:584 */
    int32_t num_heads;     /* 2                                             uit.py:586 */
    int32_t mlp_hidden;    /* int(embed_dim*mlp_ratio) = 384                uit.py:237,587 */
    int32_t outputdim;     /* 537 (527 also used)                           uit.py:642, evaluate.py:33 */
    int32_t target_length; /* 102                                           uit.py:642 */
    int32_t n_mels;        /* 64   uit.py:287 */
    int32_t n_fft;         /* 512  uit.py:288 */
    int32_t hop_size;      /* 160  uit.py:289 */
    int32_t win_size;      /* 512 (any 1..n_fft: the window is zero-padded to n_fft as torch.stft does)  uit.py:290 */
    int32_t patch_size;    /* 16 (== stride)  uit.py:583 */
    int32_t center;        /* 1 (default) or 0: torch.stft centring, uit.py:293.  0: no reflect padding, 1 + (T - n_fft)/hop frames */
    int32_t eval_avg;      /* uitk_eval_avg   uit.py:294 */
    int32_t precision;     /* uitk_precision */
    float top_db;          /* 120  uit.py:307 */
    int32_t attention_full;/* 0 = BNeckAttention (uit.py:89-122: qkv 128->96, proj 32->128; the uit_* family)
                              1 = Attention (uit.py:124-178: qkv 128->384, proj 128->128, 2 heads x 64; the
                                  audio_transformer_h128_* factories uit.py:514-579)                      */
    int32_t act_gelu;      /* 0 = nn.ReLU (uit.py:591, 610, 629), 1 = nn.GELU (UITBase default, uit.py:344) */
    int32_t pooling;       /* 0 = 'mean' (uit.py:402-404; every factory);
                              1 = 'dm' (uit.py:405-412): per time patch the mean over the frequency patches -> outputlayer ->
                                  sigmoid, then the mean over time patches.  No pre-sigmoid output exists for it: uitk_forward /
                                  uitk_encode reject a non-NULL `logits`;
                              2 = 'token' (uit.py:389-392, 399-401; the UITBase default): cls_token + token_pos_embed is
                                  prepended as a 25th token and its output row alone feeds outputlayer                     */
    int32_t fe_autotune;   /* 1: the first uitk_frontend / uitk_forward of a batch shape nobody in this process has measured yet measures the
                              front-end's pass split on the caller's buffer and stream (~45 ms, event waits only; never under stream
                              capture); 0: never inside a call -- use uitk_tune_frontend, or run on the built-in split */
} uitk_config;

/* Per-block parameters, fp32 host pointers in the reference's state_dict layout (blocks.N.*). */
typedef struct uitk_block_weights {
    const float* norm1_w; const float* norm1_b; /* [128]      uit.py:223  (eps 1e-6, :337) */
    const float* qkv_w;   const float* qkv_b;   /* [96,128],[96]   uit.py:103   (attention_full: [384,128],[384]  uit.py:139) */
    const float* proj_w;  const float* proj_b;  /* [128,32],[128]  uit.py:105   (attention_full: [128,128],[128]  uit.py:141) */
    const float* norm2_w; const float* norm2_b; /* [128]      uit.py:235 */
    const float* fc1_w;   const float* fc1_b;   /* [384,128],[384] uit.py:192 */
    const float* fc2_w;   const float* fc2_b;   /* [128,384],[128] uit.py:194 */
} uitk_block_weights;

/* Whole-model parameters, fp32 host pointers, reference state_dict names in comments. */
typedef struct uitk_weights {
    const float* window;     /* front_end.0.spectrogram.window [win_size]          */
    const float* mel_fb;     /* front_end.0.mel_scale.fb [n_fft/2+1, n_mels]       */
    const float* bn_weight;  /* init_bn.1.weight [n_mels]          uit.py:310-313 */
    const float* bn_bias;    /* init_bn.1.bias                                     */
    const float* bn_mean;    /* init_bn.1.running_mean                             */
    const float* bn_var;     /* init_bn.1.running_var                              */
    const float* pe_w;       /* patch_embed.proj.weight [128,1,16,16]  uit.py:63-66 */
    const float* pe_b;       /* patch_embed.proj.bias [128]                        */
    const float* time_pos;   /* time_pos_embed [1,128,1,target_length/16]  uit.py:333 */
    const float* freq_pos;   /* freq_pos_embed [1,128,n_mels/16,1]         uit.py:335 */
    const float* norm_w;     /* norm.weight [128]  (eps 1e-6)              uit.py:357 */
    const float* norm_b;     /* norm.bias                                          */
    const float* head_ln_w;  /* outputlayer.0.weight [128] (eps 1e-5)      uit.py:359 */
    const float* head_ln_b;  /* outputlayer.0.bias                                 */
    const float* head_w;     /* outputlayer.1.weight [outputdim,128]       uit.py:360 */
    const float* head_b;     /* outputlayer.1.bias [outputdim]                     */
    const uitk_block_weights* blocks; /* [depth]                                   */
    const float* cls_token;  /* cls_token [1,1,128]          uit.py:329   (read for pooling 'token' only; may be NULL otherwise) */
    const float* token_pos;  /* token_pos_embed [1,128]      uit.py:332 */
} uitk_weights;

typedef struct uitk_model uitk_model;

/* Build a model on the CURRENT HIP device: validates the config, packs the weights into the kernels'
 * layouts (folded BN / LayerNorm affine / attention scale, MFMA fragment order) and uploads them.
 * Stands in for: factory call + load_state_dict + .to(DEVICE).eval()  (inference.py:39-49, evaluate.py:34-37). */
int uitk_create(const uitk_config* cfg, const uitk_weights* w, uitk_model** out);
int uitk_destroy(uitk_model* m);

/* Host-only: size of / fill the packed parameter image uitk_create uploads (no GPU needed; used by the
 * layout tests).  *needed receives the byte size; out may be NULL to query. */
int uitk_pack_weights(const uitk_config* cfg, const uitk_weights* w, void* out, size_t cap, size_t* needed);

/* ---- Calibrated rounding (round 4; round 5: also UITK_PREC_FP8 / UITK_PREC_FP8W, whose qkv / fc1 / fc2 weights are then rounded on the
 * MX e4m3 grid by the same algorithm).  The single-weight precisions (UITK_PREC_F16, UITK_PREC_BF16) round every weight matrix
 * against the second moments E[x x^T] of the operand it multiplies, measured on a calibration batch (GPTQ on the fp16 grid,
 * csrc/pack.cpp: round_gptq): same kernel, same speed, and the weight-rounding term of the logit error -- which dominates plain
 * round-to-nearest at trained-checkpoint magnitudes -- drops to the floor set by the fp16 activations (the accuracy of
 * UITK_PREC_F16X2 at the speed of UITK_PREC_F16).  The reference has no counterpart (it computes in fp32): this is how the
 * replacement holds 1e-3 on logits at the fp16 MFMA rate.
 *
 * uitk_calibrate_weights: run pcm[B,T] (device pointer; NULL = the library's own 32 synthetic calibration signals) through the
 * front-end and a plain fp32 forward of the folded network (csrc/calibrate.hip), measure the statistics, re-pack `w` (the SAME
 * weights the handle was created from) with calibrated rounding and replace the handle's parameter image.  A pack-time operation:
 * synchronises the device.  Clips must cover target_length frames; at most the first 1024 clips are used.
 *
 * Acceptance check (option "calib_accept_pct", default 105): before the calibrated image is installed, it and the image uitk_create
 * packed (every weight rounded to nearest) both run the PRODUCT kernels on 48 HELD-OUT clips of the library -- round 5: two
 * variants each of sixteen kinds of a different signal FAMILY from the calibration clips (impulse train, single click, DC offsets, hard clipping, +-1 LSB dither, Nyquist
 * alternation, AM/FM stacks at -40 / -12 dBFS, clipped noise, square wave, silence + burst, exponential sweep, bin-edge tone pair,
 * decaying resonances) plus sixteen ordinary signals from seeds of their own (noise at four levels and with a silent tail, chirps, tone
 * stacks, brown / band-limited noise: a checkpoint's worst input is as often one of those) -- and are compared with what the unrounded network answers there (the fp32 forward of calibrate.hip finished in
 * double on the host).  The calibrated image is kept only if its largest logit deviation is at most calib_accept_pct per cent of the
 * nearest image's (default: 5 % slack for ties); otherwise the handle keeps the nearest image and get_option("weights_calibrated")
 * stays 0.  uitk_get_calib_check reports the four numbers.
 *
 * Absolute budget (round 5; option "calib_budget_e6", default 600 = 6e-4 on logits, a quarter of that on the probabilities of pooling
 * 'dm'): the image that STAYS on the handle -- calibrated, or nearest after a refusal -- must itself be inside the budget on the held-out
 * clips.  If it is not (or the check failed), get_option("parity_escalate_to") names the precision the caller must rebuild the handle
 * with to stay a parity mode: UITK_PREC_F16L6 for BNeckAttention + ReLU, UITK_PREC_F16X2 for the other families; pooling 'token':
 * UITK_PREC_F16A, and from a UITK_PREC_F16A handle UITK_PREC_F16AX; from any of those split builds, measured over the budget by its own
 * uitk_calibrate_weights call: UITK_PREC_F32 (-1: inside the budget).  The Python wrapper's precision 'auto' (the factories' default) does exactly that and reports it; tests/native/
 * cabi_consumer.c shows the same four calls in C.  Not callable under stream capture (it allocates and synchronises). */
int uitk_calibrate_weights(uitk_model* m, const uitk_weights* w, const void* pcm, int in_dtype, int64_t B, int64_t T, void* stream);
/* Number of doubles of a statistics block for this config: patch embed [256][256], then per block LayerNorm-1 output [128][128],
 * attention output [inner][inner] (inner = 32, or 128 with attention_full), LayerNorm-2 output [128][128], activation output [384][384];
 * behind them (round 5) the per-block MEAN of the LayerNorm-1 output, [depth][128].  Configurations whose kernels centre that operand
 * at run time (attention_full, act_gelu or pooling 'token' on fp16 operands) hold the LayerNorm-1 moment of the centred operand. */
size_t uitk_calib_stats_doubles(const uitk_config* cfg);
/* The statistics the latest uitk_calibrate_weights of this handle measured (host buffer of uitk_calib_stats_doubles doubles). */
int uitk_get_calib_stats(const uitk_model* m, double* out, size_t cap_doubles);
/* What the acceptance check of the latest uitk_calibrate_weights measured on its held-out clips, five floats:
 * [0] max |d| of the nearest-rounded image, [1] max |d| of the calibrated image, [2] / [3] the same as root mean squares,
 * [4] 1 = calibrated image installed, 0 = rejected (the handle runs the nearest image); d = pre-sigmoid logits against the unrounded
 * network (probabilities for pooling 'dm', which has no logits).  Both sides start from the same 16-bit patch operands of the product
 * front-end: the numbers measure the encoder's arithmetic (weight + activation rounding), not the front-end's.
 * All -1: no check has run (option calib_accept_pct 0, or no calibration). */
int uitk_get_calib_check(const uitk_model* m, float* out5);
/* The oracle-free estimate of a handle's own parity margin, for ANY precision: the held-out clips of the acceptance check through the
 * product kernels on the parameter image the handle holds now, against the fp32 forward of the unrounded network built from `w` (the
 * weights the handle was created from).  out2[0] = max |d|, out2[1] = rms (pre-sigmoid logits; probabilities for pooling 'dm'); both -1
 * when no single-crop clip length covers target_length.  A pack-time operation like uitk_calibrate_weights: allocates, synchronises,
 * overwrites the workspace of a pending uitk_frontend; not under stream capture. */
int uitk_self_check(uitk_model* m, const uitk_weights* w, void* stream, float* out2);
/* Host-only (no GPU): the comparison uitk_self_check and the acceptance check make -- out2[0] = max |got - ref|, out2[1] = root mean
 * square over n elements.  A non-finite deviation anywhere makes BOTH results NaN (sticky), so a caller's `!(x <= budget)` and
 * `isfinite(x)` see it; an image that answers NaN on some held-out clips is never reported inside a budget. */
int uitk_deviation_stats(const float* got, const double* ref, size_t n, double* out2);
/* Host-only (no GPU): uitk_pack_weights with calibrated rounding against caller-supplied statistics (layout tests, offline packing).
 * Statistics with a non-finite or negative diagonal entry are rejected (UITK_ERR_INVALID_ARG). */
int uitk_pack_weights_calibrated(const uitk_config* cfg, const uitk_weights* w, const double* stats, void* out, size_t cap, size_t* needed);

/* Frames and crops the reference makes for a T-sample clip: 1 + T/hop frames (center=True); more than
 * target_length frames are split into ceil(frames/target_length) crops, a short tail being replaced by the
 * last target_length frames (uit.py:468-481). */
int uitk_num_frames(const uitk_model* m, int64_t T);
int uitk_num_crops(const uitk_model* m, int64_t T);

/* Pre-allocate the workspace for batches up to B clips of T samples (synchronises only when a buffer has to grow; call outside
 * timed / captured regions).  forward/frontend grow it on demand otherwise.  Measures nothing (round 4). */
int uitk_reserve(uitk_model* m, int64_t B, int64_t T);

/* Measure the front-end's pass split (option "fe_asym_pm") for batches of this shape on THIS device, on the caller's buffer
 * pcm[B,T] (contents irrelevant) and stream: ~140 short launches (~45 ms at 8192 x 1 s), event waits only.  The result is shared by
 * every handle of the process on this device; with uitk_config.fe_autotune = 1 the first uitk_frontend of an unmeasured shape
 * does the same implicitly.  Never changes a result bit. */
int uitk_tune_frontend(uitk_model* m, const void* pcm, int in_dtype, int64_t B, int64_t T, void* stream);

/* The whole path: probs[B,outputdim] = model(pcm[B,T])   (UITBase.forward, uit.py:452-493, eval branch).
 *   pcm     device pointer, [B,T] row-major, float32 or int16 (in_dtype)
 *   probs   device pointer, [B,outputdim] float32 sigmoid outputs (mean/max over crops)
 *   logits  optional device pointer, [B,ncrops,outputdim] float32 pre-sigmoid outputs of outputlayer[1]
 *   stream  hipStream_t (NULL = default stream)
 * The top_db floor is amax-over-this-call's-batch minus top_db, exactly as the reference computes it for the
 * tensor of one forward call.  = uitk_frontend + uitk_encode. */
int uitk_forward(uitk_model* m, const void* pcm, int in_dtype, int64_t B, int64_t T,
                 float* probs, float* logits, void* stream);

/* Stage 1: log-mel dB of every frame into the model's workspace + the batch amax.
 * Stands in for self.front_end(x) minus the clamp (uit.py:298-308, 455). */
int uitk_frontend(uitk_model* m, const void* pcm, int in_dtype, int64_t B, int64_t T, void* stream);

/* Device pointer to the batch amax slots (UITK_AMAX_SLOTS floats; the floor uses their max).  A multi-GPU
 * caller that wants the reference's batch-global floor across shards all-reduces (MAX) these in place
 * between uitk_frontend and uitk_encode.  The pointer is the slot set of the LATEST uitk_frontend call (two sets
 * alternate, so that no per-call memset is needed): query it after each uitk_frontend. */
#define UITK_AMAX_SLOTS 64
float* uitk_amax_ptr(uitk_model* m);

/* Stage 2: top_db clamp + init_bn + patch-embed + blocks + head on the mel of the last uitk_frontend call
 * (same B, T).  Stands in for uit.py:460-492. */
int uitk_encode(uitk_model* m, int64_t B, int64_t T, float* probs, float* logits, void* stream);

/* Diagnostic / stage-parity entry: run the front-end on pcm and return what self.front_end(x) returns in the
 * reference: dB mel, batch-global floor applied, layout [B][n_mels][frames].  (The production path keeps the mel
 * in the encoder's patch-major operand layout; this call re-runs the front-end kernel in a frame-major mode and
 * overwrites the amax slots.) */
int uitk_mel_db(uitk_model* m, const void* pcm, int in_dtype, int64_t B, int64_t T, float* out, void* stream);

/* Optional: device buffer [B*ncrops][depth+1][24][128] float32 receiving the residual stream after
 * patch-embed(+pos) and after every block (token = f*6+t as in uit.py:388).  NULL disables (default). */
int uitk_set_debug_tokens(uitk_model* m, float* dev_buf);

/* Diagnostics (lab builds): device buffers receiving per-phase shader-cycle sums per wave -- encoder [waves][20] uint64, front-end
 * [waves][8] (tools/phase_timing.py).  The instrumented kernels are compiled only into -DUITK_LAB builds of the library
 * (tools/mklib.sh); the product library returns UITK_ERR_UNSUPPORTED for a non-NULL buffer.  NULL (default) = normal kernels. */
int uitk_set_timing_buffers(uitk_model* m, unsigned long long* enc_buf, unsigned long long* fe_buf);

/* Explicit tuning / test options of a model handle (the library reads no environment variables).  None of them changes a result
 * bit (tests/test_gpu_parity.py::test_pass_distribution_does_not_change_results) -- "calib_accept_pct" decides whether a calibrated
 * image may be refused, not what an image computes.  Unknown name -> UITK_ERR_UNSUPPORTED.
 *   "fe_asym_pm"        300..700: per-mille share of the front-end passes walked by the first-dispatched workgroups of every CU;
 *                       -1 (default) = the share measured for the batch shape on this device (uitk_tune_frontend / fe_autotune)
 *   "fe_wg_per_cu"      persistent front-end workgroups per CU (0 = default 2)
 *   "enc_waves"         0 (default: by batch size -- 4-wave workgroups while the batch fits one of them per CU, i.e. up to 8 clip rows per
 *                       CU: 1.3-1.4x lower latency for small batches; 8-wave workgroups above, and the rows left over after the full
 *                       8-wave rounds go out as a second, 4-wave launch when they fit one), 8 (one 8-wave encoder workgroup per CU)
 *                       or 4 (two 4-wave workgroups per CU); the uit_* family in f16 / bf16 / fp8 -- other variants always run 8
 *   "split_pe"          1 / 0: patch embed on split (hi + lo) weights; -1 = the precision's default (on for f16x2)
 *   "enc_generic_clamp" 1 = use the generic top_db clamp path even when every init_bn scale is >= 0 (test of that path)
 *   "fe_tail_skip"      1 (default) / 0: a clip's leftover frames (1 s clips: frames 96..100, which only count for the batch amax)
 *                       skip their FFT when a Parseval bound shows they cannot raise it; 0 transforms every frame (test of the skip)
 *   "calib_accept_pct"  0..1000 (default 105): uitk_calibrate_weights installs its image only if its largest deviation on the library's
 *                       held-out clips is at most this many per cent of the nearest-rounded image's (see there); 0 = no check, install
 *                       unconditionally; 1 = refuse in practice (test of the fallback)
 *   "calib_budget_e6"   0..1000000 (default 600): absolute budget, in 1e-6 units, for the largest deviation of the image that stays
 *                       installed after uitk_calibrate_weights on the held-out clips (see there); 0 = no budget
 *   "calib_short_check" 0 (default) / 1 (round 6): the held-out clips of the acceptance check, the budget and uitk_self_check are ALSO
 *                       scored at half length (three of six time patches: 12 of 24 tokens, masked attention) and every reported
 *                       maximum is over both lengths -- for deployments on short clips, where fewer tokens average the rounding
 *                       (the Python wrapper's prepare() switches it on when the prepared clip length has fewer than six time patches) */
int uitk_set_option(uitk_model* m, const char* name, int value);
/* Current value of an option ("fe_asym_pm": the explicit setting, else the share the latest front-end launch used, -1 = the
 * built-in 590 because nothing was measured for its shape); also "weights_calibrated" (read-only: 1 when the
 * handle runs an image uitk_calibrate_weights built AND its acceptance check kept) and "parity_escalate_to" (read-only: -1, or the
 * uitk_precision to rebuild the handle with because the installed image is over the budget -- see uitk_calibrate_weights). */
int uitk_get_option(const uitk_model* m, const char* name, int* value);

/* ---- Step after the path: post-processing of the probability rows on the device, so that only
 * k winners (or class indices) cross PCIe instead of [B, outputdim].  Stateless: no model handle.  C <= 1024.
 *
 * uitk_topk: values[B,k], indices[B,k] = probs[b].topk(k), descending, ties by lower index.
 * Stands in for output.topk(args.topk) (inference.py:57) and pred.topk (evaluate.py:263). */
int uitk_topk(const float* probs, int64_t B, int32_t C, int32_t k, float* values, int32_t* indices, void* stream);

/* uitk_kws_transform: Accuracy@threshold output transform of the keyword-spotting evaluation, in place on
 * y_pred[B,C]: keep only the maximal score among the first num_fillers (AudioSet) classes, and clear it too when any
 * keyword class (>= num_fillers) reaches the threshold; y_index[B] = argmax of the target row y[B,C] with filler targets
 * mapped to the predicted filler (y / y_index may both be NULL).
 * Stands in for the per-sample Python loop _output_transform_xiaoai (evaluate.py:212-229). */
int uitk_kws_transform(float* y_pred, const float* y, int64_t B, int32_t C, int32_t num_fillers, float threshold,
                       int64_t* y_index, void* stream);

/* ---- The GENERAL engine (round 6; csrc/general.hip): UITBase.forward (eval) for every geometry the reference's constructor admits
 * (models/uit.py:253-363: n_mels / n_fft / win_size / hop_size / center, patch_size (== stride), target_length, embed_dim, depth,
 * num_heads, mlp_ratio, both attention classes, ReLU / GELU, the three poolings) that the MFMA kernels above are NOT specialised to --
 * uitk_create refuses those (UITK_ERR_UNSUPPORTED).  Plain fp32 arithmetic on the device in the reference's order of operations (the
 * dense layers on the matrix pipe's exact-fp32 mode); cfg->precision must be UITK_PREC_F32; nothing to calibrate or tune.  The weights
 * use uitk_weights with the shapes the config implies: pe_w [embed, 1, P, P], time_pos [1, embed, 1, target_length / P],
 * freq_pos [1, embed, n_mels / P, 1], qkv_w [3 inner, embed] (inner = embed, or embed / 4 for BNeckAttention), proj_w [embed, inner],
 * fc1_w [mlp_hidden, embed], fc2_w [embed, mlp_hidden], head_w [outputdim, embed], mel_fb [n_fft / 2 + 1, n_mels], window [win_size].
 * Limits: n_fft even, 16..4096 (a power of two runs an FFT; any other value, up to 2048, the DFT sum in double); patch_size a multiple of 4;
 * embed_dim, mlp_hidden and the attention's inner width multiples of 16.
 * Same calling rules as the uitk_model functions: one handle per device, enqueue-only on `stream` once the workspace is reserved,
 * uitk_general_frontend / uitk_general_encode are the two stages of uitk_general_forward (the batch amax -- uitk_general_amax_ptr,
 * UITK_AMAX_SLOTS floats -- may be MAX-reduced across ranks between them), errors through uitk_last_error.
 * Stand in for: UITBase(**kwargs) + load_state_dict + .to(DEVICE).eval() and model(x) (models/uit.py:253-363, 452-493). */
typedef struct uitk_general uitk_general;
int uitk_general_create(const uitk_config* cfg, const uitk_weights* w, uitk_general** out);
int uitk_general_destroy(uitk_general* m);
int uitk_general_num_frames(const uitk_general* m, int64_t T);
int uitk_general_num_crops(const uitk_general* m, int64_t T);
int uitk_general_reserve(uitk_general* m, int64_t B, int64_t T);
int uitk_general_frontend(uitk_general* m, const void* pcm, int in_dtype, int64_t B, int64_t T, void* stream);
float* uitk_general_amax_ptr(uitk_general* m);
/* probs [B, outputdim]; logits [B, ncrops, outputdim] or NULL (must be NULL for pooling 'dm') */
int uitk_general_encode(uitk_general* m, int64_t B, int64_t T, float* probs, float* logits, void* stream);
int uitk_general_forward(uitk_general* m, const void* pcm, int in_dtype, int64_t B, int64_t T, float* probs, float* logits, void* stream);
/* out [B, n_mels, frames]: what self.front_end(x) returns (models/uit.py:455), the batch-global top_db floor applied */
int uitk_general_mel_db(uitk_general* m, const void* pcm, int in_dtype, int64_t B, int64_t T, float* out, void* stream);

/* Last error message of the calling thread ("" if none). */
const char* uitk_last_error(void);
const char* uitk_version(void);

#ifdef __cplusplus
}
#endif
#endif /* UITK_H */
