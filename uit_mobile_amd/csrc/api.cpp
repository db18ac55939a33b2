// api.cpp -- the C ABI of libuitk.so (include/uitk.h): model lifetime, workspace, stage launches.
// Host code only; the kernels live in frontend.hip / encoder.hip.  There is no CPU compute path here.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/uitk.h"
#include "pack.h"
#include "uitk_dev.h"

using namespace uitk;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return fail(UITK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct uitk_model {
    uitk_config cfg;
    ImageHeader hdr;
    char* image = nullptr;        // device copy of the packed parameter image
    float* mel = nullptr;         // patch-major 16-bit operands [B*ncrops][24][16][16]
    float* mel_frames = nullptr;  // [B][nF][64], only for uitk_mel_db
    size_t mel_frames_cap = 0;
    float* amax = nullptr;        // [2][AMAX_SLOTS]: the front-end of call n fills set n&1 and re-arms the other one
    int amax_cur = 0;             // set of the latest front-end launch
    float* rowprobs = nullptr;    // [B*ncrops][C] (multi-crop only)
    size_t mel_cap = 0, rowprobs_cap = 0;
    float* dbg = nullptr;
    unsigned long long* timing = nullptr;
    unsigned long long* timing_fe = nullptr;
    int n_cu = 256;
    int enc_waves = 8;            // waves per encoder workgroup (8 = one WG per CU; 4 = two independent WGs per CU)
    int enc_sched = 0, enc_stagger = 0, enc_stagger_mode = 1, enc_ablate = 0;   // tuning knobs (UITK_ENC_SCHED, UITK_ENC_STAGGER)
    int split_pe = 0;             // UITK_SPLIT_PE=1: patch embed on split weights (W = hi + lo); measured no better than the
                                  // structured rounding of pack.cpp, +1.6 % encoder time: off by default
    int bn_nonneg = 0;            // every init_bn scale >= 0 (decided once from the packed table): the encoder's clamp fast path
    int fe_asym_pm = -1;          // UITK_FE_ASYM=<per-mille>: share of the first-dispatched workgroups (-1 = launch_frontend decides: 590 / 500)
    int fe_static = 0;            // UITK_FE_STATIC=1: purely static pass split in the front-end (A/B of the dynamic remainder)
    int fe_wg_per_cu = 0;         // persistent front-end workgroups per CU (0 = the kernel's occupancy target; UITK_FE_WG_PER_CU)
    int device = 0;
    int64_t fe_B = 0, fe_T = 0;   // shape of the latest uitk_frontend call (uitk_encode must match it)
};

extern "C" {

const char* uitk_last_error(void) { return g_err.c_str(); }
const char* uitk_version(void) { return "uitk 0.1 (gfx950)"; }

int uitk_pack_weights(const uitk_config* cfg, const uitk_weights* w, void* out, size_t cap, size_t* needed) {
    if (!cfg || !w || !needed) return fail(UITK_ERR_INVALID_ARG, "null argument");
    const char* why = "";
    if (int rc = validate_config(cfg, &why)) return fail(rc, why);
    std::vector<char> img;
    ImageHeader h;
    if (int rc = pack_image(cfg, w, &img, &h)) return fail(rc, "pack failed");
    *needed = img.size();
    if (out) {
        if (cap < img.size()) return fail(UITK_ERR_INVALID_ARG, "output buffer too small");
        std::memcpy(out, img.data(), img.size());
    }
    return UITK_OK;
}

int uitk_create(const uitk_config* cfg, const uitk_weights* w, uitk_model** out) {
    if (!cfg || !w || !out) return fail(UITK_ERR_INVALID_ARG, "null argument");
    const char* why = "";
    if (int rc = validate_config(cfg, &why)) return fail(rc, why);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(UITK_ERR_NO_DEVICE, "no HIP device: libuitk has no CPU path");
    uitk_model* m = new uitk_model();
    m->cfg = *cfg;
    std::vector<char> img;
    if (int rc = pack_image(cfg, w, &img, &m->hdr)) { delete m; return fail(rc, "pack failed"); }
    hipDeviceProp_t prop;
    if (hipGetDevice(&m->device) != hipSuccess || hipGetDeviceProperties(&prop, m->device) != hipSuccess) {
        delete m; return fail(UITK_ERR_NO_DEVICE, "cannot query HIP device");
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::string arch = prop.gcnArchName; delete m;
        return fail(UITK_ERR_NO_DEVICE, "libuitk is built for gfx950 only, found " + arch);
    }
    m->n_cu = prop.multiProcessorCount;
    {
        const float* sc = (const float*)(img.data() + m->hdr.off_bn_sc);
        m->bn_nonneg = 1;
        for (int i = 0; i < NMEL; ++i) if (!(sc[i] >= 0.f)) m->bn_nonneg = 0;
        if (const char* e = std::getenv("UITK_ENC_GENERIC_CLAMP")) if (std::atoi(e)) m->bn_nonneg = 0;     // A/B and tests of the generic path
    }
    if (const char* e = std::getenv("UITK_ENC_WAVES")) m->enc_waves = std::atoi(e) == 4 ? 4 : 8;
    if (const char* e = std::getenv("UITK_ENC_SCHED")) m->enc_sched = std::atoi(e) != 0;
    if (const char* e = std::getenv("UITK_ENC_STAGGER")) m->enc_stagger = std::atoi(e);
    if (const char* e = std::getenv("UITK_ENC_STAGGER_MODE")) m->enc_stagger_mode = std::atoi(e);
    if (const char* e = std::getenv("UITK_ENC_ABLATE")) m->enc_ablate = std::atoi(e);
    if (const char* e = std::getenv("UITK_FE_WG_PER_CU")) m->fe_wg_per_cu = std::atoi(e);
    if (const char* e = std::getenv("UITK_FE_STATIC")) m->fe_static = std::atoi(e) != 0;
    if (const char* e = std::getenv("UITK_FE_ASYM")) m->fe_asym_pm = std::atoi(e);
    if (const char* e = std::getenv("UITK_SPLIT_PE")) m->split_pe = std::atoi(e) != 0;
    if (hipMalloc((void**)&m->image, img.size()) != hipSuccess || hipMalloc((void**)&m->amax, 2 * AMAX_SLOTS * 4 + 2 * 16 * 4) != hipSuccess) {      // + two sets of (up to 16) pass counters of the front-end
        uitk_destroy(m); return fail(UITK_ERR_HIP, "hipMalloc failed for the parameter image");
    }
    float armed[2 * AMAX_SLOTS];
    for (float& v : armed) v = -INFINITY;                                       // both amax slot sets start armed (-inf)
    if (hipMemcpy(m->image, img.data(), img.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->amax, armed, sizeof armed, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(m->amax + 2 * AMAX_SLOTS, 0, 2 * 16 * 4) != hipSuccess) {
        uitk_destroy(m); return fail(UITK_ERR_HIP, "parameter upload failed");
    }
    *out = m;
    return UITK_OK;
}

int uitk_destroy(uitk_model* m) {
    if (!m) return UITK_OK;
    if (m->image) (void)hipFree(m->image);
    if (m->mel) (void)hipFree(m->mel);
    if (m->mel_frames) (void)hipFree(m->mel_frames);
    if (m->amax) (void)hipFree(m->amax);
    if (m->rowprobs) (void)hipFree(m->rowprobs);
    delete m;
    return UITK_OK;
}

int uitk_num_frames(const uitk_model* m, int64_t T) { return m ? (int)(1 + T / m->cfg.hop_size) : -1; }

int uitk_num_crops(const uitk_model* m, int64_t T) {
    if (!m) return -1;
    const int nF = uitk_num_frames(m, T), tl = m->cfg.target_length;
    return nF <= tl ? 1 : (nF + tl - 1) / tl;
}

static int check_shape(const uitk_model* m, int64_t B, int64_t T) {
    if (!m) return fail(UITK_ERR_INVALID_ARG, "null model");
    if (B < 1 || T < 1) return fail(UITK_ERR_INVALID_ARG, "B and T must be positive");
    if (T <= m->cfg.n_fft / 2) return fail(UITK_ERR_TOO_SHORT, "clip must be longer than n_fft/2 samples (reflect padding)");
    if (uitk_num_frames(m, T) < m->cfg.patch_size) return fail(UITK_ERR_TOO_SHORT, "clip shorter than one 16-frame patch");
    if (T >= (int64_t)1 << 30) return fail(UITK_ERR_INVALID_ARG, "T too large");
    if (B * (int64_t)uitk_num_frames(m, T) >= (int64_t)1 << 31) return fail(UITK_ERR_INVALID_ARG, "B * frames must be below 2^31 per call");
    return UITK_OK;
}

int uitk_reserve(uitk_model* m, int64_t B, int64_t T) {
    if (int rc = check_shape(m, B, T)) return rc;
    const int nc = uitk_num_crops(m, T);
    const size_t mel_bytes = (size_t)B * nc * NTOK * 256 * 2;
    const size_t rp_bytes = nc > 1 ? (size_t)B * nc * m->cfg.outputdim * 4 : 0;
    if (mel_bytes > m->mel_cap) {
        HIP_TRY(hipDeviceSynchronize());
        if (m->mel) (void)hipFree(m->mel);
        m->mel = nullptr; m->mel_cap = 0;
        HIP_TRY(hipMalloc((void**)&m->mel, mel_bytes));
        m->mel_cap = mel_bytes;
    }
    if (rp_bytes > m->rowprobs_cap) {
        HIP_TRY(hipDeviceSynchronize());
        if (m->rowprobs) (void)hipFree(m->rowprobs);
        m->rowprobs = nullptr; m->rowprobs_cap = 0;
        HIP_TRY(hipMalloc((void**)&m->rowprobs, rp_bytes));
        m->rowprobs_cap = rp_bytes;
    }
    return UITK_OK;
}

float* uitk_amax_ptr(uitk_model* m) { return m ? m->amax + m->amax_cur * AMAX_SLOTS : nullptr; }

int uitk_set_debug_tokens(uitk_model* m, float* dev_buf) {
    if (!m) return fail(UITK_ERR_INVALID_ARG, "null model");
    if (dev_buf && m->cfg.pooling == 2) return fail(UITK_ERR_UNSUPPORTED, "the token dump is laid out for 24 tokens per clip (not for pooling 'token')");
    m->dbg = dev_buf;
    return UITK_OK;
}

int uitk_set_timing_buffers(uitk_model* m, unsigned long long* enc_buf, unsigned long long* fe_buf) {
    if (!m) return fail(UITK_ERR_INVALID_ARG, "null model");
    if ((enc_buf || fe_buf) && (m->cfg.precision != UITK_PREC_F16 || m->cfg.attention_full || m->cfg.act_gelu || m->cfg.pooling))
        return fail(UITK_ERR_UNSUPPORTED, "the instrumented kernel builds exist for precision f16, BNeckAttention + ReLU only");
    m->timing = enc_buf; m->timing_fe = fe_buf;
    return UITK_OK;
}

static int crop_nt(const uitk_model* m, int64_t T) {
    const int nF = uitk_num_frames(m, T), nc = uitk_num_crops(m, T);
    int nt = (nc > 1 ? m->cfg.target_length : nF) / 16;
    return nt > GT ? GT : nt;
}

static int run_frontend(uitk_model* m, const void* pcm, int in_dtype, int64_t B, int64_t T, float* mel_out, int patch,
                        hipStream_t s) {
    const ImageHeader& h = m->hdr;
    FrontendArgs a{};
    a.pcm = pcm; a.B = B; a.T = T; a.nF = uitk_num_frames(m, T); a.hop = m->cfg.hop_size;
    a.window = (const float*)(m->image + h.off_window);
    a.tw256 = (const float2*)(m->image + h.off_tw256);
    a.tw512 = (const float2*)(m->image + h.off_tw512);
    a.mel_b = (const float*)(m->image + h.off_mel_b);
    a.mel_steps = h.mel_steps;
    // no memset per call: this launch fills one set of slots and re-arms (0xFFFFFFFF) the other for the next launch
    m->amax_cur ^= 1;
    a.mel = mel_out; a.amax = m->amax + m->amax_cur * AMAX_SLOTS; a.amax_rearm = m->amax + (m->amax_cur ^ 1) * AMAX_SLOTS;
    a.work_ctr = (int*)(m->amax + 2 * AMAX_SLOTS) + 16 * m->amax_cur; a.work_ctr_rearm = (int*)(m->amax + 2 * AMAX_SLOTS) + 16 * (m->amax_cur ^ 1);
    if (m->fe_static) a.work_ctr = nullptr;
    a.asym_pm = m->fe_asym_pm;
    a.bn_sc = (const float*)(m->image + h.off_bn_sc); a.bn_sh = (const float*)(m->image + h.off_bn_sh);
    a.op_bf16 = m->cfg.precision == UITK_PREC_BF16;
    a.timing = m->timing_fe;
    a.ncrops = uitk_num_crops(m, T); a.tl = m->cfg.target_length; a.nt = crop_nt(m, T);
    HIP_TRY(launch_frontend(a, in_dtype == UITK_IN_I16, patch, m->n_cu, m->fe_wg_per_cu, s));
    return UITK_OK;
}

int uitk_frontend(uitk_model* m, const void* pcm, int in_dtype, int64_t B, int64_t T, void* stream) {
    if (int rc = check_shape(m, B, T)) return rc;
    if (!pcm) return fail(UITK_ERR_INVALID_ARG, "null pcm");
    if (in_dtype != UITK_IN_F32 && in_dtype != UITK_IN_I16) return fail(UITK_ERR_INVALID_ARG, "unknown in_dtype");
    if (int rc = uitk_reserve(m, B, T)) return rc;
    m->fe_B = 0; m->fe_T = 0;
    if (int rc = run_frontend(m, pcm, in_dtype, B, T, m->mel, 1, (hipStream_t)stream)) return rc;
    m->fe_B = B; m->fe_T = T;
    return UITK_OK;
}

int uitk_encode(uitk_model* m, int64_t B, int64_t T, float* probs, float* logits, void* stream) {
    if (int rc = check_shape(m, B, T)) return rc;
    if (!probs) return fail(UITK_ERR_INVALID_ARG, "null probs");
    if (logits && m->cfg.pooling == 1) return fail(UITK_ERR_UNSUPPORTED, "pooling 'dm' has no pre-sigmoid output (models/uit.py:405-412): pass logits = NULL");
    const int nF = uitk_num_frames(m, T), nc = uitk_num_crops(m, T), tl = m->cfg.target_length;
    if ((size_t)B * nc * NTOK * 256 * 2 > m->mel_cap || B != m->fe_B || T != m->fe_T)
        return fail(UITK_ERR_WORKSPACE, "uitk_encode without a matching uitk_frontend (same B and T) before it");
    (void)nF; (void)tl;
    if (B * nc >= (int64_t)1 << 31) return fail(UITK_ERR_INVALID_ARG, "batch too large");
    hipStream_t s = (hipStream_t)stream;
    const ImageHeader& h = m->hdr;
    EncoderArgs a{};
    a.mel = (const unsigned short*)m->mel; a.amax = m->amax + m->amax_cur * AMAX_SLOTS; a.top_db = m->cfg.top_db;
    a.rows = (int)(B * nc);
    a.nt = crop_nt(m, T);
    a.depth = m->cfg.depth; a.C = m->cfg.outputdim;
    a.wstream = m->image + h.off_wstream;
    a.bias = (const float*)(m->image + h.off_bias);
    a.bn_sc = (const float*)(m->image + h.off_bn_sc);
    a.bn_sh = (const float*)(m->image + h.off_bn_sh);
    a.pe_bias = (const float*)(m->image + h.off_pe_bias);
    a.fin = (const float*)(m->image + h.off_fin);
    a.head_w = m->image + h.off_head_w;
    a.head_b = (const float*)(m->image + h.off_head_b);
    a.probs = nc > 1 ? m->rowprobs : probs;
    a.logits = logits; a.dbg = m->dbg;
    a.timing = m->timing; a.ablate = m->enc_ablate;
    a.bn_nonneg = m->bn_nonneg;
    a.split_pe = m->split_pe || m->cfg.precision == UITK_PREC_F16X2 || m->cfg.precision == UITK_PREC_F16L6;     // f16x2 / f16l6: the patch embed runs on its split weights too
    a.attention_full = m->cfg.attention_full; a.act_gelu = m->cfg.act_gelu; a.pooling = m->cfg.pooling;
    a.stagger = m->enc_stagger; a.stagger_group = m->n_cu; a.stagger_mode = m->enc_stagger_mode;
    HIP_TRY(launch_encoder(a, m->cfg.precision, m->enc_waves, m->enc_sched, s));
    if (nc > 1) HIP_TRY(launch_crop_reduce(m->rowprobs, probs, B, nc, a.C, m->cfg.eval_avg, s));
    return UITK_OK;
}

int uitk_forward(uitk_model* m, const void* pcm, int in_dtype, int64_t B, int64_t T, float* probs, float* logits,
                 void* stream) {
    if (int rc = uitk_frontend(m, pcm, in_dtype, B, T, stream)) return rc;
    return uitk_encode(m, B, T, probs, logits, stream);
}

int uitk_mel_db(uitk_model* m, const void* pcm, int in_dtype, int64_t B, int64_t T, float* out, void* stream) {
    if (int rc = check_shape(m, B, T)) return rc;
    if (!pcm || !out) return fail(UITK_ERR_INVALID_ARG, "null pointer");
    if (in_dtype != UITK_IN_F32 && in_dtype != UITK_IN_I16) return fail(UITK_ERR_INVALID_ARG, "unknown in_dtype");
    const int nF = uitk_num_frames(m, T);
    const size_t need = (size_t)B * nF * NMEL * 4;
    if (need > m->mel_frames_cap) {
        HIP_TRY(hipDeviceSynchronize());
        if (m->mel_frames) (void)hipFree(m->mel_frames);
        m->mel_frames = nullptr; m->mel_frames_cap = 0;
        HIP_TRY(hipMalloc((void**)&m->mel_frames, need));
        m->mel_frames_cap = need;
    }
    m->fe_B = 0; m->fe_T = 0;                                  // the amax slots now belong to this call, not to the operands
    if (int rc = run_frontend(m, pcm, in_dtype, B, T, m->mel_frames, 0, (hipStream_t)stream)) return rc;
    HIP_TRY(launch_mel_export(m->mel_frames, m->amax + m->amax_cur * AMAX_SLOTS, m->cfg.top_db, B, nF, out, (hipStream_t)stream));
    return UITK_OK;
}

int uitk_topk(const float* probs, int64_t B, int32_t C, int32_t k, float* values, int32_t* indices, void* stream) {
    if (!probs || !values || !indices) return fail(UITK_ERR_INVALID_ARG, "null pointer");
    if (B < 0 || C < 1 || C > 1024) return fail(UITK_ERR_INVALID_ARG, "topk: need 1 <= C <= 1024");
    if (k < 1 || k > C) return fail(UITK_ERR_INVALID_ARG, "topk: selected index k out of range");     // torch.topk's RuntimeError
    if (B == 0) return UITK_OK;
    HIP_TRY(launch_topk(probs, B, C, k, values, indices, (hipStream_t)stream));
    return UITK_OK;
}

int uitk_kws_transform(float* y_pred, const float* y, int64_t B, int32_t C, int32_t num_fillers, float threshold,
                       int64_t* y_index, void* stream) {
    if (!y_pred || ((y == nullptr) != (y_index == nullptr))) return fail(UITK_ERR_INVALID_ARG, "null pointer");
    if (B < 0 || C < 1 || C > 1024 || num_fillers < 1) return fail(UITK_ERR_INVALID_ARG, "kws_transform: need 1 <= C <= 1024, num_fillers >= 1");
    if (B == 0) return UITK_OK;
    HIP_TRY(launch_kws_transform(y_pred, y, B, C, num_fillers < C ? num_fillers : C, threshold, y_index, (hipStream_t)stream));
    return UITK_OK;
}

}  // extern "C"
