// api.cpp -- the C ABI of libuitk.so (include/uitk.h): model lifetime, workspace, stage launches.
// Host code only; the kernels live in frontend.hip / encoder.hip.  There is no CPU compute path here.
#include <hip/hip_runtime.h>

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/uitk.h"
#include "pack.h"
#include "uitk_dev.h"
#include "naive_forward.h"

using namespace uitk;

static thread_local std::string g_err;
// precisions whose weights are ONE term each (calibrated rounding applies); fp8w runs the f16 kernels on weights of the e4m3 grid
static bool single_weight_precision(int p) { return p == UITK_PREC_F16 || p == UITK_PREC_BF16 || p == UITK_PREC_FP8 || p == UITK_PREC_FP8W || p == UITK_PREC_F16A; }
// uitk_calibrate_weights also serves the split-weight precisions: their weights are not re-rounded, but the statistics install the
// operand centring of the attention (pack.cpp) -- what an escalated 'auto' handle keeps of its calibration
static bool calibratable_precision(int p) { return single_weight_precision(p) || p == UITK_PREC_F16X2 || p == UITK_PREC_F16L6 || p == UITK_PREC_F16AX; }
// what launch_encoder is told: fp8w runs the f16 kernels; f16a = 5, the f16 kernel with split activation operands in the cls tile
// precision 'f32' (round 5, the last rung of the parity ladder) runs the plain fp32 forward of calibrate.hip: the packer still builds an
// image for it -- the front-end's tables, init_bn -- in the layout of UITK_PREC_F16
static uitk_config pack_config(const uitk_config* cfg) { uitk_config c = *cfg; if (c.precision == UITK_PREC_F32) c.precision = UITK_PREC_F16; return c; }
static int kernel_precision(int p) { return p == UITK_PREC_FP8W ? UITK_PREC_F16 : p == UITK_PREC_F16A ? 5 : p == UITK_PREC_F16AX ? 6 : p; }
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
namespace uitk { int set_error(int code, const std::string& msg) { g_err = msg; return code; } }   // general.hip reports through the same uitk_last_error
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return fail(UITK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct F32Path;
struct uitk_model {
    uitk_config cfg;
    F32Path* f32 = nullptr;       // precision 'f32': the fp32 network on the device + its work buffers (below)
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
    int enc_waves = 0;            // option enc_waves: waves per encoder workgroup (8 = one WG per CU; 4 = two independent WGs per CU;
                                  // 0 = by batch size: 4 while the batch fits one 4-wave workgroup per CU, see uitk_encode)
    int split_pe_env = -1;        // option split_pe: patch embed on split weights (W = hi + lo); -1 = the precision's default (on for f16x2;
                                  // off for f16: measured no better than the structured rounding of pack.cpp at +1.6 % encoder time)
    int force_generic_clamp = 0;  // option enc_generic_clamp
    int fe_tail_skip = 1;         // option fe_tail_skip: leftover frames that provably cannot raise the batch amax skip their FFT (0: always transform)
    int bn_nonneg = 0;            // every init_bn scale >= 0 (decided once from the packed table): the encoder's clamp fast path
    int fe_asym_pm = -1;          // option fe_asym_pm: explicit share of the first-dispatched front-end workgroups; -1 = the measured one
    int fe_asym_cur = -1;         // the share the latest front-end launch used (-1 = none measured for its shape: the kernel's 590)
    int weights_calibrated = 0;   // the uploaded image carries calibrated rounding (uitk_calibrate_weights)
    std::vector<double> calib_stats;   // statistics of the latest uitk_calibrate_weights (uitk_get_calib_stats)
    std::vector<char> img_nearest;     // host copy of the image uitk_create packed (data-free rounding): what the held-out check of
                                       // uitk_calibrate_weights compares the calibrated image with, and what it falls back to
    int calib_accept_pct = 105;        // option calib_accept_pct: uitk_calibrate_weights keeps its image only if, on the held-out clips, its
                                       // largest deviation is at most this many per cent of the nearest-rounded image's; 0 = no check
    float calib_report[5] = {-1.f, -1.f, -1.f, -1.f, -1.f};   // uitk_get_calib_check
    int calib_short_check = 0;         // option calib_short_check (round 6; default off = round 5's check): the held-out clips are ALSO scored at half length (three time patches: 12 of 24
                                       // tokens, masked attention) -- with fewer tokens to average over, a checkpoint can be inside the budget on full clips and at
                                       // the gate on short ones; the report's maximum is then over both lengths.  For deployments on short clips: the wrapper's
                                       // prepare() switches it on when the prepared clip length has fewer than six time patches.  Off by default: a model that
                                       // only ever sees 1 s clips must not be sent up the ladder (to a 12x slower rung) for what it would do on 0.5 s clips
    int calib_budget_e6 = 600;         // option calib_budget_e6: ABSOLUTE budget (1e-6 units) for the largest deviation of the installed image on the
                                       // held-out clips (logits; a quarter of it for the probabilities of pooling 'dm'); over it -- or calibration
                                       // refused -- the handle reports the precision to rebuild with ("parity_escalate_to"); 0 = no budget
    int parity_escalate_to = -1;       // read-only: -1 = the handle is inside its budget (or nothing was measured), else the uitk_precision the
                                       // caller should rebuild with (f16l6 for BNeckAttention + ReLU, f16x2 otherwise): round 5, VERDICT r4 item 1
    int fe_wg_per_cu = 0;         // option fe_wg_per_cu: persistent front-end workgroups per CU (0 = the kernel's occupancy target)
    int device = 0;
    float fb_max = 0.f;           // largest mel filterbank entry (front-end: Parseval bound of the leftover-frame skip)
    int64_t fe_B = 0, fe_T = 0;   // shape of the latest uitk_frontend call (uitk_encode must match it)
};

// precision 'f32' (defined behind the naive network's helpers, below)
static int f32_create(uitk_model* m, const uitk_weights* w);
static void f32_destroy(uitk_model* m);
static int f32_reserve(uitk_model* m, int64_t rows);
static int f32_encode(uitk_model* m, int rows, int nt, int nF, int ncrops, float* rowprobs, float* rowlogits, hipStream_t s);

extern "C" {

const char* uitk_last_error(void) { return g_err.c_str(); }
const char* uitk_version(void) { return "uitk 0.1 (gfx950)"; }

int uitk_pack_weights(const uitk_config* cfg, const uitk_weights* w, void* out, size_t cap, size_t* needed) {
    if (!cfg || !w || !needed) return fail(UITK_ERR_INVALID_ARG, "null argument");
    const char* why = "";
    const uitk_config pc = pack_config(cfg);
    if (int rc = validate_config(&pc, &why)) return fail(rc, why);
    std::vector<char> img;
    ImageHeader h;
    if (int rc = pack_image(&pc, w, nullptr, &img, &h)) return fail(rc, "pack failed");
    *needed = img.size();
    if (out) {
        if (cap < img.size()) return fail(UITK_ERR_INVALID_ARG, "output buffer too small");
        std::memcpy(out, img.data(), img.size());
    }
    return UITK_OK;
}

size_t uitk_calib_stats_doubles(const uitk_config* cfg) {
    const char* why = "";
    if (!cfg || validate_config(cfg, &why) != UITK_OK) return 0;
    return calib_stats_doubles(cfg);
}

int uitk_pack_weights_calibrated(const uitk_config* cfg, const uitk_weights* w, const double* stats, void* out, size_t cap, size_t* needed) {
    if (!cfg || !w || !needed || !stats) return fail(UITK_ERR_INVALID_ARG, "null argument");
    const char* why = "";
    if (int rc = validate_config(cfg, &why)) return fail(rc, why);
    if (!calibratable_precision(cfg->precision))
        return fail(UITK_ERR_UNSUPPORTED, "unknown precision");
    std::vector<char> img;
    ImageHeader h;
    if (int rc = pack_image(cfg, w, stats, &img, &h)) return fail(rc, "pack failed (statistics with a non-finite / negative diagonal, or out of memory)");
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
    const uitk_config pc = pack_config(cfg);
    if (int rc = validate_config(&pc, &why)) return fail(rc, why);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(UITK_ERR_NO_DEVICE, "no HIP device: libuitk has no CPU path");
    uitk_model* m = new uitk_model();
    m->cfg = *cfg;
    std::vector<char> img;
    if (int rc = pack_image(&pc, w, nullptr, &img, &m->hdr)) { delete m; return fail(rc, "pack failed"); }
    m->img_nearest = img;
    hipDeviceProp_t prop;
    if (hipGetDevice(&m->device) != hipSuccess || hipGetDeviceProperties(&prop, m->device) != hipSuccess) {
        delete m; return fail(UITK_ERR_NO_DEVICE, "cannot query HIP device");
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::string arch = prop.gcnArchName; delete m;
        return fail(UITK_ERR_NO_DEVICE, "libuitk is built for gfx950 only, found " + arch);
    }
    m->n_cu = prop.multiProcessorCount;
    for (int i = 0; i < (cfg->n_fft / 2 + 1) * NMEL; ++i) if (w->mel_fb[i] > m->fb_max) m->fb_max = w->mel_fb[i];
    if (!std::isfinite(m->fb_max)) m->fb_max = 0.f;
    {
        const float* sc = (const float*)(img.data() + m->hdr.off_bn_sc);
        m->bn_nonneg = 1;
        for (int i = 0; i < NMEL; ++i) if (!(sc[i] >= 0.f)) m->bn_nonneg = 0;
    }
    // Tuning / test options are set explicitly through uitk_set_option (no environment is read here, except under -DUITK_LAB where
    // the A/B scripts of tools/ drive a variant library by environment).
#ifdef UITK_LAB
    for (const char* name : {"enc_waves", "fe_wg_per_cu", "fe_asym_pm", "split_pe", "enc_generic_clamp"}) {
        std::string env = "UITK_";
        for (const char* c = name; *c; ++c) env += (char)std::toupper((unsigned char)*c);
        if (const char* e = std::getenv(env.c_str())) (void)uitk_set_option(m, name, std::atoi(e));
    }
#endif
    if (hipMalloc((void**)&m->image, img.size()) != hipSuccess || hipMalloc((void**)&m->amax, 2 * AMAX_SLOTS * 4) != hipSuccess) {
        uitk_destroy(m); return fail(UITK_ERR_HIP, "hipMalloc failed for the parameter image");
    }
    float armed[2 * AMAX_SLOTS];
    for (float& v : armed) v = -INFINITY;                                       // both amax slot sets start armed (-inf)
    if (hipMemcpy(m->image, img.data(), img.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->amax, armed, sizeof armed, hipMemcpyHostToDevice) != hipSuccess) {
        uitk_destroy(m); return fail(UITK_ERR_HIP, "parameter upload failed");
    }
    if (cfg->precision == UITK_PREC_F32) {
        if (int rc = f32_create(m, w)) { uitk_destroy(m); return rc; }
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
    f32_destroy(m);
    delete m;
    return UITK_OK;
}

// torch.stft (torchaudio Spectrogram): center=True pads n_fft/2 each side -> 1 + T/hop frames; center=False -> 1 + (T - n_fft)/hop
int uitk_num_frames(const uitk_model* m, int64_t T) {
    if (!m) return -1;
    if (m->cfg.center) return (int)(1 + T / m->cfg.hop_size);
    return T < m->cfg.n_fft ? 0 : (int)(1 + (T - m->cfg.n_fft) / m->cfg.hop_size);
}

int uitk_num_crops(const uitk_model* m, int64_t T) {
    if (!m) return -1;
    const int nF = uitk_num_frames(m, T), tl = m->cfg.target_length;
    return nF <= tl ? 1 : (nF + tl - 1) / tl;
}

static int check_shape(const uitk_model* m, int64_t B, int64_t T) {
    if (!m) return fail(UITK_ERR_INVALID_ARG, "null model");
    if (B < 1 || T < 1) return fail(UITK_ERR_INVALID_ARG, "B and T must be positive");
    if (T <= m->cfg.n_fft / 2) return fail(UITK_ERR_TOO_SHORT, "clip must be longer than n_fft/2 samples (reflect padding)");
    if (!m->cfg.center && T < m->cfg.n_fft) return fail(UITK_ERR_TOO_SHORT, "center=False: clip shorter than one n_fft window");
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
    // (growing a buffer frees the old one, which the device may still be reading: the only synchronisation in here)
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
    if (m->cfg.precision == UITK_PREC_F32) {                     // its front-end writes fp32 dB frames (the layout of uitk_mel_db), not 16-bit operands
        const size_t need = (size_t)B * uitk_num_frames(m, T) * NMEL * 4;
        if (need > m->mel_frames_cap) {
            HIP_TRY(hipDeviceSynchronize());
            if (m->mel_frames) (void)hipFree(m->mel_frames);
            m->mel_frames = nullptr; m->mel_frames_cap = 0;
            HIP_TRY(hipMalloc((void**)&m->mel_frames, need));
            m->mel_frames_cap = need;
        }
        return f32_reserve(m, B * nc);
    }
    return UITK_OK;
}

static int run_frontend(uitk_model* m, const void* pcm, int in_dtype, int64_t B, int64_t T, float* mel_out, int patch, hipStream_t s);

// ---- the asymmetric pass split of the front-end (frontend.hip: the first-dispatched workgroup of a CU is served first by the
// instruction arbiter and gets a larger pool of passes) depends on dispatch order and arbitration, i.e. on the device and on the
// number of passes a workgroup walks.  It is MEASURED: five candidate shares, round-robin after a clock warm-up, smallest median
// wins.  Round 4: one result per (device, workgroups per CU, passes-per-workgroup bucket) for the whole PROCESS -- every handle
// shares it; measured on the caller's own input buffer (the kernel has no data-dependent control flow) and stream, with event waits
// only: no allocation, no hipDeviceSynchronize.  Who measures: uitk_tune_frontend (explicit), or -- uitk_config.fe_autotune = 1 --
// the first uitk_frontend / uitk_forward of a shape nobody has measured yet (~45 ms once per process and shape; never under stream
// capture).  Results never depend on the split (tests/test_gpu_parity.py::test_pass_distribution_does_not_change_results).
static std::mutex g_fe_mu;
static std::map<std::tuple<int, int, int>, int> g_fe_split;      // (device, workgroups per CU, bucket) -> per-mille share

static bool fe_split_key(const uitk_model* m, int64_t B, int64_t T, std::tuple<int, int, int>* key) {
    // (the split is MEASURED on at most 8192 clips: larger batches share that measurement, and its key -- ADVICE r4: the key used to be
    // derived from the full batch while the tuning ran on the capped one)
    if (B > 8192) B = 8192;
    const int nF = uitk_num_frames(m, T), rem = nF & 15;
    const int64_t npass = B * (int64_t)(nF >> 4) + (rem ? (B + 16 / rem - 1) / (16 / rem) : 0);
    const int wg = m->fe_wg_per_cu > 0 ? m->fe_wg_per_cu : 2;
    if (wg != 2 || uitk_num_crops(m, T) != 1 || npass < 4 * (int64_t)m->n_cu * wg) return false;   // no asymmetric split at this size
    const double per_wg = (double)npass / ((double)m->n_cu * wg);
    *key = std::make_tuple(m->device, wg, (int)std::lround(2.0 * std::log2(per_wg)));
    return true;
}

static int tune_frontend_split(uitk_model* m, const void* pcm, int in_dtype, int64_t B, int64_t T, hipStream_t s, int* best_out) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int best = 590;
    const int keep = m->fe_asym_cur;
    if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        // A cold device ramps its clocks over the first milliseconds of work, and a candidate measured during the ramp loses (round 3:
        // fresh processes picked 590, 630 or 670 on the same kind of box, the wrong pick costing 5 % of the kernel).  So: 48 untimed
        // launches first, then the candidates ROUND-ROBIN (9 rounds), median per candidate.
        constexpr int NC = 5, ROUNDS = 9;
        const int cands[NC] = {500, 550, 590, 630, 670};
        float t[NC][ROUNDS];
        bool ok = true;
        m->fe_asym_cur = 590;
        for (int i = 0; i < 48 && ok; ++i) ok = run_frontend(m, pcm, in_dtype, B, T, m->mel, 1, s) == UITK_OK;
        for (int r = 0; r < ROUNDS && ok; ++r)
            for (int ci = 0; ci < NC && ok; ++ci) {
                m->fe_asym_cur = cands[ci];
                ok = run_frontend(m, pcm, in_dtype, B, T, m->mel, 1, s) == UITK_OK;       // (the first launch of a split: untimed)
                (void)hipEventRecord(e0, s);
                ok = ok && run_frontend(m, pcm, in_dtype, B, T, m->mel, 1, s) == UITK_OK;
                (void)hipEventRecord(e1, s);
                ok = ok && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&t[ci][r], e0, e1) == hipSuccess;
            }
        if (ok) {
            float best_ms = 1e30f;
            for (int ci = 0; ci < NC; ++ci) {
                std::sort(t[ci], t[ci] + ROUNDS);
                if (t[ci][ROUNDS / 2] < best_ms) { best_ms = t[ci][ROUNDS / 2]; best = cands[ci]; }
            }
        }
        (void)hipGetLastError();
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    m->fe_asym_cur = keep;
    m->fe_B = 0; m->fe_T = 0;
    *best_out = best;
    return UITK_OK;
}

// the share this launch uses: the explicit option, else the process-wide measurement for this shape (measuring it now if allowed)
static void pick_frontend_split(uitk_model* m, const void* pcm, int in_dtype, int64_t B, int64_t T, hipStream_t s, bool may_tune) {
    if (m->fe_asym_pm >= 0) { m->fe_asym_cur = m->fe_asym_pm; return; }
    std::tuple<int, int, int> key;
    if (!fe_split_key(m, B, T, &key)) { m->fe_asym_cur = -1; return; }
    {
        std::lock_guard<std::mutex> lk(g_fe_mu);
        auto it = g_fe_split.find(key);
        if (it != g_fe_split.end()) { m->fe_asym_cur = it->second; return; }
    }
    m->fe_asym_cur = -1;
    if (!may_tune) return;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }
    int best = 590;
    (void)tune_frontend_split(m, pcm, in_dtype, B < 8192 ? B : 8192, T, s, &best);
    std::lock_guard<std::mutex> lk(g_fe_mu);
    g_fe_split.emplace(key, best);                                  // (a concurrent handle may have been first: its result stands)
    m->fe_asym_cur = g_fe_split[key];
}

int uitk_tune_frontend(uitk_model* m, const void* pcm, int in_dtype, int64_t B, int64_t T, void* stream) {
    if (int rc = check_shape(m, B, T)) return rc;
    if (!pcm) return fail(UITK_ERR_INVALID_ARG, "null pcm");
    if (in_dtype != UITK_IN_F32 && in_dtype != UITK_IN_I16) return fail(UITK_ERR_INVALID_ARG, "unknown in_dtype");
    if (int rc = uitk_reserve(m, B, T)) return rc;
    std::tuple<int, int, int> key;
    if (!fe_split_key(m, B, T, &key)) return UITK_OK;               // nothing to measure at this size
    int best = 590;
    if (int rc = tune_frontend_split(m, pcm, in_dtype, B < 8192 ? B : 8192, T, (hipStream_t)stream, &best)) return rc;
    std::lock_guard<std::mutex> lk(g_fe_mu);
    g_fe_split[key] = best;                                         // an explicit measurement replaces an earlier one
    if (m->fe_asym_pm < 0) m->fe_asym_cur = best;
    return UITK_OK;
}

float* uitk_amax_ptr(uitk_model* m) { return m ? m->amax + m->amax_cur * AMAX_SLOTS : nullptr; }

int uitk_set_debug_tokens(uitk_model* m, float* dev_buf) {
    if (!m) return fail(UITK_ERR_INVALID_ARG, "null model");
#ifndef UITK_LAB                                                    // (lab builds: allowed -- columns 1..24 land in rows 0..23; the cls and the dead columns all write row 0)
    if (dev_buf && m->cfg.pooling == 2) return fail(UITK_ERR_UNSUPPORTED, "the token dump is laid out for 24 tokens per clip (not for pooling 'token')");
#endif
    m->dbg = dev_buf;
    return UITK_OK;
}

int uitk_set_timing_buffers(uitk_model* m, unsigned long long* enc_buf, unsigned long long* fe_buf) {
    if (!m) return fail(UITK_ERR_INVALID_ARG, "null model");
#ifndef UITK_LAB
    if (enc_buf || fe_buf) return fail(UITK_ERR_UNSUPPORTED, "the instrumented kernels exist in -DUITK_LAB builds of the library only (tools/mklib.sh)");
#else
    if ((enc_buf || fe_buf) && ((m->cfg.precision != UITK_PREC_F16 && m->cfg.precision != UITK_PREC_F16L6) || m->cfg.attention_full || m->cfg.act_gelu || m->cfg.pooling))
        return fail(UITK_ERR_UNSUPPORTED, "the instrumented kernel builds exist for precision f16, BNeckAttention + ReLU only");
#endif
    m->timing = enc_buf; m->timing_fe = fe_buf;
    return UITK_OK;
}

int uitk_set_option(uitk_model* m, const char* name, int value) {
    if (!m || !name) return fail(UITK_ERR_INVALID_ARG, "null argument");
    const std::string n = name;
    if (n == "fe_asym_pm") {                  // per-mille share of the front-end passes for the first-dispatched workgroups; -1 = calibrate
        if (value != -1 && (value < 300 || value > 700)) return fail(UITK_ERR_INVALID_ARG, "fe_asym_pm: -1 or 300..700");
        m->fe_asym_pm = value; m->fe_asym_cur = value;
    } else if (n == "fe_wg_per_cu") {         // persistent front-end workgroups per CU (0 = the kernel's occupancy target)
        if (value < 0 || value > 4) return fail(UITK_ERR_INVALID_ARG, "fe_wg_per_cu: 0..4");
        m->fe_wg_per_cu = value;
    } else if (n == "enc_waves") {            // 8 = one encoder workgroup per CU (default), 4 = two independent workgroups per CU
        if (value != 0 && value != 4 && value != 8) return fail(UITK_ERR_INVALID_ARG, "enc_waves: 0 (by batch size), 4 or 8");
        m->enc_waves = value;
    } else if (n == "split_pe") {             // patch embed on split weights: 1 / 0 / -1 = the precision's default
        m->split_pe_env = value < 0 ? -1 : (value != 0);
    } else if (n == "enc_generic_clamp") {    // 1 = take the generic top_db clamp path even when every init_bn scale is >= 0 (tests)
        m->force_generic_clamp = value != 0;
    } else if (n == "fe_tail_skip") {         // 0 = transform every leftover frame even when it provably cannot raise the batch amax (tests)
        m->fe_tail_skip = value != 0;
    } else if (n == "calib_accept_pct") {     // 0 = uitk_calibrate_weights installs its image without the held-out comparison
        if (value < 0 || value > 1000) return fail(UITK_ERR_INVALID_ARG, "calib_accept_pct: 0..1000");
        m->calib_accept_pct = value;
    } else if (n == "calib_short_check") {    // 1: the held-out check also scores its clips at half length; 0 (default): full clips only
        if (value < 0 || value > 1) return fail(UITK_ERR_INVALID_ARG, "calib_short_check: 0 or 1");
        m->calib_short_check = value;
    } else if (n == "calib_budget_e6") {      // absolute deviation budget of the held-out check, 1e-6 units (0 = none)
        if (value < 0 || value > 1000000) return fail(UITK_ERR_INVALID_ARG, "calib_budget_e6: 0..1000000");
        m->calib_budget_e6 = value;
    } else {
        return fail(UITK_ERR_UNSUPPORTED, "unknown option: " + n);
    }
    return UITK_OK;
}

int uitk_get_option(const uitk_model* m, const char* name, int* value) {
    if (!m || !name || !value) return fail(UITK_ERR_INVALID_ARG, "null argument");
    const std::string n = name;
    if (n == "fe_asym_pm") *value = m->fe_asym_pm >= 0 ? m->fe_asym_pm : m->fe_asym_cur;      // explicit, else what the latest launch used
    else if (n == "weights_calibrated") *value = m->weights_calibrated;
    else if (n == "fe_wg_per_cu") *value = m->fe_wg_per_cu;
    else if (n == "enc_waves") *value = m->enc_waves;
    else if (n == "split_pe") *value = m->split_pe_env;
    else if (n == "enc_generic_clamp") *value = m->force_generic_clamp;
    else if (n == "fe_tail_skip") *value = m->fe_tail_skip;
    else if (n == "calib_accept_pct") *value = m->calib_accept_pct;
    else if (n == "calib_budget_e6") *value = m->calib_budget_e6;
    else if (n == "calib_short_check") *value = m->calib_short_check;
    else if (n == "parity_escalate_to") *value = m->parity_escalate_to;
    else return fail(UITK_ERR_UNSUPPORTED, "unknown option: " + n);
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
    a.pad = m->cfg.center ? m->cfg.n_fft / 2 : 0;
    a.window = (const float*)(m->image + h.off_window);
    a.tw256 = (const float2*)(m->image + h.off_tw256);
    a.tw512 = (const float2*)(m->image + h.off_tw512);
    a.mel_b = (const float*)(m->image + h.off_mel_b);
    a.mel_steps = h.mel_steps;
    // no memset per call: this launch fills one set of slots and re-arms (0xFFFFFFFF) the other for the next launch
    m->amax_cur ^= 1;
    a.mel = mel_out; a.amax = m->amax + m->amax_cur * AMAX_SLOTS; a.amax_rearm = m->amax + (m->amax_cur ^ 1) * AMAX_SLOTS;
    a.asym_pm = m->fe_asym_cur;
    a.fb_max = m->fe_tail_skip ? m->fb_max : 0.f;
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
    if (m->cfg.precision == UITK_PREC_F32) {                     // fp32 dB frames for the fp32 forward (no pass-split tuning: not a timed path)
        if (int rc = run_frontend(m, pcm, in_dtype, B, T, m->mel_frames, 0, (hipStream_t)stream)) return rc;
        m->fe_B = B; m->fe_T = T;
        return UITK_OK;
    }
    pick_frontend_split(m, pcm, in_dtype, B, T, (hipStream_t)stream, m->cfg.fe_autotune != 0);
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
    a.timing = m->timing;
    a.bn_nonneg = m->bn_nonneg && !m->force_generic_clamp;
    a.split_pe = m->split_pe_env >= 0 ? m->split_pe_env : (m->cfg.precision == UITK_PREC_F16X2 || m->cfg.precision == UITK_PREC_F16AX);     // f16x2 / f16ax: the patch embed runs on its split weights too
    // (f16l6 keeps the structurally rounded single weights: same-box, worst sweep seeds 6.0e-4 / 5.1e-4 without vs 6.6e-4 / 4.8e-4 with, +1.2 % time)
    a.attention_full = m->cfg.attention_full; a.act_gelu = m->cfg.act_gelu; a.pooling = m->cfg.pooling;
    if (m->cfg.precision == UITK_PREC_F32) {                     // the last rung of the parity ladder: the plain fp32 forward (calibrate.hip)
        if (a.dbg || a.timing) return fail(UITK_ERR_UNSUPPORTED, "precision f32 has no token dump / stamps");
        if (int rc = f32_encode(m, a.rows, a.nt, nF, nc, a.probs, a.logits, s)) return rc;
        if (nc > 1) HIP_TRY(launch_crop_reduce(m->rowprobs, probs, B, nc, a.C, m->cfg.eval_avg, s));
        return UITK_OK;
    }
    // Workgroup shape.  Large batches: 8 waves (two per SIMD, every weight fragment read once per 16 clips).  A batch that fits ONE
    // 4-wave workgroup per CU (rows <= 8 per CU) runs on those instead: a lone wave per SIMD finishes its two clips 1.3-1.4x sooner
    // than a pair sharing the SIMD finishes four (XS: 0.14-0.15 ms against 0.20 ms up to 2048 clips; profiles/round4/occ_test.log),
    // which is what a small-batch / latency caller wants.  The same holds for the LAST round of a large batch: rows beyond the full
    // 8-wave rounds (16 per CU each) that fit one 4-wave workgroup per CU go out as a second launch of that shape (rows are
    // independent; both launches read the same amax slots).  Same arithmetic, same bits (tests: test_pass_distribution_...).
    // Only the uit_* family in f16 / bf16 / fp8 has 4-wave instantiations; launch_encoder ignores the hint for the others.
    // (not with the split patch embed switched on: only 8-wave workgroups have the LDS for its low-order fragments, and a batch must
    // not change its arithmetic with its size)
    const bool has_w4 = !a.attention_full && !a.act_gelu && !a.pooling && !a.split_pe &&
                        (kernel_precision(m->cfg.precision) == UITK_PREC_F16 || m->cfg.precision == UITK_PREC_BF16 || m->cfg.precision == UITK_PREC_FP8);
    const int small = 8 * m->n_cu, round8 = 16 * m->n_cu;
    auto launch_rows = [&](int row0, int nrows, int waves) {
        EncoderArgs b = a;
        b.rows = nrows;
        b.mel = a.mel + (size_t)row0 * NTOK * 256;
        b.probs = a.probs + (size_t)row0 * a.C;
        if (a.logits) b.logits = a.logits + (size_t)row0 * a.C;
        return launch_encoder(b, kernel_precision(m->cfg.precision), waves, s);
    };
    const int tail = a.rows % round8;
    if (m->enc_waves || !has_w4 || a.dbg || a.timing) {
        HIP_TRY(launch_encoder(a, kernel_precision(m->cfg.precision), m->enc_waves ? m->enc_waves : 8, s));
    } else if (a.rows <= small) {
        HIP_TRY(launch_encoder(a, kernel_precision(m->cfg.precision), 4, s));
    } else if (a.rows > round8 && tail > 0 && tail <= small) {
        HIP_TRY(launch_rows(0, a.rows - tail, 8));
        HIP_TRY(launch_rows(a.rows - tail, tail, 4));
    } else {
        HIP_TRY(launch_encoder(a, kernel_precision(m->cfg.precision), 8, s));
    }
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

// ---- calibrated rounding of the weights (pack.cpp: round_gptq; calibrate.hip) ---------------------------------
// The library's own calibration signals (used when the caller passes none): 32 clips of eight kinds -- white noise at four levels,
// pure tones, noise with a silent tail, near-silence, chirps, low-passed noise, amplitude-modulated noise bursts, vibrato harmonic
// stacks with a syllable-rate envelope -- deterministic (an LCG), int16.  They only have to EXCITE the directions real audio excites:
// the damping of round_gptq bounds what the error feedback may do along directions they leave out.
static void builtin_calibration_set(int n, int64_t T, std::vector<int16_t>* out) {
    out->assign((size_t)n * T, 0);
    uint32_t lcg = 0x2545F491u;
    auto noise = [&]() { lcg = lcg * 1664525u + 1013904223u; return (double)(lcg >> 8) / 16777216.0 - 0.5; };
    std::vector<double> x(T);
    for (int i = 0; i < n; ++i) {
        const int kind = i & 7, v = (i >> 3) & 3;
        const double sr = 16000.0;
        switch (kind) {
        case 0: { const double g[4] = {1.0, 0.3, 0.1, 0.03}; for (int64_t t = 0; t < T; ++t) x[t] = g[v] * noise(); break; }
        case 1: { const double f = 220.0 * std::pow(1.9, v) + 37.0 * v;
                  for (int64_t t = 0; t < T; ++t) x[t] = 0.9 * std::sin(2.0 * M_PI * f * t / sr) + 1e-3 * noise(); break; }
        case 2: { const int64_t cut = T - T / 4;
                  for (int64_t t = 0; t < T; ++t) x[t] = t < cut ? (0.25 + 0.25 * v) * noise() : 0.0; break; }
        case 3: { for (int64_t t = 0; t < T; ++t) x[t] = 1e-3 * (v + 1) * noise(); break; }
        case 4: { const bool down = (v & 1) != 0;
                  const double f0 = down ? 7500.0 : 100.0, f1 = down ? 100.0 : 7500.0, dur = (double)T / sr;
                  for (int64_t t = 0; t < T; ++t) { const double tt = t / sr; x[t] = (v < 2 ? 0.5 : 0.05) * std::sin(2.0 * M_PI * (f0 + 0.5 * (f1 - f0) * tt / dur) * tt); }
                  break; }
        case 5: { const double a0[4] = {0.9, 0.97, 0.99, 0.7}; const double a = a0[v]; double y = 0.0, pk = 1e-9;
                  for (int64_t t = 0; t < T; ++t) { y = a * y + noise(); x[t] = y; pk = std::fmax(pk, std::fabs(y)); }
                  for (int64_t t = 0; t < T; ++t) x[t] *= 0.6 / pk;
                  break; }
        case 6: { for (int64_t t = 0; t < T; ++t) { const double e = 0.5 + 0.5 * std::sin(2.0 * M_PI * (2.0 + 3.0 * v) * t / sr); x[t] = 0.8 * e * e * e * e * noise(); } break; }
        default: { const double f0 = 100.0 + 40.0 * v;
                   for (int64_t t = 0; t < T; ++t) {
                       const double tt = t / sr, ph = 2.0 * M_PI * (f0 * tt + 0.8 * std::sin(2.0 * M_PI * 5.0 * tt));
                       double acc = 0.0;
                       for (int hh = 1; hh <= 24; ++hh) {
                           const double fh = hh * f0, form = std::exp(-0.5 * std::pow((fh - 700.0 - 300.0 * v) / 400.0, 2)) + 0.5 * std::exp(-0.5 * std::pow((fh - 2200.0) / 600.0, 2));
                           acc += (0.15 + form) / hh * std::sin(hh * ph);
                       }
                       const double env = std::pow(0.5 + 0.5 * std::sin(2.0 * M_PI * 3.0 * tt), 2);
                       x[t] = 0.25 * env * acc + 2e-4 * noise();
                   }
                   break; }
        }
        for (int64_t t = 0; t < T; ++t) {
            double s = std::nearbyint(x[t] * 32768.0);
            s = s > 32767.0 ? 32767.0 : s < -32768.0 ? -32768.0 : s;
            (*out)[(size_t)i * T + t] = (int16_t)s;
        }
    }
}

// The HELD-OUT clips of the acceptance / self check (round 5: a DIFFERENT family from the calibration signals above, not the same
// generator with other parameters -- VERDICT r4 item 1b): two variants each of sixteen kinds of signal a fit on noise, tones, chirps and harmonic stacks leaves
// unconstrained -- impulse train, single full-scale click, DC offsets (+0.5 with -40 dB noise, -0.5 with a tone), hard-clipped tone,
// +-1 LSB dither, Nyquist alternation (half scale, and at 1 LSB), AM/FM harmonic stacks at -40 and -12 dBFS, full-scale noise driven into
// clipping, 50 Hz square wave, silence ending in a loud burst, exponential sweep, two tones on a bin edge, re-struck decaying
// resonances -- and, behind them, sixteen ordinary signals (see the loop).  Deterministic (own LCG), int16.  One batch: the loud clips set the batch-global top_db floor for the quiet ones, as a
// mixed batch of a caller does.
static void builtin_heldout_set(int n, int64_t T, std::vector<int16_t>* out) {
    out->assign((size_t)n * T, 0);
    uint32_t lcg = 0x9E3779B9u;
    auto noise = [&]() { lcg = lcg * 1664525u + 1013904223u; return (double)(lcg >> 8) / 16777216.0 - 0.5; };
    const double sr = 16000.0, lsb = 1.0 / 32768.0;
    std::vector<double> x(T);
    for (int i = 0; i < n; ++i) {
        // clips 0..31: two variants of the sixteen adversarial kinds; clips 32..: ORDINARY signals (kinds 16..31: noise at several levels and
        // with a silent tail, chirps, tone stacks, brown / band-limited noise) from this generator's own seeds -- the worst input of a
        // checkpoint is as often a plain noise clip as an exotic one (tools/auto_sweep.py: sixteen adversarial clips alone estimated 5.3e-4
        // where a noise clip of the test set measured 9.4e-4), and the estimate is a maximum: it grows with what it is taken over
        const int kind = i < 32 ? (i & 15) : 16 + ((i - 32) & 15), v = i < 32 ? (i >> 4) : ((i - 32) >> 4);
        std::fill(x.begin(), x.end(), 0.0);
        switch (kind) {
        case 0: { const int64_t per = 173 + 31 * v; int sgn = 1; for (int64_t t = per / 2; t < T; t += per) { x[t] = 0.9 * sgn; sgn = -sgn; } break; }
        case 1: { x[T / 3 + 7 * v] = 32767.0 * lsb; break; }
        case 2: { for (int64_t t = 0; t < T; ++t) x[t] = 0.5 + 0.02 * noise(); break; }
        case 3: { for (int64_t t = 0; t < T; ++t) x[t] = -0.5 + 0.3 * std::sin(2.0 * M_PI * (440.0 + 60.0 * v) * t / sr); break; }
        case 4: { for (int64_t t = 0; t < T; ++t) { const double y = 3.0 * std::sin(2.0 * M_PI * (311.0 + 40.0 * v) * t / sr); x[t] = 0.98 * (y > 1.0 ? 1.0 : y < -1.0 ? -1.0 : y); } break; }
        case 5: { for (int64_t t = 0; t < T; ++t) x[t] = std::nearbyint(3.0 * noise()) * lsb; break; }
        case 6: { const double a6 = v ? 0.06 : 0.5; for (int64_t t = 0; t < T; ++t) x[t] = (t & 1) ? -a6 : a6; break; }
        case 7: { for (int64_t t = 0; t < T; ++t) x[t] = ((t & 1) ? -1.0 : 1.0) * lsb + std::nearbyint(1.2 * noise()) * lsb; break; }
        case 8: case 9: {
            const double lvl = kind == 8 ? 0.01 : 0.25, f0 = kind == 8 ? 131.0 + 17.0 * v : 197.0 + 13.0 * v;
            for (int64_t t = 0; t < T; ++t) {
                const double tt = t / sr, ph = 2.0 * M_PI * (f0 * tt + 1.1 * std::sin(2.0 * M_PI * 4.3 * tt));
                double acc = 0.0;
                for (int hh = 1; hh <= 18; ++hh) {
                    const double fh = hh * f0, form = std::exp(-0.5 * std::pow((fh - 520.0) / 300.0, 2)) + 0.6 * std::exp(-0.5 * std::pow((fh - 1700.0) / 500.0, 2)) + 0.3 * std::exp(-0.5 * std::pow((fh - 3100.0) / 700.0, 2));
                    acc += (0.1 + form) / hh * std::sin(hh * ph);
                }
                const double env = std::pow(0.5 + 0.5 * std::sin(2.0 * M_PI * (3.7 + 0.9 * (kind - 8)) * tt + 1.0), 2);
                x[t] = lvl * env * acc;
            }
            break; }
        case 10: { for (int64_t t = 0; t < T; ++t) x[t] = 2.6 * noise(); break; }                                      // clips ~23 % of the samples
        case 11: { for (int64_t t = 0; t < T; ++t) x[t] = std::sin(2.0 * M_PI * (50.0 + 7.0 * v) * t / sr) >= 0.0 ? 0.8 : -0.8; break; }
        case 12: { const int64_t on = v ? T / 2 : T - T / 10; for (int64_t t = on; t < T; ++t) x[t] = 0.9 * noise() + 0.4 * std::sin(2.0 * M_PI * 1250.0 * t / sr); break; }
        case 13: { const double dur = (double)T / sr, k = std::log(7900.0 / 20.0) / dur;
                   for (int64_t t = 0; t < T; ++t) { const double tt = (v ? T - 1 - t : t) / sr; x[t] = (v ? 0.05 : 0.5) * std::sin(2.0 * M_PI * 20.0 * (std::exp(k * tt) - 1.0) / k); }
                   break; }
        case 14: { const double fb = v ? 2500.0 : 1000.0;     // two tones half a bin apart, on a bin edge
                   for (int64_t t = 0; t < T; ++t) x[t] = 0.4 * std::sin(2.0 * M_PI * fb * t / sr) + 0.4 * std::sin(2.0 * M_PI * (fb + 15.625) * t / sr + 0.7); break; }
        case 16: case 17: case 18: case 19: { const double g4[4] = {1.0, 0.5, 0.2, 0.04}; for (int64_t t = 0; t < T; ++t) x[t] = g4[kind - 16] * noise(); break; }
        case 20: case 21: { const int64_t cut = T - T / 4; for (int64_t t = 0; t < cut; ++t) x[t] = (kind == 20 ? 1.0 : 0.3) * noise(); break; }       // silent tail: the dB floor
        case 22: case 23: { const double f0 = kind == 22 ? 60.0 : 7800.0, f1 = kind == 22 ? 7800.0 : 60.0, dur = (double)T / sr;
                            for (int64_t t = 0; t < T; ++t) { const double tt = t / sr; x[t] = 0.45 * std::sin(2.0 * M_PI * (f0 + 0.5 * (f1 - f0) * tt / dur) * tt) + 2e-3 * noise(); }
                            break; }
        case 24: case 25: case 26: { const double fs[3][3] = {{233.0, 1870.0, 4310.0}, {97.0, 610.0, 3333.0}, {1450.0, 2900.0, 6100.0}};
                            for (int64_t t = 0; t < T; ++t) { double acc = 0.0; for (int r = 0; r < 3; ++r) acc += std::sin(2.0 * M_PI * fs[kind - 24][r] * t / sr + r); x[t] = 0.28 * acc + 1e-3 * noise(); }
                            break; }
        case 27: case 28: { const double a = kind == 27 ? 0.995 : 0.9; double y = 0.0, pk = 1e-9;                                   // brown / low-passed noise
                            for (int64_t t = 0; t < T; ++t) { y = a * y + noise(); x[t] = y; pk = std::fmax(pk, std::fabs(y)); }
                            for (int64_t t = 0; t < T; ++t) x[t] *= 0.7 / pk;
                            break; }
        case 29: { for (int64_t t = 0; t < T; ++t) { const double e = 0.5 + 0.5 * std::sin(2.0 * M_PI * 7.0 * t / sr); x[t] = 0.9 * e * e * noise(); } break; }
        case 30: { double y1 = 0.0, y2 = 0.0;                                                                                     // resonant (band-passed) noise around 3 kHz
                   const double rr = 0.98, th = 2.0 * M_PI * 3000.0 / sr;
                   for (int64_t t = 0; t < T; ++t) { const double y = 2.0 * rr * std::cos(th) * y1 - rr * rr * y2 + noise(); y2 = y1; y1 = y; x[t] = 0.02 * y; }
                   break; }
        case 31: { for (int64_t t = 0; t < T; ++t) x[t] = 0.3 * noise() + 0.3 * std::sin(2.0 * M_PI * 440.0 * t / sr) + (t > T / 2 ? 0.2 * std::sin(2.0 * M_PI * 5200.0 * t / sr) : 0.0); break; }
        default: { const double fr[3] = {330.0 + 20.0 * v, 1240.0, 2750.0}, dec[3] = {9.0, 14.0, 22.0};
                   for (int64_t t = 0; t < T; ++t) {
                       const double tt = std::fmod(t / sr, 0.25);
                       double acc = 0.0;
                       for (int r = 0; r < 3; ++r) acc += std::exp(-dec[r] * tt) * std::sin(2.0 * M_PI * fr[r] * tt) / (r + 1);
                       x[t] = 0.6 * acc;
                   }
                   break; }
        }
        for (int64_t t = 0; t < T; ++t) {
            double s = std::nearbyint(x[t] * 32768.0);
            s = s > 32767.0 ? 32767.0 : s < -32768.0 ? -32768.0 : s;
            (*out)[(size_t)i * T + t] = (int16_t)s;
        }
    }
}

namespace {
struct DevBuf {                                                  // scoped device allocation of the pack-time helpers
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

// The folded network (pack.h) as one float array on the device + where everything sits in it: what the plain fp32 forward of
// calibrate.hip multiplies with.
struct NaiveNet {
    DevBuf weights, work;
    CalibArgs a{};
    size_t n_work_per_token = 0;
};

int naive_net_upload(const uitk_model* m, const uitk_weights* w, NaiveNet* nn, hipStream_t s) {
    Folded F;
    fold_model(&m->cfg, w, &F);
    const int inner = F.inner;
    CalibArgs& a = nn->a;
    a.o_qkv_w = 0; a.o_qkv_b = a.o_qkv_w + (size_t)3 * inner * DIM; a.o_proj_w = a.o_qkv_b + 3 * inner; a.o_proj_b = a.o_proj_w + (size_t)DIM * inner;
    a.o_fc1_w = a.o_proj_b + DIM; a.o_fc1_b = a.o_fc1_w + (size_t)HID * DIM; a.o_fc2_w = a.o_fc1_b + HID; a.o_fc2_b = a.o_fc2_w + (size_t)DIM * HID;
    a.block_floats = a.o_fc2_b + DIM;
    const size_t n_model = (size_t)DIM * 256 + (size_t)(NTOK + 1) * DIM + (size_t)m->cfg.depth * a.block_floats;
    std::vector<float> hostw(n_model);
    {
        float* p = hostw.data();
        for (double v : F.pe_w) *p++ = (float)v;
        for (double v : F.pe_bias) *p++ = (float)v;
        for (const FoldedBlock& fb : F.blocks)
            for (const std::vector<double>* arr : {&fb.qkv_w, &fb.qkv_b, &fb.proj_w, &fb.proj_b, &fb.fc1_w, &fb.fc1_b, &fb.fc2_w, &fb.fc2_b})
                for (double v : *arr) *p++ = (float)v;
    }
    HIP_TRY(hipMalloc(&nn->weights.p, n_model * 4));
    HIP_TRY(hipMemcpyAsync(nn->weights.p, hostw.data(), n_model * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));                            // (hostw goes out of scope)
    const float* dw = (const float*)nn->weights.p;
    a.top_db = m->cfg.top_db;
    a.bn_sc = (const float*)(m->image + m->hdr.off_bn_sc); a.bn_sh = (const float*)(m->image + m->hdr.off_bn_sh);
    a.op_bf16 = m->cfg.precision == UITK_PREC_BF16;
    a.ntok = m->cfg.pooling == 2 ? NTOK + 1 : NTOK; a.depth = m->cfg.depth; a.inner = inner; a.act_gelu = m->cfg.act_gelu;
    a.pe_w = dw; a.pe_bias = dw + (size_t)DIM * 256; a.blocks = a.pe_bias + (size_t)(NTOK + 1) * DIM;
    nn->n_work_per_token = (size_t)(256 + DIM + DIM + 3 * inner + inner + HID);
    return UITK_OK;
}

// work buffers for `rows` clip-crops (+ `n_stats` floats of statistics behind them, 0 = none: forward only)
int naive_net_bind(const uitk_model* m, NaiveNet* nn, int rows, size_t n_stats) {
    CalibArgs& a = nn->a;
    const size_t M = (size_t)rows * a.ntok;
    if (nn->work.p) { (void)hipFree(nn->work.p); nn->work.p = nullptr; }
    HIP_TRY(hipMalloc(&nn->work.p, (M * nn->n_work_per_token + n_stats) * 4));
    float* wk = (float*)nn->work.p;
    a.rows = rows;
    a.mel = (const unsigned short*)m->mel; a.amax = m->amax + m->amax_cur * AMAX_SLOTS;
    a.X0 = wk; a.R = a.X0 + M * 256; a.Xn = a.R + M * DIM; a.QKV = a.Xn + M * DIM; a.O = a.QKV + M * 3 * a.inner;
    a.Hd = a.O + M * a.inner; a.stats = n_stats ? a.Hd + M * HID : nullptr;
    return UITK_OK;
}

}  // namespace

// ---- precision 'f32' (UITK_PREC_F32; round 5): the LAST rung of the parity ladder.  The folded network in fp32 on the device
// (naive_net_upload: the weights the calibration forward multiplies with), the head's parameters, and work buffers for F32_CHUNK
// clip-crops at a time; uitk_encode walks the batch in chunks through launch_naive_forward (calibrate.hip).  Front-end, crop
// reduction, options and the C ABI are the product's.  ~13x slower than the MFMA encoder (round 5: 20x): for checkpoints no fp16
// mode holds inside the budget (tools/auto_sweep.py ... sink4), where slow and right beats fast and warned about.
constexpr int64_t F32_CHUNK = 8192;                         // (round 6: 2048 -> 8192 rows per chunk: the small kernels fill the chip; ~135 KB of work buffers per row)
struct F32Path {
    NaiveNet nn;
    DevBuf head;                                                 // norm w/b, outputlayer LayerNorm w/b, Linear w/b (fp32)
    NaiveHeadArgs hargs{};
    int64_t work_rows = 0;
};

static int f32_create(uitk_model* m, const uitk_weights* w) {
    F32Path* fp = new F32Path();
    m->f32 = fp;
    if (int rc = naive_net_upload(m, w, &fp->nn, nullptr)) return rc;
    const int C = m->cfg.outputdim;
    std::vector<float> host((size_t)4 * DIM + (size_t)C * DIM + C);
    float* p = host.data();
    std::memcpy(p, w->norm_w, DIM * 4); p += DIM; std::memcpy(p, w->norm_b, DIM * 4); p += DIM;
    std::memcpy(p, w->head_ln_w, DIM * 4); p += DIM; std::memcpy(p, w->head_ln_b, DIM * 4); p += DIM;
    std::memcpy(p, w->head_w, (size_t)C * DIM * 4); p += (size_t)C * DIM; std::memcpy(p, w->head_b, (size_t)C * 4);
    HIP_TRY(hipMalloc(&fp->head.p, host.size() * 4));
    HIP_TRY(hipMemcpy(fp->head.p, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    const float* d = (const float*)fp->head.p;
    NaiveHeadArgs& h = fp->hargs;
    h.norm_w = d; h.norm_b = d + DIM; h.hln_w = d + 2 * DIM; h.hln_b = d + 3 * DIM; h.head_w = d + 4 * DIM; h.head_b = h.head_w + (size_t)C * DIM;
    h.C = C; h.pooling = m->cfg.pooling;
    return UITK_OK;
}

static void f32_destroy(uitk_model* m) { delete m->f32; m->f32 = nullptr; }

static int f32_reserve(uitk_model* m, int64_t rows) {
    F32Path* fp = m->f32;
    const int64_t want = std::min<int64_t>(rows, F32_CHUNK);
    if (want <= fp->work_rows) return UITK_OK;
    HIP_TRY(hipDeviceSynchronize());                             // (growing frees a buffer the device may still be reading)
    if (int rc = naive_net_bind(m, &fp->nn, (int)want, 0)) return rc;
    fp->work_rows = want;
    return UITK_OK;
}

static int f32_encode(uitk_model* m, int rows, int nt, int nF, int ncrops, float* rowprobs, float* rowlogits, hipStream_t s) {
    F32Path* fp = m->f32;
    if (!fp || fp->work_rows < std::min<int64_t>(rows, F32_CHUNK)) return fail(UITK_ERR_WORKSPACE, "precision f32: no workspace (uitk_frontend / uitk_reserve first)");
    for (int64_t r0 = 0; r0 < rows; r0 += F32_CHUNK) {
        const int n = (int)std::min<int64_t>(F32_CHUNK, rows - r0);
        CalibArgs a = fp->nn.a;                                  // (the buffers are laid out for work_rows: a shorter chunk uses their heads)
        a.rows = n;
        a.amax = m->amax + m->amax_cur * AMAX_SLOTS;
        a.stats = nullptr;
        NaiveFramesArgs fa{};
        fa.frames = m->mel_frames; fa.nF = nF; fa.ncrops = ncrops; fa.tl = m->cfg.target_length; fa.row0 = r0;
        NaiveHeadArgs h = fp->hargs;
        h.rowprobs = rowprobs + (size_t)r0 * h.C;
        h.rowlogits = rowlogits ? rowlogits + (size_t)r0 * h.C : nullptr;
        HIP_TRY(launch_naive_forward(a, nt, &h, s, &fa));
    }
    return UITK_OK;
}

namespace {
// What the reference computes from the last block's residual stream (models/uit.py:395, 398-412), in double on the host:
// final LayerNorm (eps 1e-6) -> pooling -> outputlayer = LayerNorm (eps 1e-5) + Linear.  R: [rows][ntok][128] (the folded stream is
// centred per token, which LayerNorm does not see).  out: [rows][C] pre-sigmoid logits; pooling 'dm': probabilities (it has no logits).
void head_reference(const uitk_config& c, const uitk_weights* w, const float* R, int rows, int ntok, std::vector<double>* out, int nt = GT) {
    // nt < GT (short clips): only the tokens of time patches 0 .. nt-1 exist (naive_token_valid / naive_head_kernel in calibrate.hip)
    const int C = c.outputdim;
    out->assign((size_t)rows * C, 0.0);
    auto layernorm = [](const double* x, const float* g, const float* b, double eps, double* y) {
        double mean = 0.0, var = 0.0;
        for (int i = 0; i < DIM; ++i) mean += x[i];
        mean /= DIM;
        for (int i = 0; i < DIM; ++i) var += (x[i] - mean) * (x[i] - mean);
        const double rstd = 1.0 / std::sqrt(var / DIM + eps);
        for (int i = 0; i < DIM; ++i) y[i] = (x[i] - mean) * rstd * g[i] + b[i];
    };
    std::vector<double> tokn((size_t)ntok * DIM), z(DIM), zn(DIM), x(DIM);
    auto head = [&](const double* pooled, double* logits) {
        layernorm(pooled, w->head_ln_w, w->head_ln_b, 1e-5, zn.data());
        for (int o = 0; o < C; ++o) {
            double acc = w->head_b[o];
            for (int i = 0; i < DIM; ++i) acc += zn[i] * (double)w->head_w[(size_t)o * DIM + i];
            logits[o] = acc;
        }
    };
    std::vector<double> lg(C);
    for (int r = 0; r < rows; ++r) {
        for (int t = 0; t < ntok; ++t) {
            for (int i = 0; i < DIM; ++i) x[i] = R[((size_t)r * ntok + t) * DIM + i];
            layernorm(x.data(), w->norm_w, w->norm_b, 1e-6, tokn.data() + (size_t)t * DIM);
        }
        double* dst = out->data() + (size_t)r * C;
        if (c.pooling == 2) {                                    // 'token': the cls row
            head(tokn.data(), dst);
        } else if (c.pooling == 0) {                             // 'mean' over the valid tokens (all 24 of a full clip)
            for (int i = 0; i < DIM; ++i) {
                double acc = 0.0;
                for (int t = 0; t < NTOK; ++t) if (t % GT < nt) acc += tokn[(size_t)t * DIM + i];
                z[i] = acc / (double)((NTOK / GT) * nt);
            }
            head(z.data(), dst);
        } else {                                                 // 'dm': mean over frequency patches, head + sigmoid per time patch, mean over time
            for (int tp = 0; tp < nt; ++tp) {
                for (int i = 0; i < DIM; ++i) { double acc = 0.0; for (int f = 0; f < NTOK / GT; ++f) acc += tokn[(size_t)(f * GT + tp) * DIM + i]; z[i] = acc / (NTOK / GT); }
                head(z.data(), lg.data());
                for (int o = 0; o < C; ++o) dst[o] += 1.0 / (1.0 + std::exp(-lg[o])) / nt;
            }
        }
    }
}
}  // namespace

// The comparison behind the acceptance check, the absolute budget and uitk_self_check (host only, exported so that a CPU test can pin
// it): largest and root-mean-square deviation of `got` from `ref`.  A non-finite deviation is STICKY -- one NaN / Inf anywhere makes
// both results NaN, so `!(x <= budget)` and `isfinite(x)` see it whatever follows (`if (!(d <= mx)) mx = d` kept a NaN only
// until the next finite element replaced it -- an image that answers NaN on some held-out clips could pass as inside the budget).
int uitk_deviation_stats(const float* got, const double* ref, size_t n, double* out2) {
    if (!got || !ref || !out2) return fail(UITK_ERR_INVALID_ARG, "null argument");
    double mxv = 0.0, sq = 0.0;
    bool bad = false;
    for (size_t i = 0; i < n; ++i) {
        const double d = std::fabs((double)got[i] - ref[i]);
        if (!std::isfinite(d)) { bad = true; continue; }
        if (d > mxv) mxv = d;
        sq += d * d;
    }
    out2[0] = bad ? std::nan("") : mxv;
    out2[1] = bad ? std::nan("") : (n ? std::sqrt(sq / (double)n) : 0.0);
    return UITK_OK;
}

// The held-out evaluation shared by the acceptance check of uitk_calibrate_weights and by uitk_self_check: the 48 held-out clips of
// the library (builtin_heldout_set) go through the PRODUCT front-end; the plain fp32 forward of calibrate.hip + head_reference give
// what the unrounded network answers on those 16-bit patch operands; heldout_score runs the product encoder on whatever parameter
// image the handle holds and returns the largest / root-mean-square deviation (pre-sigmoid logits; probabilities for pooling 'dm').
namespace {
struct HeldOut {
    int Be = 48;                                                 // (round 5: 16 -> 48 clips, see builtin_heldout_set)
    int64_t Te = 0;                                              // 0: no single-crop clip length covers target_length (check does not apply)
    int nt = GT;                                                 // valid time patches of the clips (GT: full clips; 3: the half-length leg)
    DevBuf pcm, outb;
    std::vector<double> ref;
};

int heldout_prepare(uitk_model* m, const uitk_weights* w, NaiveNet* nn, hipStream_t s, HeldOut* ho, int want_nt = GT) {
    const int C = m->cfg.outputdim;
    ho->Te = 0;
    ho->nt = want_nt;
    for (int64_t k = want_nt * 16 - 1; k < 4096 && !ho->Te; ++k) {
        const int64_t T = k * (int64_t)m->cfg.hop_size + (m->cfg.center ? 0 : m->cfg.n_fft);
        const int nf = uitk_num_frames(m, T);
        const bool fits = want_nt == GT ? (nf >= GT * 16 && nf <= m->cfg.target_length) : (nf >= want_nt * 16 && nf < (want_nt + 1) * 16);
        if (uitk_num_crops(m, T) == 1 && fits) ho->Te = T;
        if (nf >= (want_nt + 1) * 16 && want_nt != GT) break;
    }
    if (!ho->Te) return UITK_OK;
    std::vector<int16_t> host;
    builtin_heldout_set(ho->Be, ho->Te, &host);
    HIP_TRY(hipMalloc(&ho->pcm.p, host.size() * 2));
    HIP_TRY(hipMemcpy(ho->pcm.p, host.data(), host.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&ho->outb.p, (size_t)2 * ho->Be * C * 4));
    if (int rc = uitk_frontend(m, ho->pcm.p, UITK_IN_I16, ho->Be, ho->Te, (void*)s)) return rc;
    if (int rc = naive_net_bind(m, nn, ho->Be, 0)) return rc;
    if (m->cfg.precision == UITK_PREC_F32) {                     // (its front-end left fp32 frames, not patch operands)
        NaiveFramesArgs fa{};
        fa.frames = m->mel_frames; fa.nF = uitk_num_frames(m, ho->Te); fa.ncrops = 1; fa.tl = m->cfg.target_length; fa.row0 = 0;
        HIP_TRY(launch_naive_forward(nn->a, want_nt, nullptr, s, &fa));
    } else {
        HIP_TRY(launch_naive_forward(nn->a, want_nt, nullptr, s, nullptr));
    }
    const int ntok = nn->a.ntok;
    std::vector<float> R((size_t)ho->Be * ntok * DIM);
    HIP_TRY(hipMemcpyAsync(R.data(), nn->a.R, R.size() * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    head_reference(m->cfg, w, R.data(), ho->Be, ntok, &ho->ref, want_nt);
    return UITK_OK;
}
// the legs of the held-out evaluation: full clips, and (option calib_short_check) clips of three time patches
int heldout_legs(const uitk_model* m, int* nts) {
    nts[0] = GT;
    if (m->calib_short_check && GT > 3) { nts[1] = 3; return 2; }
    return 1;
}
// maximum over the legs (a non-finite leg stays non-finite), root mean square over all their elements (legs of equal size)
void heldout_merge(double* mx, double* msq, int leg, double leg_mx, double leg_rms) {
    if (leg == 0) { *mx = leg_mx; *msq = leg_rms * leg_rms; return; }
    *mx = (std::isfinite(*mx) && std::isfinite(leg_mx)) ? std::max(*mx, leg_mx) : std::nan("");
    *msq += leg_rms * leg_rms;
}

// the product encoder on the operands heldout_prepare left in the workspace, with the image the handle holds right now
int heldout_score(uitk_model* m, HeldOut* ho, hipStream_t s, double* mx, double* rms) {
    const int C = m->cfg.outputdim;
    float* d_probs = (float*)ho->outb.p;
    float* d_logits = m->cfg.pooling == 1 ? nullptr : d_probs + (size_t)ho->Be * C;
    std::vector<float> got((size_t)ho->Be * C);
    if (int rc = uitk_encode(m, ho->Be, ho->Te, d_probs, d_logits, (void*)s)) return rc;
    HIP_TRY(hipMemcpyAsync(got.data(), d_logits ? d_logits : d_probs, got.size() * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    double st[2];
    uitk_deviation_stats(got.data(), ho->ref.data(), got.size(), st);
    *mx = st[0]; *rms = st[1];
    return UITK_OK;
}

// the budget a deviation on the held-out clips is compared with (option calib_budget_e6; pooling 'dm' is compared on probabilities,
// whose slope against the logit is at most 1/4)
double parity_budget(const uitk_model* m) { return m->calib_budget_e6 * 1e-6 * (m->cfg.pooling == 1 ? 0.25 : 1.0); }
// over budget: pooling 'token' reads ONE row, what is left there is activation rounding -> split activation operands (f16a), and from
// there split weights as well (f16ax: the top of that fp16 ladder); the other poolings: split weights (f16l6 for BNeckAttention + ReLU,
// f16x2 otherwise)
// from a split-weight handle (f16l6 / f16x2 / f16ax) the next and last rung is the plain fp32 forward (f32): slow, and right
int parity_fallback_precision(const uitk_model* m) {
    const int p = m->cfg.precision;
    if (p == UITK_PREC_F16L6 || p == UITK_PREC_F16X2 || p == UITK_PREC_F16AX) return UITK_PREC_F32;
    if (m->cfg.pooling == 2) return p == UITK_PREC_F16A ? UITK_PREC_F16AX : UITK_PREC_F16A;
    return (m->cfg.attention_full || m->cfg.act_gelu) ? UITK_PREC_F16X2 : UITK_PREC_F16L6;
}
// the precisions of the parity ladder (held to the budget); bf16 / fp8 / fp8w are measured modes
bool parity_precision(int p) { return p == UITK_PREC_F16 || p == UITK_PREC_F16A || p == UITK_PREC_F16L6 || p == UITK_PREC_F16X2 || p == UITK_PREC_F16AX; }
}  // namespace

// The acceptance check of uitk_calibrate_weights: both candidate images -- rounded to nearest (uitk_create's) and calibrated -- run the
// PRODUCT kernels on the held-out clips; the calibrated image is installed only if its largest deviation is at most calib_accept_pct per
// cent of the nearest image's (default 105: 5 % slack for ties).  report: max |d| nearest, max |d| calibrated, rms nearest, rms
// calibrated, accepted.  Leaves the chosen image on the device.  Returns UITK_OK with report[4] = -1 when the check does not apply.
static int calibration_acceptance(uitk_model* m, const uitk_weights* w, NaiveNet* nn, const std::vector<char>& img_cal, hipStream_t s, float* report) {
    for (int i = 0; i < 5; ++i) report[i] = -1.f;
    int nts[2];
    const int nlegs = heldout_legs(m, nts);
    double mx[2] = {0.0, 0.0}, msq[2] = {0.0, 0.0}, rms[2] = {0.0, 0.0};
    int legs_done = 0;
    for (int leg = 0; leg < nlegs; ++leg) {
        HeldOut ho;
        if (int rc = heldout_prepare(m, w, nn, s, &ho, nts[leg])) return rc;
        if (!ho.Te) { if (leg == 0) return UITK_OK; else continue; }   // (no full clip: the check does not apply; no short one: full clips only)
        for (int which = 0; which < 2; ++which) {                // 0: nearest, 1: calibrated
            const std::vector<char>& img = which ? img_cal : m->img_nearest;
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipMemcpy(m->image, img.data(), img.size(), hipMemcpyHostToDevice));
            double lmx = 0.0, lrms = 0.0;
            if (int rc = heldout_score(m, &ho, s, &lmx, &lrms)) return rc;
            heldout_merge(&mx[which], &msq[which], legs_done, lmx, lrms);
        }
        ++legs_done;
    }
    for (int which = 0; which < 2; ++which) rms[which] = std::sqrt(msq[which] / std::max(legs_done, 1));
    // (a nearest image that answers NaN on a held-out clip loses to any finite calibrated one; two NaN images: the nearest one stays and
    //  the budget check below sends the handle up the ladder)
    const bool accept = std::isfinite(mx[1]) && (!std::isfinite(mx[0]) || mx[1] * 100.0 <= (double)m->calib_accept_pct * mx[0]);
    {                                                            // leave the chosen image on the device
        const std::vector<char>& keep = accept ? img_cal : m->img_nearest;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(m->image, keep.data(), keep.size(), hipMemcpyHostToDevice));
    }
    report[0] = (float)mx[0]; report[1] = (float)mx[1];
    report[2] = (float)rms[0]; report[3] = (float)rms[1];
    report[4] = accept ? 1.f : 0.f;
    return UITK_OK;
}

int uitk_self_check(uitk_model* m, const uitk_weights* w, void* stream, float* out2) {
    if (!m || !w || !out2) return fail(UITK_ERR_INVALID_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return fail(UITK_ERR_INVALID_ARG, "uitk_self_check synchronises the device: not under stream capture");
    }
    out2[0] = out2[1] = -1.f;
    NaiveNet nn;
    if (int rc = naive_net_upload(m, w, &nn, s)) return rc;
    int nts[2];
    const int nlegs = heldout_legs(m, nts);
    double mx = 0.0, msq = 0.0;
    int rc = UITK_OK, legs_done = 0;
    for (int leg = 0; leg < nlegs && rc == UITK_OK; ++leg) {
        HeldOut ho;
        rc = heldout_prepare(m, w, &nn, s, &ho, nts[leg]);
        if (rc != UITK_OK || !ho.Te) { if (leg == 0) break; else continue; }
        double lmx = 0.0, lrms = 0.0;
        rc = heldout_score(m, &ho, s, &lmx, &lrms);
        if (rc == UITK_OK) { heldout_merge(&mx, &msq, legs_done, lmx, lrms); ++legs_done; }
    }
    m->fe_B = 0; m->fe_T = 0;                                    // the workspace held the check's operands, not the caller's
    if (rc != UITK_OK) return rc;
    if (legs_done) { out2[0] = (float)mx; out2[1] = (float)std::sqrt(msq / legs_done); }
    return UITK_OK;
}

int uitk_calibrate_weights(uitk_model* m, const uitk_weights* w, const void* pcm, int in_dtype, int64_t B, int64_t T, void* stream) {
    if (!m || !w) return fail(UITK_ERR_INVALID_ARG, "null argument");
    if (!calibratable_precision(m->cfg.precision))
        return fail(UITK_ERR_UNSUPPORTED, m->cfg.precision == UITK_PREC_F32 ? "precision f32 has nothing to calibrate (nothing in it is rounded to 16 bits)" : "unknown precision");
    hipStream_t s = (hipStream_t)stream;
    {                                                            // a pack-time operation (allocates, synchronises): never inside a capture
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return fail(UITK_ERR_INVALID_ARG, "uitk_calibrate_weights synchronises the device: not under stream capture (call it -- or model.prepare() -- before capturing)");
        }
    }
    DevBuf own_pcm;
    if (!pcm) {                                                  // the library's own signals: full-length clips for this front-end
        B = 32; in_dtype = UITK_IN_I16;
        T = std::max<int64_t>(100 * (int64_t)m->cfg.hop_size, (int64_t)(m->cfg.target_length / 16 * 16) * m->cfg.hop_size + m->cfg.n_fft);
        std::vector<int16_t> host;
        builtin_calibration_set((int)B, T, &host);
        HIP_TRY(hipMalloc(&own_pcm.p, host.size() * 2));
        HIP_TRY(hipMemcpy(own_pcm.p, host.data(), host.size() * 2, hipMemcpyHostToDevice));
        pcm = own_pcm.p;
    }
    if (B < 1 || T < 1) return fail(UITK_ERR_INVALID_ARG, "B and T must be positive");
    {   // at most 1024 calibration ROWS (clip-crops): more do not change a second moment that 768+ tokens already pin, and the fp32 work
        // buffer costs ~140 KB per row (long multi-crop clips used to be capped by clips only)
        const int64_t nc = m ? std::max(1, uitk_num_crops(m, T)) : 1;
        const int64_t maxB = std::max<int64_t>(1, 1024 / nc);
        if (B > maxB) B = maxB;
    }
    if (int rc = check_shape(m, B, T)) return rc;
    if (crop_nt(m, T) != GT) return fail(UITK_ERR_TOO_SHORT, "calibration clips must cover the full target_length (every time patch valid)");
    if (int rc = uitk_frontend(m, pcm, in_dtype, B, T, stream)) return rc;
    const int rows = (int)(B * uitk_num_crops(m, T));
    NaiveNet nn;
    if (int rc = naive_net_upload(m, w, &nn, s)) return rc;
    const CalibOffsets co = calib_offsets(&m->cfg);
    const size_t n_stats = calib_stats_doubles(&m->cfg);
    if (int rc = naive_net_bind(m, &nn, rows, n_stats)) return rc;
    CalibArgs& a = nn.a;
    a.off_pe = co.pe; a.off_ln1 = co.ln1; a.off_att = co.att; a.off_ln2 = co.ln2; a.off_hid = co.hid; a.per_block = co.per_block; a.off_mu1 = co.mu1;
    a.centre_ln1 = operand_centring(m->cfg.attention_full, m->cfg.act_gelu, m->cfg.pooling, kernel_precision(m->cfg.precision)) ? 1 : 0;
    HIP_TRY(launch_calibration(a, s));
    std::vector<float> stats_f(n_stats);
    HIP_TRY(hipMemcpyAsync(stats_f.data(), a.stats, n_stats * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    m->calib_stats.assign(stats_f.begin(), stats_f.end());
    for (double v : m->calib_stats) if (!std::isfinite(v)) { m->calib_stats.clear(); return fail(UITK_ERR_HIP, "calibration statistics are not finite"); }
    // re-pack with the calibrated rounding and replace the image (same layout, same size); pack-time operation: device-synchronous
    std::vector<char> img;
    ImageHeader h;
    if (int rc = pack_image(&m->cfg, w, m->calib_stats.data(), &img, &h)) return fail(rc, "pack failed");
    if (img.size() != m->hdr.total_bytes || img.size() != m->img_nearest.size())
        return fail(UITK_ERR_INVALID_ARG, "calibrated image has a different layout (other weights than at uitk_create?)");
    m->fe_B = 0; m->fe_T = 0;
    m->parity_escalate_to = -1;
    if (m->calib_accept_pct > 0) {
        const int rc = calibration_acceptance(m, w, &nn, img, s, m->calib_report);
        m->fe_B = 0; m->fe_T = 0;
        if (rc != UITK_OK) {                                     // whatever happened, the handle must not keep a half-decided image
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(m->image, m->img_nearest.data(), m->img_nearest.size(), hipMemcpyHostToDevice);
            m->weights_calibrated = 0;
            if (parity_precision(m->cfg.precision)) m->parity_escalate_to = parity_fallback_precision(m);
            return rc;
        }
        if (m->calib_report[4] >= 0.f) {                         // the check applied and left its choice on the device
            m->weights_calibrated = m->calib_report[4] > 0.f;
            // The ABSOLUTE part of the guarantee (round 5): the image that stays must itself be inside the budget on the held-out clips.
            // A refused calibration leaves the nearest-rounded image (measured at up to 1.2e-3 on trained-like checkpoints; harmless at init
            // scale): whichever image stays, over budget it is reported as "rebuild with the family's split-weight precision".
            const double dev = m->weights_calibrated ? m->calib_report[1] : m->calib_report[0];
            if (parity_precision(m->cfg.precision) && m->calib_budget_e6 > 0 && !(dev <= parity_budget(m)))       // (bf16 / fp8 / fp8w are
                m->parity_escalate_to = parity_fallback_precision(m);                                                  // measured modes, not parity modes)
            return UITK_OK;
        }
    } else {
        for (float& v : m->calib_report) v = -1.f;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(m->image, img.data(), img.size(), hipMemcpyHostToDevice));
    m->weights_calibrated = 1;
    return UITK_OK;
}

int uitk_get_calib_check(const uitk_model* m, float* out5) {
    if (!m || !out5) return fail(UITK_ERR_INVALID_ARG, "null argument");
    std::memcpy(out5, m->calib_report, sizeof m->calib_report);
    return UITK_OK;
}

int uitk_get_calib_stats(const uitk_model* m, double* out, size_t cap_doubles) {
    if (!m || !out) return fail(UITK_ERR_INVALID_ARG, "null argument");
    if (m->calib_stats.empty()) return fail(UITK_ERR_INVALID_ARG, "no uitk_calibrate_weights call on this handle");
    if (cap_doubles < m->calib_stats.size()) return fail(UITK_ERR_INVALID_ARG, "output buffer too small");
    std::memcpy(out, m->calib_stats.data(), m->calib_stats.size() * sizeof(double));
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
