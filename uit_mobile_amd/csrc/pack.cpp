// pack.cpp -- host-only: turn reference-layout fp32 parameters into the kernels' packed image.
//
// No HIP calls here (uitk_pack_weights works without a GPU; the layout tests drive it on CPU).
// Folding done at pack time (exact algebra, computed in double, rounded once to the operand type):
//   * init_bn (eval BatchNorm2d, models/uit.py:310-313)      -> per-mel-bin scale/shift applied before conversion
//   * conv bias + time_pos_embed + freq_pos_embed (:380-383)  -> bias table [24][128] = accumulator init
//   * LayerNorm affine of norm1 / norm2 (:223, 235)           -> folded into qkv / fc1 weights and biases
//   * attention scale (dim//heads)^-0.5 = 0.125 (:99-100,115), times log2(e) (softmax computed with exp2)
//                                                             -> folded into the q rows of qkv
//   * residual-stream centring: every consumer of the residual stream is a LayerNorm (norm1, norm2, final norm), which is
//     invariant to adding c*1 to a token.  So the channel-mean of everything that is ADDED to the stream is removed here:
//     column means of the patch-embed / proj / fc2 weights over their 128 output channels, the means of the bias table,
//     proj bias and fc2 bias.  The kernel's residual stream then has (up to rounding) zero mean per token, and its one-pass
//     variance E[x^2] - mean^2 cannot cancel however large the common offset of the checkpoint's stream is.
// Fragment order: see uitk_dev.h and encoder.hip.  A "fragment" is the 1 KiB A/B operand of one
// mfma_f32_16x16x32: lane l (c = l&15, g = l>>4), element j  <->  W[16*ft + c][ 32*s + 16*(j>>2) + 4*g + (j&3) ]
// (the k-order a C/D tile pair has after in-register conversion, so activations never move between lanes).
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <functional>
#include <numeric>
#include <thread>
#include <vector>

#include "../../include/uitk.h"
#include "pack.h"
#include "uitk_dev.h"

namespace uitk {

static uint16_t f32_to_f16(float f) {          // round-to-nearest-even, subnormals, inf/nan
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);          // overflow -> inf (after rounding)
    if (x < 0x33000001u) return (uint16_t)sign;                        // underflow -> 0 (< half the smallest subnormal)
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t base;
    if (e < -14) { shift = 13 + (-14 - e); base = 0; }                 // subnormal
    else { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x7fffffu; }
    uint32_t r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) ++r;
    return (uint16_t)(sign | (base + r));
}

static uint16_t f32_to_bf16(float f) {
    uint32_t x; std::memcpy(&x, &f, 4);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40u);
    x += 0x7fffu + ((x >> 16) & 1u);
    return (uint16_t)(x >> 16);
}

// 16-bit operand of the f16-class paths; precision fp8 keeps every 16-bit operand (patch embed, proj, head) in f16
static inline uint16_t cvt(double v, int prec) { return prec == UITK_PREC_BF16 ? f32_to_bf16((float)v) : f32_to_f16((float)v); }

static double op16_value(uint16_t u, int prec) {
    if (prec == UITK_PREC_BF16) { uint32_t x = (uint32_t)u << 16; float f; std::memcpy(&f, &x, 4); return f; }
    const uint32_t sign = (u >> 15) & 1u, e = (u >> 10) & 31u, m = u & 1023u;
    const double v = e == 0 ? std::ldexp((double)m, -24) : std::ldexp((double)(m | 1024u), (int)e - 25);
    return sign ? -v : v;
}
// the representable neighbour of u in the direction of larger (up) or smaller values (sign-magnitude encodings)
static uint16_t op16_step(uint16_t u, bool up) {
    const bool neg = (u & 0x8000u) != 0;
    if ((u & 0x7fffu) == 0) return up ? 0x0001u : 0x8001u;
    return (uint16_t)((neg != up) ? u + 1 : u - 1);
}

// STRUCTURED ROUNDING of a weight matrix to the 16-bit operand type.  A weight's rounding error is the same
// for every token of a clip, so its effect survives the mean-pool -- and it is largest along the input directions that carry most
// of the energy.  For two linears those directions are known without any data: a log-mel patch is dominated by its per-mel-row
// means (stationary spectra), and ReLU outputs are non-negative (their common level).  So: round to nearest, then inside every
// column group flip the cheapest near-tie roundings until the group's SUMMED rounding error is below half its smallest step --
// the error then cannot see an input that is constant over a group.  The flipped elements were near ties: the per-element rms
// error grows by ~4 %, the patch embed's weight-induced output error drops 5-9x (2x on speech) on the reference's sample clips.
static std::vector<uint16_t> round_structured(int rows, int K, const std::function<double(int, int)>& W, int prec, int group) {
    std::vector<uint16_t> q((size_t)rows * K);
    std::vector<double> w(K), dq(K), dalt(K);
    std::vector<uint16_t> alt(K);
    std::vector<char> used(K);
    for (int r = 0; r < rows; ++r) {
        for (int k = 0; k < K; ++k) {
            w[k] = W(r, k);
            const uint16_t u = cvt(w[k], prec);
            q[(size_t)r * K + k] = u;
            dq[k] = op16_value(u, prec) - w[k];
            alt[k] = dq[k] == 0.0 ? u : op16_step(u, dq[k] < 0.0);          // the neighbour on the far side of w
            dalt[k] = op16_value(alt[k], prec) - w[k];
            used[k] = 0;
        }
        for (int g0 = 0; g0 < K; g0 += group) {
            const int g1 = g0 + group < K ? g0 + group : K;
            double sum = 0.0;
            for (int k = g0; k < g1; ++k) sum += dq[k];
            for (int it = 0; it < group; ++it) {
                int best = -1; double best_cost = 0.0;
                for (int k = g0; k < g1; ++k) {
                    if (used[k] || alt[k] == q[(size_t)r * K + k] || !std::isfinite(dalt[k])) continue;
                    const double ns = sum - dq[k] + dalt[k];
                    if (std::fabs(ns) >= std::fabs(sum)) continue;
                    const double cost = dalt[k] * dalt[k] - dq[k] * dq[k];
                    if (best < 0 || cost < best_cost) { best = k; best_cost = cost; }
                }
                if (best < 0) break;
                sum += dalt[best] - dq[best];
                q[(size_t)r * K + best] = alt[best]; dq[best] = dalt[best]; used[best] = 1;
            }
        }
    }
    return q;
}

// CALIBRATED ROUNDING (round 4): every row of a weight matrix is rounded to the 16-bit operand type one column at a
// time, and the rounding error of column j is pushed onto the columns not yet rounded so that (W - Q) x stays small for the inputs x
// the linear actually sees -- H = E[x x^T] (uncentred: the mean direction counts) from a calibration batch.  This is GPTQ (Frantar
// et al. 2022: greedy optimal-brain-quantisation through the Cholesky factor of H^-1, columns in order of decreasing H_jj) on the
// fp16 grid instead of an integer grid.  Costs nothing at run time: the kernel multiplies the same single fp16 weights.  The damping
// (1 % of the mean diagonal) bounds what the rounding may do along directions the calibration batch never excited.
// tools/calib_study.py: worst max|dlogit| over 12 trained-like XS checkpoints 1.4e-3 (nearest) -> 4.1e-4 = the floor set by the fp16
// ACTIVATIONS (split weights, f16x2: 3.6e-4), calibrated on synthetic signals and scored on held-out recordings.
static bool chol_lower(std::vector<double>& A, int n) {          // in place, lower triangle; false if not positive definite
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double v = A[(size_t)i * n + j];
            const double* ai = &A[(size_t)i * n];
            const double* aj = &A[(size_t)j * n];
            for (int k = 0; k < j; ++k) v -= ai[k] * aj[k];
            A[(size_t)i * n + j] = v / d;
        }
    }
    return true;
}

// the column order (decreasing H_jj) and the upper-triangular factor U with (H_perm + damping)^-1 = U^T U; false = statistics unusable
static bool gptq_factor(int K, const double* H, double damp, std::vector<int>* perm_out, std::vector<double>* U_out) {
    std::vector<int>& perm = *perm_out;
    perm.resize(K);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return H[(size_t)a * K + a] > H[(size_t)b * K + b]; });
    std::vector<double> A((size_t)K * K);
    double tr = 0.0;
    for (int i = 0; i < K; ++i) {
        for (int j = 0; j < K; ++j) A[(size_t)i * K + j] = H[(size_t)perm[i] * K + perm[j]];
        if (!(A[(size_t)i * K + i] > 0.0)) A[(size_t)i * K + i] = 1.0;                      // an input that never fired
        tr += A[(size_t)i * K + i];
    }
    bool ok = std::isfinite(tr);
    for (int i = 0; i < K; ++i) A[(size_t)i * K + i] += damp * tr / K;
    std::vector<double>& U = *U_out;
    U.assign((size_t)K * K, 0.0);                                // upper triangular, (H + damping)^-1 = U^T U
    if (ok) ok = chol_lower(A, K);
    if (ok) {
        // Linv (lower) by forward substitution, column by column; Hinv = Linv^T Linv
        std::vector<double> Li((size_t)K * K, 0.0);
        for (int c = 0; c < K; ++c) {
            Li[(size_t)c * K + c] = 1.0 / A[(size_t)c * K + c];
            for (int i = c + 1; i < K; ++i) {
                double v = 0.0;
                for (int k = c; k < i; ++k) v -= A[(size_t)i * K + k] * Li[(size_t)k * K + c];
                Li[(size_t)i * K + c] = v / A[(size_t)i * K + i];
            }
        }
        std::vector<double> Hi((size_t)K * K, 0.0);
        for (int k = 0; k < K; ++k) {                            // Hi += Li[k][:]^T Li[k][:]  (row k of Li has entries 0..k)
            const double* r = &Li[(size_t)k * K];
            for (int i = 0; i <= k; ++i) {
                const double ri = r[i];
                if (ri == 0.0) continue;
                double* hi = &Hi[(size_t)i * K];
                for (int j = 0; j <= i; ++j) hi[j] += ri * r[j];
            }
        }
        for (int i = 0; i < K; ++i) for (int j = i + 1; j < K; ++j) Hi[(size_t)i * K + j] = Hi[(size_t)j * K + i];
        ok = chol_lower(Hi, K);
        if (ok) for (int i = 0; i < K; ++i) for (int j = 0; j <= i; ++j) U[(size_t)j * K + i] = Hi[(size_t)i * K + j];
    }
    return ok;
}

// (the error feedback may move a later column by MORE than half a step from its own nearest value -- that is the point: what bounds the
// deviation from the exact weights is the damping, not a per-element half-step rule)
static std::vector<uint16_t> round_gptq(int rows, int K, const std::function<double(int, int)>& W, int prec, const double* H, double damp = 0.01) {
    std::vector<uint16_t> q((size_t)rows * K);
    std::vector<int> perm;
    std::vector<double> U;
    const bool ok = gptq_factor(K, H, damp, &perm, &U);
    std::vector<double> wr(K);
    for (int r = 0; r < rows; ++r) {
        for (int k = 0; k < K; ++k) wr[k] = W(r, perm[k]);
        for (int j = 0; j < K; ++j) {
            const uint16_t u = cvt(wr[j], prec);
            q[(size_t)r * K + perm[j]] = u;
            if (!ok) continue;                                   // (statistics unusable: plain round-to-nearest)
            const double v = op16_value(u, prec);
            if (!std::isfinite(v)) continue;
            const double e = (wr[j] - v) / U[(size_t)j * K + j];
            const double* uj = &U[(size_t)j * K];
            for (int k = j + 1; k < K; ++k) wr[k] -= e * uj[k];
        }
    }
    return q;
}

// OCP e4m3fn: 1-4-3, bias 7, max 448 (0x7e), no infinity, subnormal step 2^-9; round-to-nearest-even, saturating
static uint8_t f32_to_e4m3(double x) {
    const uint8_t s = std::signbit(x) ? 0x80 : 0;
    const double a = std::fabs(x);
    if (!(a == a)) return 0x7f;
    if (a >= 464.0) return (uint8_t)(s | 0x7e);                       // beyond the midpoint above 448: saturate
    if (a <= std::ldexp(1.0, -10)) return s;                           // at most half the smallest subnormal -> 0 (ties to even)
    int e;
    (void)std::frexp(a, &e);
    int E = e - 1;                                                     // exponent of the leading bit
    if (E < -6) E = -6;                                                // subnormal range
    double r = std::nearbyint(a / std::ldexp(1.0, E - 3));             // units of the last mantissa bit, ties to even
    if (r >= 16.0) { r = 8.0; E += 1; }
    if (E > 8 || (E == 8 && r > 14.0)) return (uint8_t)(s | 0x7e);
    if (r < 8.0) return (uint8_t)(s | (uint8_t)r);                     // subnormal (exponent field 0)
    return (uint8_t)(s | ((E + 7) << 3) | ((int)r - 8));
}

static double e4m3_value(uint8_t u) {
    const int e = (u >> 3) & 15, m = u & 7;
    const double v = e == 0 ? std::ldexp((double)m, -9) : std::ldexp((double)(m | 8), e - 10);
    return (u & 0x80) ? -v : v;
}

// Weights on the MX e4m3 grid (round 5): every row is cut into 32-element scale blocks (block_of(k): the hardware's interleaved blocks
// of a fragment8 for precision fp8, contiguous ones for the weights-only variant); a block's scale is the smallest power of two that
// brings its largest |w| inside e4m3's 448 (as pack_frag8 chooses it), its elements are e4m3 values times that scale.  Returns the
// VALUES (doubles, exactly representable).  H = nullptr: every element to nearest.  With the operand's second moments H: GPTQ
// (round_gptq above) on this grid -- the rounding error of a column is pushed onto the columns not yet rounded, saturating at the
// block's 448 * scale; the scales stay those of the exact weights.
static std::vector<double> round_fp8_grid(int rows, int K, const std::function<double(int, int)>& W, const std::function<int(int)>& block_of,
                                          int nblocks, const double* H, double damp = 0.01) {
    std::vector<double> q((size_t)rows * K);
    std::vector<int> perm;
    std::vector<double> U;
    bool ok = false;
    if (H) ok = gptq_factor(K, H, damp, &perm, &U);
    if (!ok) { perm.resize(K); std::iota(perm.begin(), perm.end(), 0); }
    std::vector<double> wr(K), sc(nblocks);
    for (int r = 0; r < rows; ++r) {
        std::fill(sc.begin(), sc.end(), 0.0);
        for (int k = 0; k < K; ++k) sc[block_of(k)] = std::fmax(sc[block_of(k)], std::fabs(W(r, k)));
        for (int b = 0; b < nblocks; ++b) {
            const double m = sc[b];
            int e = 0;
            if (m > 0.0) {                                             // smallest e with m / 2^e <= 448 (pack_frag8's rule)
                e = (int)std::ceil(std::log2(m / 448.0));
                while (m / std::ldexp(1.0, e) > 448.0) ++e;
                while (m / std::ldexp(1.0, e - 1) <= 448.0) --e;
            }
            if (e < -126) e = -126;
            if (e > 126) e = 126;
            sc[b] = std::ldexp(1.0, e);
        }
        for (int k = 0; k < K; ++k) wr[k] = W(r, perm[k]);
        for (int j = 0; j < K; ++j) {
            const double s_ = sc[block_of(perm[j])];
            const double v = e4m3_value(f32_to_e4m3(wr[j] / s_)) * s_;
            q[(size_t)r * K + perm[j]] = v;
            if (!ok) continue;
            const double e = (wr[j] - v) / U[(size_t)j * K + j];
            const double* uj = &U[(size_t)j * K];
            for (int k = j + 1; k < K; ++k) wr[k] -= e * uj[k];
        }
    }
    return q;
}

// one 2 KiB fragment8 (see uitk_dev.h) of rows 16*tile .. +15, k = kbase + [0, 128) from an accessor W(row, k) -> double, plus the
// four MX scale bytes of every row: scales[lane = (row, g' = beta)] = E8M0 exponent byte of block beta
static void pack_frag8(uint8_t* dst, uint8_t* scales, const std::function<double(int, int)>& W, int tile, int kbase) {
    double v[64][32];
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) v[l][j] = W(16 * tile + (l & 15), kbase + 16 * (j >> 2) + 4 * (l >> 4) + (j & 3));
    int ex[16][4];
    for (int row = 0; row < 16; ++row)
        for (int beta = 0; beta < 4; ++beta) {
            double m = 0.0;
            for (int gg = 0; gg < 2; ++gg)
                for (int b = 0; b < 16; ++b) m = std::fmax(m, std::fabs(v[(2 * (beta & 1) + gg) * 16 + row][16 * (beta >> 1) + b]));
            int e = 0;
            if (m > 0.0) {                                             // smallest e with m / 2^e <= 448
                e = (int)std::ceil(std::log2(m / 448.0));
                while (m / std::ldexp(1.0, e) > 448.0) ++e;
                while (m / std::ldexp(1.0, e - 1) <= 448.0) --e;
            }
            if (e < -126) e = -126;
            if (e > 126) e = 126;
            ex[row][beta] = e;
            scales[beta * 16 + row] = (uint8_t)(127 + e);
        }
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            const int beta = 2 * (j >> 4) + (l >> 5);                  // hardware block of (lane group g = l>>4, byte j)
            const uint8_t q = f32_to_e4m3(v[l][j] / std::ldexp(1.0, ex[l & 15][beta]));
            dst[(j >> 4) * 1024 + l * 16 + (j & 15)] = q;              // two lane-linear 1 KiB halves
        }
}

// OCP fp6 e2m3: 1-2-3, bias 1, max 7.5, no inf/nan, subnormal step 1/8; round-to-nearest-even, saturating
static uint8_t f32_to_e2m3(double x) {
    const uint8_t s = std::signbit(x) ? 0x20 : 0;
    double a = std::fabs(x);
    if (!(a == a)) return 0;
    if (a >= 7.5) return (uint8_t)(s | 0x1f);
    int E = 0;                                                         // exponent of the leading bit, 0 for [0, 2)
    if (a >= 4.0) E = 2; else if (a >= 2.0) E = 1;
    double r = std::nearbyint(a / std::ldexp(1.0, E - 3));             // units of the last mantissa bit (ties to even)
    if (r >= 16.0) { r = 8.0; E += 1; }
    if (r < 8.0) return (uint8_t)(s | (uint8_t)r);                     // subnormal: exponent field 0, value r / 8
    return (uint8_t)(s | ((E + 1) << 3) | ((int)r - 8));
}

// one 1536-byte fragment6 (uitk_dev.h) of rows 16*tile .. +15, k = kbase + [0, 128) from an accessor W(row, k) -> double; ONE MX
// scale per row (the smallest power of two that brings the row's maximum inside e2m3's 7.5): scales[row] = E8M0 byte
static void pack_frag6(uint8_t* dst, uint8_t* scales, const std::function<double(int, int)>& W, int tile, int kbase) {
    double v[64][32];
    double mx[16] = {};
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 32; ++i) {
            v[l][i] = W(16 * tile + (l & 15), kbase + 32 * (i >> 3) + 16 * ((i & 7) >> 2) + 4 * (l >> 4) + (i & 3));
            mx[l & 15] = std::fmax(mx[l & 15], std::fabs(v[l][i]));
        }
    int ex[16];
    for (int row = 0; row < 16; ++row) {
        int e = -126;
        if (mx[row] > 0.0) {
            e = (int)std::ceil(std::log2(mx[row] / 7.5));
            while (mx[row] / std::ldexp(1.0, e) > 7.5) ++e;
            while (mx[row] / std::ldexp(1.0, e - 1) <= 7.5) --e;
        }
        if (e < -126) e = -126;
        if (e > 126) e = 126;
        ex[row] = e; scales[row] = (uint8_t)(127 + e);
    }
    std::memset(dst, 0, FRAG6_BYTES);
    for (int l = 0; l < 64; ++l) {
        uint32_t w[6] = {};
        for (int i = 0; i < 32; ++i) {
            const uint32_t q = f32_to_e2m3(v[l][i] / std::ldexp(1.0, ex[l & 15]));
            const int bit = 6 * i;
            w[bit >> 5] |= q << (bit & 31);
            if ((bit & 31) > 26) w[(bit >> 5) + 1] |= q >> (32 - (bit & 31));
        }
        std::memcpy(dst + l * 16, w, 16);                              // lane-linear 1 KiB part: dwords 0..3
        std::memcpy(dst + 1024 + l * 8, w + 4, 8);                     // lane-linear 512-byte part: dwords 4, 5
    }
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int validate_config(const uitk_config* c, const char** why) {
    auto bad = [&](const char* m) { *why = m; return UITK_ERR_UNSUPPORTED; };
    if (c->embed_dim != DIM) return bad("embed_dim must be 128 (uit_* family)");
    if (c->num_heads != 2) return bad("num_heads must be 2");
    if (c->mlp_hidden != HID) return bad("mlp hidden must be 384 (mlp_ratio 3.0)");
    if (c->n_mels != NMEL) return bad("n_mels must be 64");
    if (c->n_fft != 512) return bad("n_fft must be 512");
    if (c->win_size < 1 || c->win_size > c->n_fft) return bad("win_size must be 1..n_fft (the window is zero-padded to n_fft as torch.stft does)");
    if (c->hop_size < 1 || c->hop_size > 512) return bad("hop_size out of range");
    if (c->patch_size != 16) return bad("patch_size must be 16");
    if (c->target_length / 16 != GT) return bad("target_length must give 6 time patches (96..111)");
    if (c->center != 0 && c->center != 1) return bad("center must be 0 or 1");
    if (c->depth < 1 || c->depth > 64) return bad("depth out of range");
    if (c->outputdim < 1 || c->outputdim > 65536) return bad("outputdim out of range");
    if (c->eval_avg != UITK_AVG_MEAN && c->eval_avg != UITK_AVG_MAX) return bad("unknown eval_avg");
    if (c->precision != UITK_PREC_F16 && c->precision != UITK_PREC_BF16 && c->precision != UITK_PREC_FP8 && c->precision != UITK_PREC_F16X2 && c->precision != UITK_PREC_F16L6 && c->precision != UITK_PREC_FP8W && c->precision != UITK_PREC_F16A && c->precision != UITK_PREC_F16AX) return bad("unknown precision");
    if ((c->precision == UITK_PREC_F16A || c->precision == UITK_PREC_F16AX) && c->pooling != 2) return bad("precisions f16a / f16ax (split activation operands of the cls tile) are built for pooling 'token'");

    if (c->precision == UITK_PREC_F16L6 && (c->attention_full || c->act_gelu)) return bad("precision f16l6 is built for BNeckAttention + ReLU (any pooling)");
    if ((c->attention_full != 0 && c->attention_full != 1) || (c->act_gelu != 0 && c->act_gelu != 1)) return bad("attention_full / act_gelu must be 0 or 1");
    if ((c->attention_full || c->act_gelu || c->pooling) && c->precision == UITK_PREC_FP8) return bad("precision fp8 is built for the uit_* family (BNeckAttention, ReLU, mean pooling) only");
    if (c->pooling < 0 || c->pooling > 2) return bad("pooling must be 0 (mean), 1 (dm) or 2 (token)");
    if (c->pooling && c->precision != UITK_PREC_F16 && c->precision != UITK_PREC_F16L6 && c->precision != UITK_PREC_F16X2 && c->precision != UITK_PREC_FP8W && c->precision != UITK_PREC_F16A && c->precision != UITK_PREC_F16AX) return bad("pooling 'dm' / 'token' are built for precision f16 / f16l6 / f16x2");
    if (c->pooling == 1 && c->outputdim > 640) return bad("pooling 'dm' is built for outputdim <= 640");
    return UITK_OK;
}

// one 1 KiB fragment from an accessor W(row, k) -> double
static void pack_frag(uint16_t* dst, int prec, const std::function<double(int, int)>& W, int ft, int s) {
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
            const int row = 16 * ft + (l & 15);
            const int k = 32 * s + 16 * (j >> 2) + 4 * (l >> 4) + (j & 3);
            dst[l * 8 + j] = cvt(W(row, k), prec);
        }
}

CalibOffsets calib_offsets(const uitk_config* c) {
    const size_t inner = c->attention_full ? DIM : INNER;
    CalibOffsets o;
    o.pe = 0; o.ln1 = 256 * 256; o.att = o.ln1 + DIM * DIM; o.ln2 = o.att + inner * inner; o.hid = o.ln2 + DIM * DIM;
    o.per_block = 2 * DIM * DIM + inner * inner + (size_t)HID * HID;
    o.mu1 = 256 * 256 + (size_t)c->depth * o.per_block;
    return o;
}
size_t calib_stats_doubles(const uitk_config* c) { return calib_offsets(c).mu1 + (size_t)c->depth * DIM; }

void fold_model(const uitk_config* c, const uitk_weights* w, Folded* F) {
    const int depth = c->depth, inner = c->attention_full ? DIM : INNER;
    F->depth = depth; F->inner = inner;
    // patch embed: Conv2d weight [oc][0][fi][ti] as [oc][k = fi*16 + ti]; residual centring: mean over the 128 output channels removed
    F->pe_w.assign((size_t)DIM * 256, 0.0);
    for (int k = 0; k < 256; ++k) {
        double cm = 0.0;
        for (int oc = 0; oc < DIM; ++oc) cm += (double)w->pe_w[oc * 256 + k];
        cm /= DIM;
        for (int oc = 0; oc < DIM; ++oc) F->pe_w[(size_t)oc * 256 + k] = (double)w->pe_w[oc * 256 + k] - cm;
    }
    // conv bias + pos-embeds per token (uit.py:380-383), centred; row 24: the cls token of pooling 'token' (uit.py:389-391)
    F->pe_bias.assign((size_t)(NTOK + 1) * DIM, 0.0);
    const int gt_model = c->target_length / 16;
    for (int pf = 0; pf < GF; ++pf)
        for (int pt = 0; pt < GT; ++pt) {
            double row[DIM], mean = 0.0;
            for (int oc = 0; oc < DIM; ++oc) {
                row[oc] = (double)w->pe_b[oc] + (double)w->time_pos[oc * gt_model + pt] + (double)w->freq_pos[oc * GF + pf];
                mean += row[oc];
            }
            mean /= DIM;                                                // residual centring (see the file header)
            for (int oc = 0; oc < DIM; ++oc) F->pe_bias[(size_t)(pf * GT + pt) * DIM + oc] = row[oc] - mean;
        }
    {
        double row[DIM], mean = 0.0;
        for (int oc = 0; oc < DIM; ++oc) {
            row[oc] = (w->cls_token ? (double)w->cls_token[oc] : 0.0) + (w->token_pos ? (double)w->token_pos[oc] : 0.0);
            mean += row[oc];
        }
        mean /= DIM;
        for (int oc = 0; oc < DIM; ++oc) F->pe_bias[(size_t)NTOK * DIM + oc] = row[oc] - mean;
    }
    F->blocks.assign(depth, FoldedBlock());
    // head_dim = dim // num_heads (uit.py:99-100); log2(e) rides along so the kernel's softmax is exp2
    const double scale = 1.0 / std::sqrt((double)(DIM / 2)) * 1.4426950408889634074;
    for (int b = 0; b < depth; ++b) {
        const uitk_block_weights& bw = w->blocks[b];
        FoldedBlock& fb = F->blocks[b];
        fb.qkv_w.resize((size_t)3 * inner * DIM); fb.qkv_b.resize(3 * inner);
        for (int r = 0; r < 3 * inner; ++r) {
            const double sc = r < inner ? scale : 1.0;
            double acc = bw.qkv_b[r];
            for (int k = 0; k < DIM; ++k) {
                fb.qkv_w[(size_t)r * DIM + k] = (double)bw.qkv_w[r * DIM + k] * (double)bw.norm1_w[k] * sc;
                acc += (double)bw.qkv_w[r * DIM + k] * (double)bw.norm1_b[k];
            }
            fb.qkv_b[r] = acc * sc;
        }
        fb.proj_w.resize((size_t)DIM * inner); fb.proj_b.resize(DIM);
        for (int k = 0; k < inner; ++k) {
            double cm = 0.0;
            for (int r = 0; r < DIM; ++r) cm += (double)bw.proj_w[r * inner + k];
            cm /= DIM;
            for (int r = 0; r < DIM; ++r) fb.proj_w[(size_t)r * inner + k] = (double)bw.proj_w[r * inner + k] - cm;
        }
        fb.fc2_w.resize((size_t)DIM * HID); fb.fc2_b.resize(DIM);
        for (int k = 0; k < HID; ++k) {
            double cm = 0.0;
            for (int r = 0; r < DIM; ++r) cm += (double)bw.fc2_w[r * HID + k];
            cm /= DIM;
            for (int r = 0; r < DIM; ++r) fb.fc2_w[(size_t)r * HID + k] = (double)bw.fc2_w[r * HID + k] - cm;
        }
        double proj_bm = 0.0, fc2_bm = 0.0;
        for (int r = 0; r < DIM; ++r) { proj_bm += (double)bw.proj_b[r]; fc2_bm += (double)bw.fc2_b[r]; }
        proj_bm /= DIM; fc2_bm /= DIM;
        for (int r = 0; r < DIM; ++r) { fb.proj_b[r] = (double)bw.proj_b[r] - proj_bm; fb.fc2_b[r] = (double)bw.fc2_b[r] - fc2_bm; }
        fb.fc1_w.resize((size_t)HID * DIM); fb.fc1_b.resize(HID);
        for (int r = 0; r < HID; ++r) {
            double acc = bw.fc1_b[r];
            for (int k = 0; k < DIM; ++k) {
                fb.fc1_w[(size_t)r * DIM + k] = (double)bw.fc1_w[r * DIM + k] * (double)bw.norm2_w[k];
                acc += (double)bw.fc1_w[r * DIM + k] * (double)bw.norm2_b[k];
            }
            fb.fc1_b[r] = acc;
        }
    }
}

int pack_image(const uitk_config* c, const uitk_weights* w, const double* calib, std::vector<char>* out, ImageHeader* hdr_out) {
    const bool l6 = c->precision == UITK_PREC_F16L6;                     // f16 operands + fp6 low-order weight term
    const bool w8 = c->precision == UITK_PREC_FP8W;                      // f16 operands (the f16 kernel) holding weights on the MX e4m3 grid
    // (f16a: the image of precision f16 -- its second activation terms are formed in the kernel)
    const int prec = (c->precision == UITK_PREC_F16X2 || l6 || w8 || c->precision == UITK_PREC_F16A || c->precision == UITK_PREC_F16AX) ? UITK_PREC_F16 : c->precision, depth = c->depth, C = c->outputdim;
    const int ntiles = (C + 15) / 16;
    const int nb = c->n_fft / 2 + 1;
    // ---- mel filterbank as banded B operands of v_mfma_f32_4x4x1 (16 independent 4x4 blocks): block g owns mels
    // 4g..4g+3 and walks its own bin window [klo_g, klo_g + S), S = the widest group's band (HTK, 64 mels: 49 bins)
    int grp_klo[16], mel_S = 4;
    {
        int first[16], last[16];
        for (int g = 0; g < 16; ++g) {
            first[g] = -1; last[g] = -1;
            for (int k = 0; k < nb; ++k)
                for (int m = 4 * g; m < 4 * g + 4; ++m)
                    if (w->mel_fb[k * NMEL + m] != 0.0f) { if (first[g] < 0) first[g] = k; last[g] = k; }
        }
        // A window starts on a multiple of 4 bins and must end inside the power row (272 = 4 * 68 columns): a group whose band
        // starts high is clamped DOWN, which widens the span it has to walk -- so the span is measured from the clamped start
        // and the clamp repeated until the walk length is stable (it only grows, bounded by the row).
        int walk = MEL_BMAX;                                         // steps the kernel actually reads
        for (int iter = 0; iter < 64; ++iter) {
            int S = 4;
            for (int g = 0; g < 16; ++g) {
                int klo = first[g] < 0 ? 0 : (first[g] & ~3);
                if (klo > MEL_PROW - walk) klo = MEL_PROW - walk;
                if (klo < 0) klo = 0;
                grp_klo[g] = klo;
                const int span = first[g] < 0 ? 0 : last[g] - klo + 1;
                if (span > S) S = (span + 3) & ~3;
            }
            mel_S = S;
            const int nw = S > MEL_BMAX ? S : MEL_BMAX;
            if (nw == walk) break;
            walk = nw;
        }
        // LDS bank spreading (fast path only): a lane's operand read starts at bank (48 x + klo_g + 4 c) mod 64 (x = its frame row),
        // so two mel groups served in the same 16-lane slice of a ds_read_b128 -- {0,3,5,6}, {1,2,4,7}, {8,11,13,14}, {9,10,12,15}
        // -- collide when their window starts agree mod 16.  A window may start up to 12 bins early as long as its band still
        // ends inside the MEL_BMAX steps: pick the cheapest shifts that make the four residues distinct.
        bool bank_spread = true;
#ifdef UITK_LAB
        bank_spread = std::getenv("UITK_PACK_NO_BANK_SPREAD") == nullptr;      // lab A/B of the LDS bank spreading
#endif
        if (mel_S <= MEL_BMAX && bank_spread) {
            static const int svc[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};
            for (const auto& sg : svc) {
                int best = -1, best_cost = 1 << 30;
                for (int code = 0; code < 256; ++code) {
                    int res[4], cost = 0; bool ok = true;
                    for (int i = 0; i < 4 && ok; ++i) {
                        const int g = sg[i], sh = 4 * ((code >> (2 * i)) & 3), k = grp_klo[g] - sh;
                        if (k < 0 || (first[g] >= 0 && last[g] - k + 1 > MEL_BMAX)) ok = false;
                        res[i] = ((k % 16) + 16) % 16; cost += sh;
                    }
                    for (int i = 0; i < 4 && ok; ++i)
                        for (int j = i + 1; j < 4; ++j) if (res[i] == res[j]) ok = false;
                    if (ok && cost < best_cost) { best = code; best_cost = cost; }
                }
                if (best >= 0)
                    for (int i = 0; i < 4; ++i) grp_klo[sg[i]] -= 4 * ((best >> (2 * i)) & 3);
            }
            int S = 4;
            for (int g = 0; g < 16; ++g) { const int span = first[g] < 0 ? 0 : last[g] - grp_klo[g] + 1; if (span > S) S = (span + 3) & ~3; }
            mel_S = S;
        }
    }
    ImageHeader h{};
    h.magic = IMAGE_MAGIC; h.precision = c->precision; h.depth = depth; h.outputdim = C; h.mel_steps = mel_S;
    size_t o = align_up(sizeof(ImageHeader), 256);
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    const bool full = c->attention_full != 0;
    const int inner = full ? DIM : INNER;                                // qkv = Linear(dim, 3 * inner), proj = Linear(inner, dim)
    const bool x2 = c->precision == UITK_PREC_F16X2 || c->precision == UITK_PREC_F16AX;      // split weights: operand type fp16 (prec 0 for every conversion); f16ax = this image, other kernel
    const int blk_stages = full ? (x2 ? BLOCK_STAGES_X2_ATT : BLOCK_STAGES_ATT) : (prec == UITK_PREC_FP8 ? BLOCK_STAGES_FP8 : x2 ? BLOCK_STAGES_X2 : BLOCK_STAGES);
    const int b_proj = full ? BIASF_PROJ : BIAS_PROJ, b_fc1 = full ? BIASF_FC1 : BIAS_FC1, b_fc2 = full ? BIASF_FC2 : BIAS_FC2;
    h.off_wstream = take(l6 ? (size_t)PE_STAGES * STAGE_BYTES + (size_t)depth * L6_BLOCK_BYTES + L6_TAIL_BYTES
                            : (size_t)(PE_STAGES + depth * blk_stages + TAIL_STAGES) * STAGE_BYTES);
    h.off_bias = take((size_t)depth * BIAS_FLOATS * 4);
    h.off_bn_sc = take(NMEL * 4); h.off_bn_sh = take(NMEL * 4);
    h.off_pe_bias = take((NTOK + 1) * DIM * 4);                       // + the cls row (pooling 'token')
    h.off_fin = take(4 * DIM * 4);
    h.off_head_w = take((size_t)ntiles * 4 * 2 * FRAG_BYTES);         // hi + lo term per fragment
    h.off_head_b = take((size_t)ntiles * 16 * 4);
    h.off_window = take(512 * 4);
    h.off_tw256 = take(256 * 8);
    h.off_tw512 = take(128 * 8);
    h.off_mel_b = take((size_t)(1 + mel_S) * 64 * 4);                 // [64 lane klo (int32)] + [S/4][64 lanes][4] floats
    h.total_bytes = o;
    out->assign(o, 0);
    char* img = out->data();
    std::memcpy(img, &h, sizeof h);

    // ---- front-end tables
    // win_size < n_fft (uit.py:290, 301-303): torch.stft zero-pads the window on both sides to n_fft, left = (n_fft - win_size) / 2
    std::memcpy(img + h.off_window + (size_t)((c->n_fft - c->win_size) / 2) * 4, w->window, (size_t)c->win_size * 4);
    float* tw256 = (float*)(img + h.off_tw256);
    for (int p = 0; p < 16; ++p)
        for (int k1 = 0; k1 < 16; ++k1) {
            const double a = 2.0 * M_PI * (double)(p * k1) / 256.0;
            tw256[2 * (p * 16 + k1)] = (float)std::cos(a);
            tw256[2 * (p * 16 + k1) + 1] = (float)-std::sin(a);
        }
    float* tw512 = (float*)(img + h.off_tw512);
    for (int k = 0; k < 128; ++k) {
        const double a = 2.0 * M_PI * (double)k / 512.0;
        tw512[2 * k] = (float)std::cos(a); tw512[2 * k + 1] = (float)std::sin(a);
    }
    int32_t* mklo = (int32_t*)(img + h.off_mel_b);
    float* mb = (float*)(img + h.off_mel_b) + 64;
    for (int l = 0; l < 64; ++l) mklo[l] = grp_klo[l >> 2];
    for (int st = 0; st < mel_S; ++st)
        for (int l = 0; l < 64; ++l) {                               // lane l = mel l (block l/4, column l%4)
            const int k = grp_klo[l >> 2] + st;
            // x 1/4 (exact): the front-end's power tile holds 4 |X|^2 (the real-FFT split without its two halvings)
            mb[((size_t)(st >> 2) * 64 + l) * 4 + (st & 3)] = k < nb ? 0.25f * w->mel_fb[k * NMEL + l] : 0.0f;
        }

    // ---- init_bn -> scale/shift
    float* sc = (float*)(img + h.off_bn_sc); float* sh = (float*)(img + h.off_bn_sh);
    for (int m = 0; m < NMEL; ++m) {
        const double s = (double)w->bn_weight[m] / std::sqrt((double)w->bn_var[m] + 1e-5);
        sc[m] = (float)s; sh[m] = (float)((double)w->bn_bias[m] - (double)w->bn_mean[m] * s);
    }
    // ---- everything that is folded at pack time, in double (shared with the calibration forward of uitk_calibrate)
    Folded F;
    fold_model(c, w, &F);
    // ---- conv bias + pos-embeds (rows 0..23) and the cls token (row 24), centred
    float* peb = (float*)(img + h.off_pe_bias);
    for (int i = 0; i < (NTOK + 1) * DIM; ++i) peb[i] = (float)F.pe_bias[i];
    // ---- final norms
    float* fin = (float*)(img + h.off_fin);
    std::memcpy(fin + 0 * DIM, w->norm_w, DIM * 4); std::memcpy(fin + 1 * DIM, w->norm_b, DIM * 4);
    std::memcpy(fin + 2 * DIM, w->head_ln_w, DIM * 4); std::memcpy(fin + 3 * DIM, w->head_ln_b, DIM * 4);

    // ---- weight stream
    uint16_t* ws = (uint16_t*)(img + h.off_wstream);
    auto frag_ptr = [&](int stage, int f) { return ws + ((size_t)stage * STAGE_FRAGS + f) * (FRAG_BYTES / 2); };
    // patch embed: k-slot (s,g,j) <-> (mel-in-patch fi = 2s + (g>>1), frame-in-patch ti = 8(g&1) + j), i.e. operand
    // offset fi*16 + ti = 32s + 8g + j in the front-end's [mel][frame] patch; Conv2d weight [oc][0][fi][ti]
    // conv weights: structured rounding, column groups = the 16 frames of one mel row of the patch (k = fi*16 + ti); the
    // low-order fragments hold the rounding residuals of the high-order ones (split patch embed, optional at run time)
    bool structured = true;
#ifdef UITK_LAB
    structured = std::getenv("UITK_PACK_RTN") == nullptr;       // lab A/B: UITK_PACK_RTN=1 -> plain round-to-nearest
#endif
    auto Wpe = [&](int r, int k) { return F.pe_w[(size_t)r * 256 + k]; };
    // CALIBRATED ROUNDING (round_gptq) of every single-weight linear when calibration statistics are given: the single-term modes (the
    // split modes f16x2 / f16l6 carry the rounding residual as a second term).  Round 5: the fp8 modes too -- qkv / fc1 / fc2 are
    // rounded on THEIR grid (MX e4m3: round_fp8_grid), the patch embed and proj, which stay fp16 there, on the fp16 grid.
    // One task per matrix (1 + 4 per block, independent), spread over host threads: ~0.1 s for UiT-XS.
    const bool calibrated = calib != nullptr && !x2 && !l6;
    // OPERAND CENTRING of the attention (round 5), whenever calibration statistics are given (any fp16-operand precision,
    // split weights included -- the means are what matters here, not the rounding):
    //   * k - c_k leaves the softmax unchanged for ANY constant c_k, v - c_v moves Wp c_v into the proj bias: with c = the calibration
    //     mean of k / v the fp16 roundings of k, v and of the attention output act on the small centred values instead of on a large
    //     common part (a massive-activation channel of LayerNorm-1 is ~ the same in every token) -- exact algebra, free at run time;
    //   * the builds named by operand_centring() also subtract mu = E[LayerNorm-1 output] from the operand before it is rounded
    //     (W (x - mu) + (b + W mu), exact for any mu): then the k / v biases are exactly 0 and -mu rides in the block's bias table.
    // tools/err_attrib_full.py, d4-GELU seed 206 (the round-4 gate case): fp16 activation floor 9.8e-4 -> 2.1e-4.
    const int kprec = (c->precision == UITK_PREC_FP8W || c->precision == UITK_PREC_F16A) ? UITK_PREC_F16 : c->precision == UITK_PREC_F16AX ? UITK_PREC_F16X2 : c->precision;
    // (a build with the runtime centring ALWAYS reads -mu from its bias table -- for full-width attention from the K rows' bias slot -- so
    // its images are folded this way with mu = 0 when there are no statistics)
    const bool centre_rt = operand_centring(c->attention_full, c->act_gelu, c->pooling, kprec);
    const bool centre = centre_rt || (calib != nullptr && (kprec == UITK_PREC_F16 || kprec == UITK_PREC_F16X2 || kprec == UITK_PREC_F16L6));
    const std::vector<double> mu_zero(DIM, 0.0);
    // (with the runtime centring the statistics block holds the LayerNorm-1 moment of the CENTRED operand, round16(x - mu): what qkv
    // multiplies -- calibrate.hip / tests/calib_ref.py form it that way for these configurations)
    const bool grid8 = w8 || (prec == UITK_PREC_FP8 && calibrated);     // qkv / fc1 / fc2 as VALUES on the MX e4m3 grid (Q8_*)
    // scale blocks: precision fp8 = the hardware's 32-element blocks of a fragment8 (uitk_dev.h: per 128 k, block 2 (kk >> 6) + ((kk >> 3) & 1));
    // the weights-only variant = 32 consecutive k
    const std::function<int(int)> block_of = w8 ? std::function<int(int)>([](int k) { return k >> 5; })
                                                : std::function<int(int)>([](int k) { const int kk = k & 127; return 4 * (k >> 7) + 2 * (kk >> 6) + ((kk >> 3) & 1); });
    std::vector<uint16_t> pe_q;
    std::vector<std::vector<uint16_t>> cq_qkv(depth), cq_proj(depth), cq_fc1(depth), cq_fc2(depth);
    std::vector<std::vector<double>> Q8_qkv(depth), Q8_fc1(depth), Q8_fc2(depth);
    auto grid8_block = [&](int b, const double* base, const CalibOffsets* co) {       // base == nullptr: every element to nearest
        const FoldedBlock& fb = F.blocks[b];
        Q8_qkv[b] = round_fp8_grid(3 * inner, DIM, [&](int r, int k) { return fb.qkv_w[(size_t)r * DIM + k]; }, block_of, DIM / 32,
                                   base ? base + co->ln1 : nullptr);
        Q8_fc1[b] = round_fp8_grid(HID, DIM, [&](int r, int k) { return fb.fc1_w[(size_t)r * DIM + k]; }, block_of, DIM / 32, base ? base + co->ln2 : nullptr);
        Q8_fc2[b] = round_fp8_grid(DIM, HID, [&](int r, int k) { return fb.fc2_w[(size_t)r * HID + k]; }, block_of, HID / 32, base ? base + co->hid : nullptr);
    };
    if (calibrated) {
        const CalibOffsets co = calib_offsets(c);
        {   // a NaN on a diagonal makes round_gptq's column order comparator a non-strict-weak ordering (undefined behaviour in
            // std::stable_sort) and a negative one is no second moment: reject instead of sorting
            auto diag_ok = [&](const double* H, int n) { for (int i = 0; i < n; ++i) { const double d = H[(size_t)i * n + i]; if (!std::isfinite(d) || d < 0.0) return false; } return true; };
            bool ok = diag_ok(calib + co.pe, 256);
            for (int b = 0; b < depth && ok; ++b) {
                const double* base = calib + (size_t)b * co.per_block;
                ok = diag_ok(base + co.ln1, DIM) && diag_ok(base + co.att, inner) && diag_ok(base + co.ln2, DIM) && diag_ok(base + co.hid, HID);
            }
            if (!ok) return UITK_ERR_INVALID_ARG;
        }
        const int ntask = 1 + 4 * depth;
        std::atomic<int> next{0};
        auto worker = [&]() {
            for (int t = next.fetch_add(1); t < ntask; t = next.fetch_add(1)) {
                if (t == 0) { pe_q = round_gptq(DIM, 256, Wpe, prec, calib + co.pe); continue; }
                const int b = (t - 1) / 4, which = (t - 1) % 4;
                const FoldedBlock& fb = F.blocks[b];
                const double* base = calib + (size_t)b * co.per_block;
                if (which == 1) { cq_proj[b] = round_gptq(DIM, inner, [&](int r, int k) { return fb.proj_w[(size_t)r * inner + k]; }, prec, base + co.att); continue; }
                if (grid8) {                                 // one of the three tasks does the block's three grid roundings
                    if (which == 0) grid8_block(b, base, &co);
                    continue;
                }
                if (which == 0) cq_qkv[b] = round_gptq(3 * inner, DIM, [&](int r, int k) { return fb.qkv_w[(size_t)r * DIM + k]; }, prec, base + co.ln1);
                else if (which == 2) cq_fc1[b] = round_gptq(HID, DIM, [&](int r, int k) { return fb.fc1_w[(size_t)r * DIM + k]; }, prec, base + co.ln2);
                else cq_fc2[b] = round_gptq(DIM, HID, [&](int r, int k) { return fb.fc2_w[(size_t)r * HID + k]; }, prec, base + co.hid);
            }
        };
        unsigned nth = std::thread::hardware_concurrency();
        nth = nth == 0 ? 4 : nth > 16 ? 16 : nth;
        // no exception may cross the extern "C" boundary above: a thread that cannot be created (std::system_error) just means
        // fewer workers -- this thread drains the task counter anyway; an allocation failure inside a task is reported as an error code
        std::vector<std::thread> pool;
        std::atomic<bool> failed{false};
        auto guarded = [&]() { try { worker(); } catch (...) { failed.store(true); } };
        try {
            for (unsigned i = 1; i < nth; ++i) pool.emplace_back(guarded);
        } catch (...) {}
        guarded();
        for (auto& th : pool) th.join();
        if (failed.load()) return UITK_ERR_INVALID_ARG;
    } else {
        pe_q = round_structured(DIM, 256, Wpe, prec, structured ? 16 : 1);
        if (grid8) for (int b = 0; b < depth; ++b) grid8_block(b, nullptr, nullptr);      // fp8w without statistics: nearest on the grid
    }
    for (int i = 0; i < PE_HI_STAGES; ++i)
        for (int f = 0; f < STAGE_FRAGS; ++f) {
            const int ft = 2 * i + f / 8, s = f % 8;
            uint16_t* dst = frag_ptr(i, f);
            uint16_t* dlo = frag_ptr(PE_HI_STAGES + i, f);              // low-order term, same fragment order
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int g = l >> 4, fi = 2 * s + (g >> 1), ti = 8 * (g & 1) + j, row = 16 * ft + (l & 15);
                    dst[l * 8 + j] = pe_q[(size_t)row * 256 + fi * 16 + ti];
                    dlo[l * 8 + j] = cvt(Wpe(row, fi * 16 + ti) - op16_value(dst[l * 8 + j], prec), prec);
                }
        }
    float* bias = (float*)(img + h.off_bias);
    for (int b = 0; b < depth; ++b) {
        const FoldedBlock& fb = F.blocks[b];
        const int st = PE_STAGES + b * blk_stages;
        float* bb = bias + (size_t)b * BIAS_FLOATS;
        // the folded weights; with calibration statistics: their calibrated 16-bit values (exactly representable, so the
        // conversions below reproduce them)
        auto Wqkv = [&](int r, int k) { return grid8 ? Q8_qkv[b][(size_t)r * DIM + k] : calibrated ? op16_value(cq_qkv[b][(size_t)r * DIM + k], prec) : fb.qkv_w[(size_t)r * DIM + k]; };
        auto Wproj = [&](int r, int k) { return calibrated ? op16_value(cq_proj[b][(size_t)r * inner + k], prec) : fb.proj_w[(size_t)r * inner + k]; };
        auto Wfc1 = [&](int r, int k) { return grid8 ? Q8_fc1[b][(size_t)r * DIM + k] : calibrated ? op16_value(cq_fc1[b][(size_t)r * DIM + k], prec) : fb.fc1_w[(size_t)r * DIM + k]; };
        auto Wfc2_exact = [&](int r, int k) { return fb.fc2_w[(size_t)r * HID + k]; };
        // fc2 consumes ReLU / GELU outputs (a common non-negative level): structured rounding over each 32-wide k-step; the fp8
        // stream quantises the exact values itself
        std::vector<uint16_t> fc2_q;
        if (grid8) {}                                                    // (values on the e4m3 grid: Q8_fc2)
        else if (calibrated && prec != UITK_PREC_FP8) fc2_q = cq_fc2[b];
        else if (prec != UITK_PREC_FP8) fc2_q = round_structured(DIM, HID, Wfc2_exact, prec, structured ? 32 : 1);
        auto Wfc2 = [&](int r, int k) { return grid8 ? Q8_fc2[b][(size_t)r * HID + k] : prec == UITK_PREC_FP8 ? Wfc2_exact(r, k) : op16_value(fc2_q[(size_t)r * HID + k], prec); };
        if (!centre) {
            for (int r = 0; r < 3 * inner; ++r) bb[BIAS_QKV + r] = (float)fb.qkv_b[r];
            for (int r = 0; r < DIM; ++r) bb[b_proj + r] = (float)fb.proj_b[r];
        } else {
            // (the weights the kernel multiplies with: the calibrated / grid values where there are any, else the exact ones -- a split
            // mode's hi + lo is the exact weight to 22 bits)
            const double* mu = calib ? calib + calib_offsets(c).mu1 + (size_t)b * DIM : mu_zero.data();
            std::vector<double> wmu(3 * inner), cv(inner);
            for (int r = 0; r < 3 * inner; ++r) { double a = 0.0; for (int k = 0; k < DIM; ++k) a += Wqkv(r, k) * mu[k]; wmu[r] = a; }
            for (int r = 0; r < inner; ++r) bb[BIAS_QKV + r] = (float)(fb.qkv_b[r] + (centre_rt ? wmu[r] : 0.0));            // q
            for (int r = inner; r < 3 * inner; ++r) bb[BIAS_QKV + r] = (float)(centre_rt ? 0.0 : -wmu[r]);                   // k - c_k, v - c_v
            for (int j = 0; j < inner; ++j) cv[j] = wmu[2 * inner + j] + fb.qkv_b[2 * inner + j];                           // c_v = E[v]
            for (int r = 0; r < DIM; ++r) {
                double a = fb.proj_b[r];
                for (int j = 0; j < inner; ++j) a += Wproj(r, j) * cv[j];
                bb[b_proj + r] = (float)a;
            }
            if (centre_rt) {
                float* nm = bb + (full ? BIASF_QKV + 128 : BIAS_NEGMU);      // full-width attention: the K rows' (zero) bias slot
                for (int k = 0; k < DIM; ++k) nm[k] = (float)-mu[k];
            }
        }
        for (int r = 0; r < HID; ++r) bb[b_fc1 + r] = (float)fb.fc1_b[r];
        for (int r = 0; r < DIM; ++r) bb[b_fc2 + r] = (float)fb.fc2_b[r];
        if (prec == UITK_PREC_FP8) {
            // fp8 stream (uitk_dev.h): stage 0 qkv, stage 1 proj (f16) + MLP scales, then fc1 / fc2 per k-group q
            uint8_t* s0 = (uint8_t*)frag_ptr(st, 0);
            uint8_t* s1 = (uint8_t*)frag_ptr(st + 1, 0);
            uint8_t sc[64];
            for (int t = 0; t < 6; ++t) {
                pack_frag8(s0 + t * FRAG8_BYTES, sc, Wqkv, t, 0);
                for (int l = 0; l < 64; ++l) s0[FP8_QKV_SCALES + (t >> 2) * 256 + l * 4 + (t & 3)] = sc[l];
            }
            for (int ft = 0; ft < 8; ++ft) pack_frag(frag_ptr(st + 1, ft), prec, Wproj, ft, 0);
            for (int q = 0; q < 3; ++q) {
                uint8_t* sf1 = (uint8_t*)frag_ptr(st + 2 + 2 * q, 0);
                uint8_t* sf2 = (uint8_t*)frag_ptr(st + 3 + 2 * q, 0);
                for (int i = 0; i < 8; ++i) {
                    const int t1 = 8 * q + i;                          // fc1 tile (hidden rows 16 t1 ..), all 128 inputs
                    pack_frag8(sf1 + i * FRAG8_BYTES, sc, Wfc1, t1, 0);
                    for (int l = 0; l < 64; ++l) s1[FP8_MLP_SCALES + (t1 >> 2) * 256 + l * 4 + (t1 & 3)] = sc[l];
                    const int t2 = 8 * q + i;                          // fc2 fragment: output tile ft = i, hidden k-group q
                    pack_frag8(sf2 + i * FRAG8_BYTES, sc, Wfc2, i, 128 * q);
                    for (int l = 0; l < 64; ++l) s1[FP8_MLP_SCALES + 6 * 256 + (t2 >> 2) * 256 + l * 4 + (t2 & 3)] = sc[l];
                }
            }
            continue;
        }
        if (l6) {
            // precision f16l6 (uitk_dev.h): the f16 stages of precision f16 + the low-order term lo = W - value(hi) of fc1 / fc2 as
            // fragment6 and of proj as full fp16 fragments (qkv keeps single weights); row scales into the bias table
            auto lo_of = [&](const std::function<double(int, int)>& W) {
                return [&W, prec](int r, int k) { const double v = W(r, k); return v - op16_value(cvt(v, prec), prec); };
            };
            const std::function<double(int, int)> fq = Wqkv, fp = Wproj, f1 = Wfc1, f2x = Wfc2_exact, f2h = Wfc2;
            const std::function<double(int, int)> lp = lo_of(fp), l1 = lo_of(f1);
            const std::function<double(int, int)> l2 = [&](int r, int k) { return f2x(r, k) - f2h(r, k); };
            uint8_t* blk_base = (uint8_t*)(img + h.off_wstream) + (size_t)PE_STAGES * STAGE_BYTES + (size_t)b * L6_BLOCK_BYTES;
            uint8_t* sc_tab = (uint8_t*)(bb + L6_SCALES);                 // [quad][row][4 fragments]
            uint8_t sc[16];
            auto put_scales = [&](int frag_index) { for (int row = 0; row < 16; ++row) sc_tab[((frag_index >> 2) * 16 + row) * 4 + (frag_index & 3)] = sc[row]; };
            auto f16frag = [&](uint8_t* stage, int f) { return (uint16_t*)(stage + (size_t)f * FRAG_BYTES); };
            uint8_t* sqk = blk_base;
            for (int f = 0; f < 16; ++f) pack_frag(f16frag(sqk, f), prec, fq, f / 4, f % 4);
            uint8_t* svp = blk_base + L6_QK_BYTES;
            for (int f = 0; f < 8; ++f) pack_frag(f16frag(svp, f), prec, fq, 4 + f / 4, f % 4);
            for (int ft = 0; ft < 8; ++ft) { pack_frag(f16frag(svp, 8 + ft), prec, fp, ft, 0); pack_frag(f16frag(svp, 16 + ft), prec, lp, ft, 0); }
            for (int ch = 0; ch < 12; ++ch) {
                uint8_t* sm = blk_base + l6_mlp_offset(ch);
                for (int f = 0; f < 8; ++f) pack_frag(f16frag(sm, f), prec, f1, 2 * ch + f / 4, f % 4);
                for (int ft = 0; ft < 8; ++ft) pack_frag(f16frag(sm, 8 + ft), prec, f2h, ft, ch);
                for (int i = 0; i < 2; ++i) { pack_frag6(sm + STAGE_BYTES + i * FRAG6_BYTES, sc, l1, 2 * ch + i, 0); put_scales(2 * ch + i); }
                if ((ch & 3) == 3)
                    for (int ft = 0; ft < 8; ++ft) { pack_frag6(sm + L6_M_BYTES + ft * FRAG6_BYTES, sc, l2, ft, 128 * (ch / 4)); put_scales(24 + 8 * (ch / 4) + ft); }
            }
            continue;
        }
        if (x2) {
            // precision f16x2 (uitk_dev.h): lo = W - value(hi), hi as the single-term modes round it (fc2: structured)
            auto lo_of = [&](const std::function<double(int, int)>& W) {
                return [&W, prec](int r, int k) { const double v = W(r, k); return v - op16_value(cvt(v, prec), prec); };
            };
            const std::function<double(int, int)> fq = Wqkv, fp = Wproj, f1 = Wfc1, f2x = Wfc2_exact, f2h = Wfc2;
            const std::function<double(int, int)> lq = lo_of(fq), lp = lo_of(fp), l1 = lo_of(f1);
            const std::function<double(int, int)> l2 = [&](int r, int k) { return f2x(r, k) - f2h(r, k); };
            auto hi_lo = [&](int stage, int slot8, const std::function<double(int, int)>& Whi, const std::function<double(int, int)>& Wlo, int tile, int nk) {
                for (int s = 0; s < nk; ++s) {                          // tile slot: hi k-steps, then the lo k-steps
                    pack_frag(frag_ptr(stage, slot8 * 8 + s), prec, Whi, tile, s);
                    pack_frag(frag_ptr(stage, slot8 * 8 + 4 + s), prec, Wlo, tile, s);
                }
            };
            if (full) {
                // full-width Attention (uitk_dev.h, BLOCK_STAGES_X2_ATT): qkv row tile which*8 + hd*4 + dt -> stage 2 (3 hd + which) + dt/2
                for (int hd = 0; hd < 2; ++hd)
                    for (int which = 0; which < 3; ++which)
                        for (int dt = 0; dt < 4; ++dt) hi_lo(st + 2 * (3 * hd + which) + dt / 2, dt & 1, fq, lq, which * 8 + hd * 4 + dt, 4);
                for (int ft = 0; ft < 8; ++ft) hi_lo(st + 12 + ft / 2, ft & 1, fp, lp, ft, 4);
                for (int ch = 0; ch < 12; ++ch) {
                    for (int i = 0; i < 2; ++i) hi_lo(st + 16 + 2 * ch, i, f1, l1, 2 * ch + i, 4);
                    for (int ft = 0; ft < 8; ++ft) { pack_frag(frag_ptr(st + 17 + 2 * ch, ft), prec, f2h, ft, ch); pack_frag(frag_ptr(st + 17 + 2 * ch, 8 + ft), prec, l2, ft, ch); }
                }
                continue;
            }
            for (int t = 0; t < 6; ++t)                                  // qkv row tiles 0..5 -> stages 0, 1, 2, half t & 1
                for (int s = 0; s < 4; ++s) {
                    pack_frag(frag_ptr(st + t / 2, (t & 1) * 8 + s), prec, fq, t, s);
                    pack_frag(frag_ptr(st + t / 2, (t & 1) * 8 + 4 + s), prec, lq, t, s);
                }
            for (int ft = 0; ft < 8; ++ft) { pack_frag(frag_ptr(st + 3, ft), prec, fp, ft, 0); pack_frag(frag_ptr(st + 3, 8 + ft), prec, lp, ft, 0); }
            for (int ch = 0; ch < 12; ++ch) {
                for (int i = 0; i < 2; ++i)
                    for (int s = 0; s < 4; ++s) {
                        pack_frag(frag_ptr(st + 4 + 2 * ch, i * 8 + s), prec, f1, 2 * ch + i, s);
                        pack_frag(frag_ptr(st + 4 + 2 * ch, i * 8 + 4 + s), prec, l1, 2 * ch + i, s);
                    }
                for (int ft = 0; ft < 8; ++ft) { pack_frag(frag_ptr(st + 5 + 2 * ch, ft), prec, f2h, ft, ch); pack_frag(frag_ptr(st + 5 + 2 * ch, 8 + ft), prec, l2, ft, ch); }
            }
            continue;
        }
        if (full) {
            // full-width Attention (uitk_dev.h): qkv row = which*128 + head*64 + d (uit.py:147-148) -> row tile which*8 + head*4 + dt
            for (int hd = 0; hd < 2; ++hd)
                for (int which = 0; which < 3; ++which)
                    for (int f = 0; f < 16; ++f) pack_frag(frag_ptr(st + 3 * hd + which, f), prec, Wqkv, which * 8 + hd * 4 + f / 4, f % 4);
            for (int i = 0; i < 32; ++i) pack_frag(frag_ptr(st + 6 + (i >> 4), i & 15), prec, Wproj, i / 4, i % 4);
            for (int ch = 0; ch < 12; ++ch) {
                for (int f = 0; f < 8; ++f) pack_frag(frag_ptr(st + 8 + ch, f), prec, Wfc1, 2 * ch + f / 4, f % 4);
                for (int ft = 0; ft < 8; ++ft) pack_frag(frag_ptr(st + 8 + ch, 8 + ft), prec, Wfc2, ft, ch);
            }
            continue;
        }
        // stage QK: ft = f/4 over qkv rows 0..63 (q h0, q h1, k h0, k h1), s = f%4
        for (int f = 0; f < 16; ++f) pack_frag(frag_ptr(st, f), prec, Wqkv, f / 4, f % 4);
        // stage VP: V rows 64..95 (tile 4 + h), then proj (K = 32: one k-step)
        for (int f = 0; f < 8; ++f) pack_frag(frag_ptr(st + 1, f), prec, Wqkv, 4 + f / 4, f % 4);
        for (int ft = 0; ft < 8; ++ft) pack_frag(frag_ptr(st + 1, 8 + ft), prec, Wproj, ft, 0);
        // MLP chunks
        for (int ch = 0; ch < 12; ++ch) {
            for (int f = 0; f < 8; ++f) pack_frag(frag_ptr(st + 2 + ch, f), prec, Wfc1, 2 * ch + f / 4, f % 4);
            for (int ft = 0; ft < 8; ++ft) pack_frag(frag_ptr(st + 2 + ch, 8 + ft), prec, Wfc2, ft, ch);
        }
    }
    // ---- head
    uint16_t* hw = (uint16_t*)(img + h.off_head_w);
    auto Whead = [&](int r, int k) { return r < C ? (double)w->head_w[r * DIM + k] : 0.0; };
    // The classifier is the most rounding-sensitive GEMM of the path (it alone caused ~3/4 of the fp16 logit error),
    // and it is tiny: it runs on split operands, W = hi + lo (two 16-bit terms), layout [ot][s][hi|lo].
    auto op_to_double = [&](uint16_t u) -> double {
        if (prec == UITK_PREC_BF16) { uint32_t x = (uint32_t)u << 16; float f; std::memcpy(&f, &x, 4); return f; }
        const uint32_t sign = (u >> 15) & 1u, e = (u >> 10) & 31u, m = u & 1023u;
        double v = e == 0 ? std::ldexp((double)m, -24) : std::ldexp((double)(m | 1024u), (int)e - 25);
        return sign ? -v : v;
    };
    auto Whead_lo = [&](int r, int k) { const double w = Whead(r, k); return w - op_to_double(cvt(w, prec)); };
    for (int ot = 0; ot < ntiles; ++ot)
        for (int s = 0; s < 4; ++s) {
            pack_frag(hw + ((size_t)(ot * 4 + s) * 2 + 0) * (FRAG_BYTES / 2), prec, Whead, ot, s);
            pack_frag(hw + ((size_t)(ot * 4 + s) * 2 + 1) * (FRAG_BYTES / 2), prec, Whead_lo, ot, s);
        }
    float* hb = (float*)(img + h.off_head_b);
    for (int r = 0; r < C; ++r) hb[r] = w->head_b[r];
    if (hdr_out) *hdr_out = h;
    return UITK_OK;
}

}  // namespace uitk
