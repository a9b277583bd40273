// libccsm host side: C-ABI (include/ccsm.h), weight fragment packing, workspaces, kernel sequencing.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include <functional>

#include "../../include/ccsm.h"
#include "ccsm_kernels.hip"
#include "ccsm_gru_f8.hip"
#include "ccsm_gru_mx.hip"
#include "ccsm_gru_f3.hip"
#include "ccsm_gru_f3s.hip"
#include "ccsm_gru_mx16.hip"
#include "ccsm_aggr.hip"
#include "ccsm_extract.hip"
#include "ccsm_ceiling.hip"

#include <cmath>
#include <random>

using namespace ccsm;

namespace {

thread_local std::string g_err;

ccsm_status fail(ccsm_status st, const std::string& msg) {
    g_err = msg;
    return st;
}

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return fail(CCSM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                       \
    } while (0)

constexpr size_t kReadTableBytes = 8 + 64 + 4 * 5 + 4 + 8;   // per read: offset | stats | length, fn, rn, nsites, first_site | spare | h0 key
constexpr int kNBGru2 = 3;   // batch tiles (of 32 rows) per workgroup of the GRU kernels
constexpr int kRowPad = 192;                    // rows are padded to whole workgroups of 96 AND of 64 rows (the split-mx family's small-launch form)
// GRU kernels' dynamic LDS: h fragments + x chunk ring + 4 KiB bias table
constexpr int gru2_lds(int kx) { return (kKBH * kNBGru2 * 2 + 2 * (kx >= 4 ? 4 : kx) * kNBGru2 * 2) * 1024 + kWaves * 4 * 32 * 4; }
// attention kernel dynamic LDS: 2 staging buffers x 28 KiB + e partials + fc partials + fc1.weight
constexpr int kAttF8Lds = 3 * 28 * 1024 + kWaves * 7 * 32 * 4 + kSeqLen * 32 * 4 + kSeqLen * 32 * 2 * 4 + kAttFc3 * 16 + kHidden * 4 + kWaves * 4 * 64 * 16;   // attn_fc_f8_kernel: three staging buffers (14 hi + 7 lo fragments + 7 KiB derived fp8), one group's score partials, scores, fc partials, fc1 tile fragments, va, q = 139.9 KiB
constexpr int kAttLds = 2 * 28 * 1024 + kWaves * kSeqLen * 32 * 4 + kSeqLen * 32 * 2 * 4 + kAttFc3 * 16 + kHidden * 4;   // staging, score partials, fc partials, fc1 tile fragments, va

inline int rows_padded(int n_sites) { return ((2 * n_sites + kRowPad - 1) / kRowPad) * kRowPad; }

struct HalfPair {
    _Float16 hi, lo;
};
inline HalfPair split_host(float v) {
    HalfPair p;
    p.hi = (_Float16)v;
    p.lo = (_Float16)(v - (float)p.hi);
    return p;
}
inline int crow(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }  // MFMA 32x32 C row of reg r, lane half hh

}  // namespace

struct ccsm_model {
    int device = 0;
    int precision = 3;
    uint4* wst2[kLayers] = {nullptr, nullptr, nullptr}; // split3: [dir][wave][A: KX x (r,z) | B: 16 x (r,z,n) | C: KX x (n)][hl][64]
    float mx_quant_err = 0.f;                            // relative RMS quantisation error of the weight correction blobs (worst layer)
    int feat = kFeatNpass;                               // optional input features of the variant (kFeat* bits); feat0 = columns of weight_ih_l0
    int feat0 = kFeat0;
    int fold = 0;                                        // 1: embedding folded into the layer-0 matrix, rows are [one-hot(5) | features] (pack_x0_kernel)
    float probe_err_hybrid = -1.f;                       // ... of the hybrid arithmetic (-1: not run: split-mx was accepted)
    float probe_tail = -1.f, probe_tail_hybrid = -1.f;   // fraction of the probe sites beyond kProbeTailAt
    float probe_err = -1.f;                              // max |dprob| split-mx vs split-fp16 on the probe batch of ccsm_create (-1: not run)
    float probe_err_dyn = -1.f, probe_tail_dyn = -1.f;   // ... of split-mx-d (never probed since round 4: explicit choices only; kept for the ABI)
    bool l0_stag = true;                                 // layer 0 in its staggered form (96-row launches); CCSM_L0_LOCKSTEP=1 at ccsm_create: the lock-step form (A/B)
    bool split3_v2 = false;                              // CCSM_SPLIT3_V2=1 at ccsm_create: split3 on the round-1 kernel (A/B)
    float probe_q999 = -1.f;                             // 99.9th percentile of |dprob| split-mx vs split-fp16 over the probe sites
    int probe_n = 0;                                     // probe sites actually run (the probe stops at the first batch that decides it)
    uint4* wstmd[kLayers] = {nullptr, nullptr, nullptr};// split-mx-d weight streams (fp6 recurrent blobs)
    uint4* wstmx[kLayers] = {nullptr, nullptr, nullptr};// split-mx weight streams (ccsm_gru_mx.hip: hi fragments + MX correction blobs)
    uint4* wstf3[kLayers] = {nullptr, nullptr, nullptr};// split3 on the split-mx schedule (ccsm_gru_f3.hip): [0] = the hybrid's layer-0 stream, [1], [2] three-pass streams
    uint4* wstf3s[kLayers] = {nullptr, nullptr, nullptr};// split3, layers 1-2 on v_mfma_f32_16x16x32_f16 (ccsm_gru_f3s.hip): the same stream sizes, unit tiles of 16 ([0] unused)
    float* biasn[kLayers] = {nullptr, nullptr, nullptr}; // its biases in natural unit order [dir][wave][4][32] ([0] unused)
    bool f3_shape32 = false;                             // CCSM_F3_SHAPE32=1 at ccsm_create: split3's layers 1-2 on the 32x32x16 kernel of round 4 (A/B)
    uint4* wstmx16[kLayers] = {nullptr, nullptr, nullptr};// split-mx, layers 1-2 on the 16-wide instructions (ccsm_gru_mx16.hip; [0] unused)
    bool mx_shape16 = false;                             // CCSM_MX_SHAPE16=1 at ccsm_create: plain split-mx's layers 1-2 on gru_layer12_mx16_kernel (round 6: parity-green,
                                                         // 10 % slower than gru_layer12_mx_kernel - both sit on the CU's vector-memory path, DESIGN 7 - so not the default)
    uint4* wsthy[kLayers] = {nullptr, nullptr, nullptr};// hybrid weight streams (the same with fp16 lo fragments for the recurrent part)
    uint4* wa3 = nullptr;                                // split-f8 attention projections [wave][32][hi|corr][64]
    uint4* ua3 = nullptr;
    int att_scale[2] = {127, 127};                       // E8M0 scales of the Wa / Ua corr operands
    uint4* fc3 = nullptr;                                // fc1.weight as four rows (class x strand half) of split-f8 A fragments, compact: pack_fc_v3
    int fc_scale = 127;
    uint4* fc3s = nullptr;                               // the same tile as fp16 hi | lo fragments (attn_fc_kernel: three passes): pack_fc_s3
    float* bias[kLayers] = {nullptr, nullptr, nullptr};  // [dir][wave][4][hh][16]
    uint4* wa = nullptr;                                 // [wave][32][hl][64]
    uint4* ua = nullptr;
    float* va = nullptr;                                 // [wave][hh][16]
    float* fcw = nullptr;                                // (2,1024)
    float* fcb = nullptr;                                // (2)
    float* embed = nullptr;                              // (5,8)
    // the probe on the CALLER's data (ccsm_model_data_probe_*): |dprob| candidate vs split3 of every site handed in so far, and the verdict
    std::vector<float> dprobe;
    float dprobe_err = -1.f, dprobe_q999 = -1.f;
    int dprobe_n = 0;
    int dprobe_ok = -1;                                  // -1: not decided; 0: the candidate failed on the caller's data (split3 is served); 1: kept
    mutable int slices_bound = 0;                        // slices added to workspaces of this model and not yet run (ccsm_model_set_precision refuses while > 0)
};

struct ccsm_workspace {
    int device = 0;
    int max_sites = 0;
    int rows_p = 0;
    size_t bytes = 0;
    uint4* x0 = nullptr;
    uint4* act[2] = {nullptr, nullptr};
    float* h0buf = nullptr;
    float* part = nullptr;
    unsigned long long* dbg = nullptr;   // phase time stamps of one GRU layer's workgroup 0: only in a -DCCSM_PHASE_STAMPS build
    // device staging for the host-pointer path
    uint8_t* d_in = nullptr;   // features of both strands
    float* d_h0 = nullptr;     // explicit h0 (2 x 6 x max_sites x 256), allocated on first explicit use
    float* d_out = nullptr;    // logits | probs
    uint8_t* d_keys = nullptr; // per-site random-stream keys (u64 x max_sites | u32 x max_sites), allocated on first use
    uint8_t* p_keys = nullptr;
    uint8_t* p_in = nullptr;   // pinned host mirrors
    float* p_h0 = nullptr;
    float* p_out = nullptr;
    size_t in_bytes = 0;
    int pending_sites = 0;
    hipStream_t pending_stream = nullptr;
    // slices added since the last run (ccsm_group_add_device / ccsm_group_run)
    int n_slices = 0;
    int rows_used = 0;
    int slice_row[kMaxSlices] = {};
    int slice_n[kMaxSlices] = {};
    float* slice_logits[kMaxSlices] = {};
    float* slice_probs[kMaxSlices] = {};
    // read-level path (ccsm_forward_reads_host): raw per-read byte arrays and per-read tables, grown on demand
    uint8_t* r_bytes = nullptr;      // seq | fi | ri | fp | rp, each r_cap_bases long
    size_t r_cap_bases = 0;
    uint8_t* r_table = nullptr;      // offset i64 | stats f64 x8 | length i32 | fn | rn | nsites | first_site
    int r_cap_reads = 0;
    int* r_locs = nullptr;           // (max_sites)
    uint8_t* rp_bytes = nullptr;     // pinned mirrors of r_bytes / r_table / r_locs
    uint8_t* rp_table = nullptr;
    int* rp_locs = nullptr;
    bool r_pending = false;          // a read chunk is in flight (ccsm_submit_reads_host .. ccsm_wait_reads_host)
    bool r_checked = false;
    int r_nreads = 0, r_nsites = 0;
    hipStream_t r_stream = nullptr;
    bool force_split3 = false;               // the pending run holds explicit initial states outside split-mx's domain (or ccsm_workspace_force_split3: one-shot)
    const ccsm_model* bound_to = nullptr;    // the model whose slices_bound counts this workspace's group slices (ccsm_group_add_device .. ccsm_group_run)
    bool timing = false;
    static constexpr int kEvSets = 128;      // ring of event sets: one per run while timing is enabled
    hipEvent_t evs[kEvSets][8] = {};
    hipEvent_t* ev = evs[0];                 // the set of the current / last run
    int ev_runs = 0;                         // timed runs since timing was enabled
    bool ev_ok = false;
    bool timed = false;
};

namespace {

// Split-fp16 weight stream: per (dir, wave) the [hi | lo] fragments in the order the three phases consume them.
void pack_wstream_v2(int layer, int feat0, const float* const wih[2], const float* const whh[2], std::vector<_Float16>& out) {
    const int kx = layer_kx(layer);
    const int k_in = layer == 0 ? feat0 : 2 * kHidden;
    const int nfrag = kx * 4 + kKBH * 6 + kx * 2;   // (gate, hl) fragments per wave
    out.assign((size_t)2 * kWaves * nfrag * 512, (_Float16)0.f);
    for (int dir = 0; dir < 2; ++dir)
        for (int wave = 0; wave < kWaves; ++wave) {
            size_t f = (size_t)(dir * kWaves + wave) * nfrag;
            auto emit = [&](bool xpart, int kb, int g) {
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, q = lane >> 5;
                    const int row = g * kHidden + kUnitTile * wave + i;
                    for (int j = 0; j < 8; ++j) {
                        const int k = 16 * kb + 8 * q + j;
                        const float v = xpart ? (k < k_in ? wih[dir][(size_t)row * k_in + k] : 0.f) : whh[dir][(size_t)row * kHidden + k];
                        const HalfPair p = split_host(v);
                        out[f * 512 + lane * 8 + j] = p.hi;
                        out[(f + 1) * 512 + lane * 8 + j] = p.lo;
                    }
                }
                f += 2;
            };
            for (int kb = 0; kb < kx; ++kb) { emit(true, kb, 0); emit(true, kb, 1); }
            for (int kb = 0; kb < kKBH; ++kb) { emit(false, kb, 0); emit(false, kb, 1); emit(false, kb, 2); }
            for (int kb = 0; kb < kx; ++kb) emit(true, kb, 2);
        }
}

// OCP fp8 e4m3fn, round to nearest even, saturating at +-448 (what v_mfma_scale_f32_32x32x64_f8f6f4 decodes with FMT 0)
uint8_t fp8_e4m3_host(float v) {
    const uint8_t sign = std::signbit(v) ? 0x80 : 0x00;
    float a = std::fabs(v);
    if (!(a == a)) return 0x7f;
    if (a >= 448.0f) return sign | 0x7e;
    if (a < std::ldexp(1.0f, -10)) return sign;                       // below half of the smallest subnormal (2^-9)
    int e;
    (void)std::frexp(a, &e);                                          // a = m * 2^e, m in [0.5, 1)
    int ex = e - 1;                                                   // a in [2^ex, 2^(ex+1))
    if (ex < -6) ex = -6;                                             // subnormal range: step 2^-9
    const float step = std::ldexp(1.0f, ex - 3);
    const float q = std::nearbyint(a / step);                         // RNE (default rounding mode)
    float r = q * step;
    if (r >= 448.0f) r = 448.0f;
    if (r < std::ldexp(1.0f, -6)) return sign | (uint8_t)std::lrint(r / std::ldexp(1.0f, -9));
    int e2;
    const float m2 = std::frexp(r, &e2);                              // r = m2 * 2^e2, m2 in [0.5, 1)
    const int ebits = (e2 - 1) + 7;
    const int mbits = (int)std::lrint((m2 * 2.0f - 1.0f) * 8.0f);
    return sign | (uint8_t)(ebits << 3) | (uint8_t)mbits;
}

constexpr int kCorrPerm[16] = {0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15};   // see ccsm_gru_f8.hip

// log2 of the power-of-two pre-scale sw of one weight matrix: the largest with max|W| * sw <= 240 (fp8 e4m3 headroom)
int corr_scale_log2(const float* w, size_t count) {
    float mx = 0.f;
    for (size_t i = 0; i < count; ++i) mx = std::fmax(mx, std::fabs(w[i]));
    if (!(mx > 0.f) || !std::isfinite(mx)) return 0;
    int e = (int)std::floor(std::log2(240.0f / mx));
    return std::max(-100, std::min(100, e));
}

// corr fragment (1 KiB) of one (gate-row block, k-block): lane (i, g), byte j <-> k = 16 kb + kCorrPerm[j];
// g = 0: fp8(W_lo * 2^11 * sw), g = 1: fp8(W_hi * sw).  `get(row_in_block i, k)` returns the fp32 weight (0 outside).
template <typename Get>
void emit_corr_frag(uint8_t* dst, int kb, int log2sw, Get get) {
    for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 31, g = lane >> 5;
        for (int j = 0; j < 16; ++j) {
            const float v = get(i, 16 * kb + kCorrPerm[j]);
            const float hi = (float)(_Float16)v;
            const float c = g ? std::ldexp(hi, log2sw) : std::ldexp(v - hi, 11 + log2sw);
            dst[lane * 16 + j] = fp8_e4m3_host(c);
        }
    }
}

// ---- split-mx correction blobs (ccsm_gru_mx.hip).  MX element codes, round to nearest even, saturating:
//   fp6 e2m3: 1-2-3 bits, bias 1: 0 .. 0.875 in steps of 1/8 (subnormal), then 1 .. 7.5         fp4 e2m1: 0, 0.5, 1, 1.5, 2, 3, 4, 6
uint8_t mx_code(float v, int fmt) {
    const int mb = fmt == 2 ? 3 : 1;                                   // mantissa bits
    const float vmax = fmt == 2 ? 7.5f : 6.0f;
    const uint8_t sign = std::signbit(v) ? (uint8_t)(1u << (2 + mb)) : 0;
    float a = std::fabs(v);
    if (!(a == a)) a = 0.f;
    if (a > vmax) a = vmax;
    int e;
    (void)std::frexp(a, &e);
    int ex = a > 0.f ? e - 1 : 0;                                      // a in [2^ex, 2^(ex+1))
    if (ex < 0) ex = 0;                                                // subnormal range shares the step of [1, 2)
    const float step = std::ldexp(1.0f, ex - mb);
    float r = std::nearbyint(a / step) * step;
    if (r > vmax) r = vmax;
    uint8_t code;
    if (r < 1.0f) code = (uint8_t)std::lrint(r * (float)(1 << mb));    // exponent field 0
    else {
        int e2;
        const float m2 = std::frexp(r, &e2);                           // r = m2 2^e2, m2 in [0.5, 1)
        code = (uint8_t)(((e2 - 1 + 1) << mb) | (int)std::lrint((m2 * 2.0f - 1.0f) * (float)(1 << mb)));
    }
    return sign | code;
}
float mx_value(uint8_t code, int fmt) {
    const int mb = fmt == 2 ? 3 : 1;
    const int m = code & ((1 << mb) - 1), e = (code >> mb) & 3;
    const float v = e == 0 ? (float)m / (float)(1 << mb) : std::ldexp(1.0f + (float)m / (float)(1 << mb), e - 1);
    return (code >> (2 + mb)) & 1 ? -v : v;
}
inline int mx_perm(int j) { return 8 * ((j & 15) >> 2) + (j & 3) + 4 * (j >> 4); }   // blob position -> k within the pair (kMxPerm)

// One lane's weight blob of 32 values `val[j]` (already in blob order): E8M0 scale from the block's largest magnitude (max |val| s
// <= top of the format), codes packed little-endian (fp4: element j in nibble j & 1 of byte j >> 1; fp6: element j at bit 6 j) into
// dst[0..15] (+ dst8[0..7] for fp6); *scale = the scale byte of the TRUE value (exp_bias = -11 for the W_lo lanes, which carry
// W_lo 2^11).  err2 / ref2 accumulate the squared quantisation error and the squared values (in true units).
void emit_blob(uint8_t* dst16, uint8_t* dst8, uint8_t* scale, const float (&val)[32], int fmt, int exp_bias, double* err2, double* ref2) {
    float mx = 0.f;
    for (int j = 0; j < 32; ++j) mx = std::fmax(mx, std::fabs(val[j]));
    const float top = fmt == 2 ? 7.5f : 6.0f;
    int lg = 0;
    if (mx > 0.f && std::isfinite(mx)) lg = (int)std::floor(std::log2(top / mx));
    lg = std::max(-100, std::min(100, lg));
    uint8_t bits[24] = {0};
    const int w = fmt == 2 ? 6 : 4;
    for (int j = 0; j < 32; ++j) {
        const float sv = std::ldexp(val[j], lg);
        const uint8_t code = mx_code(sv, fmt);
        const double e = std::ldexp((double)mx_value(code, fmt) - (double)sv, -lg);
        *err2 += e * e;
        *ref2 += (double)val[j] * (double)val[j];
        const int bit = j * w;
        bits[bit >> 3] |= (uint8_t)(code << (bit & 7));
        if ((bit & 7) + w > 8) bits[(bit >> 3) + 1] |= (uint8_t)(code >> (8 - (bit & 7)));
    }
    std::memcpy(dst16, bits, 16);
    if (fmt == 2) std::memcpy(dst8, bits + 16, 8);
    *scale = (uint8_t)std::max(0, std::min(254, 127 + exp_bias - lg));
}

// blob of one (32-row block, pair of k-blocks) and its scale bytes: lane (i, g = 0) <- W_lo 2^11, (i, 1) <- W_hi.  c0 = 1 KiB
// fragment (lane * 16), c1 = 512 B (lane * 8; fp6 only), scales = the pair's 256-byte scale block (lane * 4 + gate);
// get(i, k) = fp32 weight of row i of the block, k in [0, 32) of the pair.  Returns the blob's relative RMS quantisation error.
struct BlobErr { double e2[2] = {0, 0}, r2[2] = {0, 0}; };
template <typename Get>
void emit_blob_frag(uint8_t* c0, uint8_t* c1, uint8_t* scales, int gate, int fmt, Get get, BlobErr* be) {
    for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 31, g = lane >> 5;
        float val[32];
        for (int j = 0; j < 32; ++j) {
            const float v = get(i, mx_perm(j));
            const float hi = (float)(_Float16)v;
            val[j] = g ? hi : std::ldexp(v - hi, 11);
        }
        emit_blob(c0 + lane * 16, c1 ? c1 + lane * 8 : nullptr, scales + lane * 4 + gate, val, fmt, g ? 0 : -11, &be->e2[g], &be->r2[g]);
    }
}
inline void emit_hi_frag(_Float16* dst, int kb, const std::function<float(int, int)>& get) {   // lane (i, g) <- hi of k = 16 kb + 8 g + j
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) dst[lane * 8 + j] = (_Float16)get(lane & 31, 16 * kb + 8 * (lane >> 5) + j);
}

// Split-mx weight stream of one layer (byte layouts: ccsm_gru_mx.hip).  Returns the relative RMS quantisation error of the layer's
// correction blobs (the larger of the W_lo and W_hi halves).
float pack_wstream_mx(int layer, int feat0, const float* const wih[2], const float* const whh[2], bool hs3, std::vector<uint8_t>& out, bool dyn = false) {
    const int k_in = layer == 0 ? feat0 : 2 * kHidden;
    const size_t wbytes = layer == 0 ? mx0_wbytes(hs3, dyn) : mx12_wbytes(hs3, dyn);
    const size_t pair_b = mx_pair_b(hs3, dyn);
    out.assign((size_t)2 * kWaves * wbytes, 0);
    // one task per (direction, wave): their byte ranges and error sums are disjoint (0.24 s single-threaded for the three layers)
    BlobErr bes[2 * kWaves];
    std::vector<std::thread> pool;
    for (int dir = 0; dir < 2; ++dir)
        for (int wave = 0; wave < kWaves; ++wave)
            pool.emplace_back([&, dir, wave] {
            BlobErr& be = bes[dir * kWaves + wave];
            uint8_t* base = out.data() + (size_t)(dir * kWaves + wave) * wbytes;
            auto wx = [&](int g) { return [=](int i, int k) -> float { return k < k_in ? wih[dir][(size_t)(g * kHidden + kUnitTile * wave + i) * k_in + k] : 0.f; }; };
            auto wh = [&](int g) { return [=](int i, int k) -> float { return whh[dir][(size_t)(g * kHidden + kUnitTile * wave + i) * kHidden + k]; }; };
            auto hi_at = [&](size_t off, const std::function<float(int, int)>& get, int kb) { emit_hi_frag(reinterpret_cast<_Float16*>(base + off), kb, get); };
            auto lo_at = [&](size_t off, const std::function<float(int, int)>& get, int kb) {      // fp16 residual fragment (layer 0's x-part)
                _Float16* dst = reinterpret_cast<_Float16*>(base + off);
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const float v = get(lane & 31, 16 * kb + 8 * (lane >> 5) + j);
                        dst[lane * 8 + j] = (_Float16)(v - (float)(_Float16)v);
                    }
            };
            auto blob_at = [&](size_t off, long off1, size_t sc_off, int gate_byte, int fmt, const std::function<float(int, int)>& get, int pair) {
                emit_blob_frag(base + off, off1 >= 0 ? base + off1 : nullptr, base + sc_off, gate_byte, fmt, [&](int i, int k) { return get(i, 32 * pair + k); }, &be);
            };
            auto phase_b = [&](size_t off_b) {
                for (int q = 0; q < kKBH / 2; ++q) {
                    const size_t pb = off_b + (size_t)q * pair_b;
                    for (int kbl = 0; kbl < 2; ++kbl)
                        for (int g = 0; g < 3; ++g) {
                            hi_at(pb + (size_t)(3 * kbl + g) * 1024, wh(g), 2 * q + kbl);
                            if (hs3) lo_at(pb + (size_t)(6 + 3 * kbl + g) * 1024, wh(g), 2 * q + kbl);      // hybrid: fp16 residual fragments
                        }
                    if (dyn)          // split-mx-d: fp6 blobs [c0 x 3][c1 x 3][scales]
                        for (int g = 0; g < 3; ++g) blob_at(pb + (size_t)(6 + g) * 1024, (long)(pb + 9 * 1024 + (size_t)g * 512), pb + 10 * 1024 + 512, g, kMxWFmtX, wh(g), q);
                    else if (!hs3)
                        for (int g = 0; g < 3; ++g) blob_at(pb + (size_t)(6 + g) * 1024, -1, pb + 9 * 1024, g, kMxWFmtH, wh(g), q);
                }
            };
            if (layer == 0) {
                for (int g = 0; g < 2; ++g) { hi_at((size_t)(2 * g) * 1024, wx(g), 0); lo_at((size_t)(2 * g + 1) * 1024, wx(g), 0); }
                phase_b(4 * 1024);
                const size_t oc = 4 * 1024 + (size_t)(kKBH / 2) * pair_b;
                hi_at(oc, wx(2), 0); lo_at(oc + 1024, wx(2), 0);
            } else {
                for (int p = 0; p < kKB12 / 2; ++p) {
                    const size_t pa = (size_t)p * kMxPairA;
                    for (int kbl = 0; kbl < 2; ++kbl)
                        for (int g = 0; g < 2; ++g) hi_at(pa + (size_t)(2 * kbl + g) * 1024, wx(g), 2 * p + kbl);
                    for (int g = 0; g < 2; ++g) blob_at(pa + (size_t)(4 + g) * 1024, -1, pa + 6 * 1024, g, kMxWFmtH, wx(g), p);     // r, z: fp4 (see ccsm_gru_mx.hip)
                }
                phase_b(kMx12OffB);
                for (int pp = 0; pp < kKB12 / 2; ++pp) {
                    const size_t pc = mx12_off_c(hs3, dyn) + (size_t)pp * kMxPairC;
                    const int p = kMxZigZag ? kKB12 / 2 - 1 - pp : pp;          // the pair phase C consumes at position pp
                    hi_at(pc, wx(2), 2 * p); hi_at(pc + 1024, wx(2), 2 * p + 1);
                    blob_at(pc + 2 * 1024, (long)(pc + 3 * 1024), pc + 3 * 1024 + 512, 0, kMxWFmtX, wx(2), p);
                }
            }
            });
    for (std::thread& t : pool) t.join();
    BlobErr be;
    for (const BlobErr& b : bes)
        for (int g = 0; g < 2; ++g) { be.e2[g] += b.e2[g]; be.r2[g] += b.r2[g]; }
    const double a = be.r2[0] > 0 ? std::sqrt(be.e2[0] / be.r2[0]) : 0.0, b = be.r2[1] > 0 ? std::sqrt(be.e2[1] / be.r2[1]) : 0.0;
    return (float)std::fmax(a, b);
}

// Weight streams of gru_layer12_f3_kernel (layout in ccsm_gru_f3.hip): fp16 hi and lo fragments of every product, per (direction,
// wave) in consumption order - phase A (r, z input part), phase B (recurrent part, the hybrid's layout), phase C (n input part, zig-zag)
void pack_wstream_f3(const float* const wih[2], const float* const whh[2], std::vector<uint8_t>& out) {
    const int k_in = 2 * kHidden;
    out.assign((size_t)2 * kWaves * kF3WBytes, 0);
    std::vector<std::thread> pool;
    for (int dir = 0; dir < 2; ++dir)
        for (int wave = 0; wave < kWaves; ++wave)
            pool.emplace_back([&, dir, wave] {
            uint8_t* base = out.data() + (size_t)(dir * kWaves + wave) * kF3WBytes;
            auto wx = [&](int g) { return [=](int i, int k) -> float { return wih[dir][(size_t)(g * kHidden + kUnitTile * wave + i) * k_in + k]; }; };
            auto wh = [&](int g) { return [=](int i, int k) -> float { return whh[dir][(size_t)(g * kHidden + kUnitTile * wave + i) * kHidden + k]; }; };
            auto hi_at = [&](size_t off, const std::function<float(int, int)>& get, int kb) { emit_hi_frag(reinterpret_cast<_Float16*>(base + off), kb, get); };
            auto lo_at = [&](size_t off, const std::function<float(int, int)>& get, int kb) {
                _Float16* dst = reinterpret_cast<_Float16*>(base + off);
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const float v = get(lane & 31, 16 * kb + 8 * (lane >> 5) + j);
                        dst[lane * 8 + j] = (_Float16)(v - (float)(_Float16)v);
                    }
            };
            for (int p = 0; p < kKB12 / 2; ++p) {
                const size_t pa = (size_t)p * kF3PairA;
                for (int kbl = 0; kbl < 2; ++kbl)
                    for (int g = 0; g < 2; ++g) {
                        hi_at(pa + (size_t)(2 * kbl + g) * 1024, wx(g), 2 * p + kbl);
                        lo_at(pa + (size_t)(4 + 2 * kbl + g) * 1024, wx(g), 2 * p + kbl);
                    }
            }
            for (int q = 0; q < kKBH / 2; ++q) {
                const size_t pb = (size_t)kF3OffB + (size_t)q * kF3PairB;
                for (int kbl = 0; kbl < 2; ++kbl)
                    for (int g = 0; g < 3; ++g) {
                        hi_at(pb + (size_t)(3 * kbl + g) * 1024, wh(g), 2 * q + kbl);
                        lo_at(pb + (size_t)(6 + 3 * kbl + g) * 1024, wh(g), 2 * q + kbl);
                    }
            }
            for (int pp = 0; pp < kKB12 / 2; ++pp) {
                const size_t pc = (size_t)kF3OffC + (size_t)pp * kF3PairC;
                const int p = kMxZigZag ? kKB12 / 2 - 1 - pp : pp;          // the pair phase C consumes at position pp
                hi_at(pc, wx(2), 2 * p); hi_at(pc + 1024, wx(2), 2 * p + 1);
                lo_at(pc + 2 * 1024, wx(2), 2 * p); lo_at(pc + 3 * 1024, wx(2), 2 * p + 1);
            }
            });
    for (std::thread& t : pool) t.join();
}

// The same streams for gru_layer12_f3s_kernel (ccsm_gru_f3s.hip): fragments of v_mfma_f32_16x16x32_f16 - lane (m, q) <- row m of a 16-unit
// tile, k = 32 pair + 8 q + j - with unit tile T, row m <-> hidden unit 32 wave + 8 (m >> 2) + 4 T + (m & 3); offsets as in pack_wstream_f3
// with the k-block index of a pair replaced by the unit tile
void pack_wstream_f3s(const float* const wih[2], const float* const whh[2], std::vector<uint8_t>& out) {
    const int k_in = 2 * kHidden;
    out.assign((size_t)2 * kWaves * kF3WBytes, 0);
    std::vector<std::thread> pool;
    for (int dir = 0; dir < 2; ++dir)
        for (int wave = 0; wave < kWaves; ++wave)
            pool.emplace_back([&, dir, wave] {
            uint8_t* base = out.data() + (size_t)(dir * kWaves + wave) * kF3WBytes;
            auto unit = [=](int T, int m) { return kUnitTile * wave + 8 * (m >> 2) + 4 * T + (m & 3); };
            auto frag = [&](size_t off_hi, size_t off_lo, const float* w, int ld, int g, int T, int pair) {
                _Float16* hi = reinterpret_cast<_Float16*>(base + off_hi);
                _Float16* lo = reinterpret_cast<_Float16*>(base + off_lo);
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const float v = w[(size_t)(g * kHidden + unit(T, lane & 15)) * ld + 32 * pair + 8 * (lane >> 4) + j];
                        const HalfPair hp = split_host(v);
                        hi[lane * 8 + j] = hp.hi;
                        lo[lane * 8 + j] = hp.lo;
                    }
            };
            for (int p = 0; p < kKB12 / 2; ++p) {
                const size_t pa = (size_t)p * kF3PairA;
                for (int T = 0; T < 2; ++T)
                    for (int g = 0; g < 2; ++g) frag(pa + (size_t)(2 * T + g) * 1024, pa + (size_t)(4 + 2 * T + g) * 1024, wih[dir], k_in, g, T, p);
            }
            for (int q = 0; q < kKBH / 2; ++q) {
                const size_t pb = (size_t)kF3OffB + (size_t)q * kF3PairB;
                for (int T = 0; T < 2; ++T)
                    for (int g = 0; g < 3; ++g) frag(pb + (size_t)(3 * T + g) * 1024, pb + (size_t)(6 + 3 * T + g) * 1024, whh[dir], kHidden, g, T, q);
            }
            for (int pp = 0; pp < kKB12 / 2; ++pp) {
                const size_t pc = (size_t)kF3OffC + (size_t)pp * kF3PairC;
                const int p = kMxZigZag ? kKB12 / 2 - 1 - pp : pp;          // the pair phase C consumes at position pp
                for (int T = 0; T < 2; ++T) frag(pc + (size_t)T * 1024, pc + (size_t)(2 + T) * 1024, wih[dir], k_in, 2, T, p);
            }
            });
    for (std::thread& t : pool) t.join();
}

// Layer 0's stream for gru_layer0_f3s_kernel: x-part fragments A1 = [W_hi | W_hi], A2 = [W_lo | 0] over the one k-block of input columns
// (k >= feat0: zero), the recurrent part as in pack_wstream_f3s
void pack_wstream_f3s0(int feat0, const float* const wih[2], const float* const whh[2], std::vector<uint8_t>& out) {
    out.assign((size_t)2 * kWaves * kF3s0WBytes, 0);
    for (int dir = 0; dir < 2; ++dir)
        for (int wave = 0; wave < kWaves; ++wave) {
            uint8_t* base = out.data() + (size_t)(dir * kWaves + wave) * kF3s0WBytes;
            auto unit = [=](int T, int m) { return kUnitTile * wave + 8 * (m >> 2) + 4 * T + (m & 3); };
            auto xfrag = [&](size_t off, int g, int T) {          // A1 at off, A2 at off + 1 KiB
                _Float16* a1 = reinterpret_cast<_Float16*>(base + off);
                _Float16* a2 = reinterpret_cast<_Float16*>(base + off + 1024);
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int q = lane >> 4, k = 8 * (q & 1) + j;
                        const float v = k < feat0 ? wih[dir][(size_t)(g * kHidden + unit(T, lane & 15)) * feat0 + k] : 0.f;
                        const HalfPair hp = split_host(v);
                        a1[lane * 8 + j] = hp.hi;
                        a2[lane * 8 + j] = (q >> 1) ? (_Float16)0.f : hp.lo;
                    }
            };
            for (int T = 0; T < 2; ++T) {
                for (int g = 0; g < 2; ++g) xfrag((size_t)(4 * T + 2 * g) * 1024, g, T);
                xfrag((size_t)kF3s0OffC + (size_t)(2 * T) * 1024, 2, T);
            }
            for (int q = 0; q < kKBH / 2; ++q) {
                const size_t pb = (size_t)kF3s0OffB + (size_t)q * kF3PairB;
                for (int T = 0; T < 2; ++T)
                    for (int g = 0; g < 3; ++g) {
                        _Float16* hi = reinterpret_cast<_Float16*>(base + pb + (size_t)(3 * T + g) * 1024);
                        _Float16* lo = reinterpret_cast<_Float16*>(base + pb + (size_t)(6 + 3 * T + g) * 1024);
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const HalfPair hp = split_host(whh[dir][(size_t)(g * kHidden + unit(T, lane & 15)) * kHidden + 32 * q + 8 * (lane >> 4) + j]);
                                hi[lane * 8 + j] = hp.hi;
                                lo[lane * 8 + j] = hp.lo;
                            }
                    }
            }
        }
}

// Layers 1-2 in split-mx for gru_layer12_mx16_kernel (layout in ccsm_gru_mx16.hip): per DOUBLE pair the hi fragments of both pairs
// (v_mfma_f32_16x16x32_f16: lane (m, q) <- unit-tile row m, k = 32 pair + 8 q + j) and the MX blobs of the scaled instruction, whose lane (m, q)
// holds the 32 values (kMxPerm order) of unit-tile row m of the double pair's pair q >> 1, term q & 1 (0: W_lo 2^11, 1: W_hi), one E8M0 scale byte
// per lane and blob; unit tile T, row m <-> hidden unit 32 wave + 8 (m >> 2) + 4 T + (m & 3) (pack_wstream_f3s).  Formats as pack_wstream_mx:
// fp4 for the r, z input part and the recurrent part, fp6 for the n gate's input part, whose pairs come in phase C's zig-zag order.
float pack_wstream_mx16(const float* const wih[2], const float* const whh[2], std::vector<uint8_t>& out) {
    const int k_in = 2 * kHidden;
    out.assign((size_t)2 * kWaves * kMx16WBytes, 0);
    BlobErr bes[2 * kWaves];
    std::vector<std::thread> pool;
    for (int dir = 0; dir < 2; ++dir)
        for (int wave = 0; wave < kWaves; ++wave)
            pool.emplace_back([&, dir, wave] {
            BlobErr& be = bes[dir * kWaves + wave];
            uint8_t* base = out.data() + (size_t)(dir * kWaves + wave) * kMx16WBytes;
            auto unit = [=](int T, int m) { return kUnitTile * wave + 8 * (m >> 2) + 4 * T + (m & 3); };
            auto hi_frag = [&](size_t off, const float* w, int ld, int g, int T, int pair) {
                _Float16* hi = reinterpret_cast<_Float16*>(base + off);
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) hi[lane * 8 + j] = (_Float16)w[(size_t)(g * kHidden + unit(T, lane & 15)) * ld + 32 * pair + 8 * (lane >> 4) + j];
            };
            // pair_of(0 | 1) = the pair the lanes q >> 1 = 0 | 1 take; off8 < 0: fp4 (16 bytes per lane), else bytes 16-23 of the fp6 blob there
            auto blob = [&](size_t off16, long off8, size_t sc_off, int sc_byte, int fmt, const float* w, int ld, int g, int T, int pair0, int pair1) {
                for (int lane = 0; lane < 64; ++lane) {
                    const int m = lane & 15, q = lane >> 4, term = q & 1, pair = (q >> 1) ? pair1 : pair0;
                    float val[32];
                    for (int j = 0; j < 32; ++j) {
                        const float v = w[(size_t)(g * kHidden + unit(T, m)) * ld + 32 * pair + mx_perm(j)];
                        const float h = (float)(_Float16)v;
                        val[j] = term ? h : std::ldexp(v - h, 11);
                    }
                    emit_blob(base + off16 + lane * 16, off8 >= 0 ? base + off8 + lane * 8 : nullptr, base + sc_off + lane * 4 + sc_byte, val, fmt, term ? 0 : -11,
                              &be.e2[term], &be.r2[term]);
                }
            };
            for (int DA = 0; DA < kKB12 / 4; ++DA) {
                const size_t pa = (size_t)DA * kMx16PA;
                for (int T = 0; T < 2; ++T)
                    for (int g = 0; g < 2; ++g) {
                        hi_frag(pa + (size_t)(2 * T + g) * 1024, wih[dir], k_in, g, T, 2 * DA);
                        hi_frag(pa + (size_t)(4 + 2 * T + g) * 1024, wih[dir], k_in, g, T, 2 * DA + 1);
                        blob(pa + (size_t)(8 + 2 * T + g) * 1024, -1, pa + 12 * 1024, 2 * T + g, kMxWFmtH, wih[dir], k_in, g, T, 2 * DA, 2 * DA + 1);
                    }
            }
            for (int D = 0; D < kKBH / 4; ++D) {
                const size_t pb = (size_t)kMx16OffB + (size_t)D * kMx16PB;
                for (int T = 0; T < 2; ++T)
                    for (int g = 0; g < 3; ++g) {
                        hi_frag(pb + (size_t)(3 * T + g) * 1024, whh[dir], kHidden, g, T, 2 * D);
                        hi_frag(pb + (size_t)(6 + 3 * T + g) * 1024, whh[dir], kHidden, g, T, 2 * D + 1);
                        blob(pb + (size_t)(12 + 3 * T + g) * 1024, -1, pb + 18 * 1024 + (size_t)T * 256, g, kMxWFmtH, whh[dir], kHidden, g, T, 2 * D, 2 * D + 1);
                    }
            }
            for (int DC = 0; DC < kKB12 / 4; ++DC) {
                const size_t pc = (size_t)kMx16OffC + (size_t)DC * kMx16PC;
                auto pair_at = [](int pp) { return kMxZigZag ? kKB12 / 2 - 1 - pp : pp; };      // the pair phase C consumes at position pp
                for (int T = 0; T < 2; ++T) {
                    hi_frag(pc + (size_t)T * 1024, wih[dir], k_in, 2, T, pair_at(2 * DC));
                    hi_frag(pc + (size_t)(2 + T) * 1024, wih[dir], k_in, 2, T, pair_at(2 * DC + 1));
                    blob(pc + (size_t)(4 + T) * 1024, (long)(pc + 6 * 1024 + (size_t)T * 512), pc + 7 * 1024, T, kMxWFmtX, wih[dir], k_in, 2, T, pair_at(2 * DC), pair_at(2 * DC + 1));
                }
            }
            });
    for (std::thread& t : pool) t.join();
    BlobErr be;
    for (const BlobErr& b : bes)
        for (int g = 0; g < 2; ++g) { be.e2[g] += b.e2[g]; be.r2[g] += b.r2[g]; }
    const double a = be.r2[0] > 0 ? std::sqrt(be.e2[0] / be.r2[0]) : 0.0, b = be.r2[1] > 0 ? std::sqrt(be.e2[1] / be.r2[1]) : 0.0;
    return (float)std::fmax(a, b);
}

// biases in natural unit order for the 16x16x32 kernels: [dir][wave][set r, z, n_x, n_h][32]
void pack_bias_natural(const float* const bih[2], const float* const bhh[2], std::vector<float>& out) {
    out.assign((size_t)2 * kWaves * 4 * 32, 0.f);
    for (int dir = 0; dir < 2; ++dir)
        for (int wave = 0; wave < kWaves; ++wave)
            for (int i = 0; i < 32; ++i) {
                const int u = kUnitTile * wave + i;
                float* o = &out[(size_t)(dir * kWaves + wave) * 4 * 32];
                o[0 * 32 + i] = bih[dir][u] + bhh[dir][u];
                o[1 * 32 + i] = bih[dir][kHidden + u] + bhh[dir][kHidden + u];
                o[2 * 32 + i] = bih[dir][2 * kHidden + u];
                o[3 * 32 + i] = bhh[dir][2 * kHidden + u];
            }
}

void pack_bias(const float* const bih[2], const float* const bhh[2], std::vector<float>& out) {
    out.assign((size_t)2 * kWaves * 4 * 32, 0.f);
    for (int dir = 0; dir < 2; ++dir)
        for (int wave = 0; wave < kWaves; ++wave)
            for (int set = 0; set < 4; ++set)
                for (int hh = 0; hh < 2; ++hh)
                    for (int r = 0; r < 16; ++r) {
                        const int u = kUnitTile * wave + crow(r, hh);
                        float v;
                        if (set == 0) v = bih[dir][u] + bhh[dir][u];
                        else if (set == 1) v = bih[dir][kHidden + u] + bhh[dir][kHidden + u];
                        else if (set == 2) v = bih[dir][2 * kHidden + u];
                        else v = bhh[dir][2 * kHidden + u];
                        out[(((size_t)(dir * kWaves + wave) * 4 + set) * 2 + hh) * 16 + r] = v;
                    }
}

// Attention projection (256 x 512) as A fragments [wave][kb 32][hl][64][8]
void pack_att(const float* w, std::vector<_Float16>& out) {
    out.assign((size_t)kWaves * kKB12 * 2 * 512, (_Float16)0.f);
    for (int wave = 0; wave < kWaves; ++wave)
        for (int kb = 0; kb < kKB12; ++kb)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, q = lane >> 5;
                const size_t base = (((size_t)wave * kKB12 + kb) * 2) * 512 + lane * 8;
                for (int j = 0; j < 8; ++j) {
                    const HalfPair p = split_host(w[(size_t)(kUnitTile * wave + i) * 2 * kHidden + 16 * kb + 8 * q + j]);
                    out[base + j] = p.hi;
                    out[base + 512 + j] = p.lo;
                }
            }
}

// Split-f8 attention projection: pack_att's hi fragments with the second fragment of every k-block = its corr fragment
void pack_att_v3(const float* w, std::vector<_Float16>& out, int& scale) {
    pack_att(w, out);
    const int lg = corr_scale_log2(w, (size_t)kAttHidden * 2 * kHidden);
    scale = 127 - 11 - lg;
    uint8_t* bytes = reinterpret_cast<uint8_t*>(out.data());
    for (int wave = 0; wave < kWaves; ++wave)
        for (int kb = 0; kb < kKB12; ++kb)
            emit_corr_frag(bytes + (((size_t)wave * kKB12 + kb) * 2 + 1) * 1024, kb, lg,
                           [&](int i, int k) { return w[(size_t)(kUnitTile * wave + i) * 2 * kHidden + k]; });
}

// fc1.weight (2, 1024) for the attention pool's MFMA: four virtual units u = class * 2 + strand half, W4[u][k] = fc1.weight[class][512 strand + k],
// as the A operand of the same split-f8 product as Ua - rows 4..31 of the 32-row tile are zero and not stored: per fragment (k-block kb,
// hi | corr) only lanes (i < 4, q) = 8 x 16 bytes, [kb 32][hi|corr][q 2][i 4] = 8 KiB, + one zero line for the other lanes.
void pack_fc_v3(const float* fcw, std::vector<uint8_t>& out, int& scale) {
    std::vector<float> w4((size_t)4 * 2 * kHidden);
    for (int cls = 0; cls < kClasses; ++cls)
        for (int sh = 0; sh < 2; ++sh)
            for (int k = 0; k < 2 * kHidden; ++k) w4[(size_t)(cls * 2 + sh) * 2 * kHidden + k] = fcw[(size_t)cls * 4 * kHidden + sh * 2 * kHidden + k];
    const int lg = corr_scale_log2(w4.data(), w4.size());
    scale = 127 - 11 - lg;
    out.assign((size_t)kKB12 * 2 * 8 * 16 + 16, 0);
    std::vector<uint8_t> frag(1024);
    for (int kb = 0; kb < kKB12; ++kb) {
        for (int q = 0; q < 2; ++q)
            for (int i = 0; i < 4; ++i) {
                _Float16 h[8];
                for (int j = 0; j < 8; ++j) h[j] = split_host(w4[(size_t)i * 2 * kHidden + 16 * kb + 8 * q + j]).hi;
                std::memcpy(out.data() + ((size_t)(kb * 2 + 0) * 8 + q * 4 + i) * 16, h, 16);
            }
        emit_corr_frag(frag.data(), kb, lg, [&](int i, int k) { return i < 4 ? w4[(size_t)i * 2 * kHidden + k] : 0.0f; });
        for (int q = 0; q < 2; ++q)
            for (int i = 0; i < 4; ++i) std::memcpy(out.data() + ((size_t)(kb * 2 + 1) * 8 + q * 4 + i) * 16, frag.data() + (q * 32 + i) * 16, 16);
    }
}

// The fc1 tile of pack_fc_v3 for the three-pass pool: [kb 32][hi|lo][q 2][i 4] x 16 bytes + one zero line
void pack_fc_s3(const float* fcw, std::vector<uint8_t>& out) {
    out.assign((size_t)kKB12 * 2 * 8 * 16 + 16, 0);
    for (int kb = 0; kb < kKB12; ++kb)
        for (int q = 0; q < 2; ++q)
            for (int i = 0; i < 4; ++i) {
                _Float16 h[8], l[8];
                for (int j = 0; j < 8; ++j) {
                    const HalfPair p = split_host(fcw[(size_t)(i >> 1) * 4 * kHidden + (i & 1) * 2 * kHidden + 16 * kb + 8 * q + j]);
                    h[j] = p.hi; l[j] = p.lo;
                }
                std::memcpy(out.data() + ((size_t)(kb * 2 + 0) * 8 + q * 4 + i) * 16, h, 16);
                std::memcpy(out.data() + ((size_t)(kb * 2 + 1) * 8 + q * 4 + i) * 16, l, 16);
            }
}

template <typename T>
ccsm_status upload(T** dst, const void* src, size_t bytes) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(dst), bytes));
    HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return CCSM_OK;
}

struct SiteKeys {             // device arrays (per site of the slice) naming the sites' random streams, or NULL
    const unsigned long long* key = nullptr;
    const unsigned int* sub = nullptr;
};

// Cheap per-slice kernels: initial states and layer-0 input fragments of rows [row_base, row_base + 2 n_sites).
ccsm_status launch_prep(const ccsm_model* m, ccsm_workspace* ws, int n_sites, int row_base, const StrandDev& s1,
                        const StrandDev& s2, int kmer_is_f32, int npass_per_base, int h0_mode, const float* h0a,
                        const float* h0b, uint64_t seed, uint64_t offset, hipStream_t st, SiteKeys sk = SiteKeys()) {
    const size_t total4 = (size_t)2 * kLayers * 2 * n_sites * (kHidden / 4);
    const int grid = (int)std::min<size_t>((total4 + 255) / 256, 4096);
    hipLaunchKernelGGL(prep_h0_kernel, dim3(grid), dim3(256), 0, st, ws->h0buf, h0a, h0b, n_sites, row_base, ws->rows_p, h0_mode,
                       seed, offset, sk.key, sk.sub);
    const int total = 2 * n_sites * kSeqLen * 2;
    hipLaunchKernelGGL(pack_x0_kernel, dim3((total + 255) / 256), dim3(256), 0, st, ws->x0, s1, s2, m->embed, n_sites, row_base,
                       kmer_is_f32, npass_per_base, m->feat, m->fold);
    HIP_TRY(hipGetLastError());
    return CCSM_OK;
}

// GRU layers in split-mx arithmetic.  A build with -DCCSM_PHASE_STAMPS also holds the instantiations that record the cycle counter
// at the phase boundaries of workgroup 0 (tools/gpu_phases.py); the product library is built without it.
template <bool HS3, bool DYN, int NB = kMxNB>
void launch_gru_mx(int layer, dim3 grid, hipStream_t st, const uint4* xin, uint4* out, const uint4* wst, const float* bias, const float* h0,
                   int rows_p, unsigned long long* dbg, bool stag = false) {
#ifdef CCSM_PHASE_STAMPS
    if (dbg && NB == kMxNB) {
        if (layer == 0 && stag) hipLaunchKernelGGL((gru_layer0_mx_kernel<true, HS3, DYN, kMxNB, false, true>), grid, dim3(512), kMx0Lds, st, xin, out, wst, bias, h0, rows_p, dbg);
        else if (layer == 0) hipLaunchKernelGGL((gru_layer0_mx_kernel<true, HS3, DYN>), grid, dim3(512), kMx0Lds, st, xin, out, wst, bias, h0, rows_p, dbg);
        else if (layer == 1) hipLaunchKernelGGL((gru_layer12_mx_kernel<false, true, HS3, DYN>), grid, dim3(512), kMx12Lds, st, xin, out, wst, bias, h0, rows_p, dbg);
        else hipLaunchKernelGGL((gru_layer12_mx_kernel<true, true, HS3, DYN>), grid, dim3(512), kMx12Lds, st, xin, out, wst, bias, h0, rows_p, dbg);
        return;
    }
#endif
    (void)dbg;
    if constexpr (NB == kMxNB) {
        if (layer == 0 && stag) {
            hipLaunchKernelGGL((gru_layer0_mx_kernel<false, HS3, DYN, NB, false, true>), grid, dim3(512), mx0_lds(NB), st, xin, out, wst, bias, h0, rows_p, nullptr);
            return;
        }
    }
    if (layer == 0) hipLaunchKernelGGL((gru_layer0_mx_kernel<false, HS3, DYN, NB>), grid, dim3(512), mx0_lds(NB), st, xin, out, wst, bias, h0, rows_p, nullptr);
    else if (layer == 1) hipLaunchKernelGGL((gru_layer12_mx_kernel<false, false, HS3, DYN, NB>), grid, dim3(512), mx12_lds(NB), st, xin, out, wst, bias, h0, rows_p, nullptr);
    else hipLaunchKernelGGL((gru_layer12_mx_kernel<true, false, HS3, DYN, NB>), grid, dim3(512), mx12_lds(NB), st, xin, out, wst, bias, h0, rows_p, nullptr);   // fp8 corr fragments for the attention kernel
}

// Heavy kernels, once over every row used by the current slices, then the per-slice logits/softmax.
// F8: the split-mx family (activations as [hi | blob] fragments); HS3: its hybrid member (recurrent part in three fp16 passes);
// DYN: split-mx-d (fp6 recurrent blobs, per-row dynamic activation scales)
template <bool F8, bool HS3 = false, bool DYN = false>
ccsm_status launch_run(const ccsm_model* m, ccsm_workspace* ws, hipStream_t st) {
    // Workgroups of 96 rows (kMxNB = 3 tiles) amortise the weight stream best; where they would leave compute units idle - a lone
    // batch (86 workgroups), a ragged group of two (172) - the split-mx family runs its 64-row or 32-row form instead: a workgroup's
    // step takes ~0.74 / ~0.55 of the time and more of the 256 CUs have one.  Picked by rounds of 256 workgroups; CCSM_WG_TILES = 1 | 2 | 3
    // forces a form (A/B runs and tests; read per launch).
    int nb_run = 3;
    if (F8 || !m->split3_v2) {
        static const double kStepCost[4] = {0.0, 0.55, 0.74, 1.0};
        double best = 1e30;
        for (int nb = 3; nb >= 1; --nb) {
            const int wgs = 2 * ((ws->rows_used + 32 * nb - 1) / (32 * nb));
            const double cost = kStepCost[nb] * ((wgs + 255) / 256);
            if (cost < best - 1e-9) { best = cost; nb_run = nb; }
        }
        const char* force = std::getenv("CCSM_WG_TILES");
        if (force && force[0] >= '1' && force[0] <= '3') nb_run = force[0] - '0';
    }
    const int wg_rows = 32 * nb_run;
    const int rows_run = ((ws->rows_used + wg_rows - 1) / wg_rows) * wg_rows;
    const int tiles = rows_run / 32;
    const bool tm = ws->timing && ws->ev_ok;
    if (tm) ws->ev = ws->evs[ws->ev_runs % ccsm_workspace::kEvSets];
    if (tm) HIP_TRY(hipEventRecord(ws->ev[1], st));
    const size_t slab = (size_t)2 * ws->rows_p * kHidden;  // floats per layer (two directions)
    const dim3 ggrid(2 * (rows_run / wg_rows));
#ifdef CCSM_PHASE_STAMPS
    static const int dbg_layer = std::getenv("CCSM_PHASE_LAYER") ? std::atoi(std::getenv("CCSM_PHASE_LAYER")) : 1;
#else
    constexpr int dbg_layer = -1;
#endif
    if constexpr (F8) {
        uint4* const* wst = HS3 ? m->wsthy : DYN ? m->wstmd : m->wstmx;
        auto layer = [&](int l, const uint4* in, uint4* out_, unsigned long long* dbg) {
            if constexpr (!HS3 && !DYN) {
                if (l >= 1 && m->wstmx16[l]) {                  // plain split-mx: layers 1-2 on the 16-wide instructions (ccsm_gru_mx16.hip)
                    auto go = [&](auto nbc, auto f8c) {
                        constexpr int NBX = decltype(nbc)::value;
                        constexpr bool F8O = decltype(f8c)::value;
#ifdef CCSM_PHASE_STAMPS
                        if (dbg && NBX == 3) {
                            hipLaunchKernelGGL((gru_layer12_mx16_kernel<F8O, 3, true>), ggrid, dim3(512), mx12_lds(3), st, in, out_, m->wstmx16[l], m->biasn[l], ws->h0buf + l * slab, ws->rows_p, dbg);
                            return;
                        }
#endif
                        hipLaunchKernelGGL((gru_layer12_mx16_kernel<F8O, NBX>), ggrid, dim3(512), mx12_lds(NBX), st, in, out_, m->wstmx16[l], m->biasn[l], ws->h0buf + l * slab, ws->rows_p, nullptr);
                    };
                    auto by_nb = [&](auto f8c) {
                        if (nb_run == 1) go(std::integral_constant<int, 1>{}, f8c);
                        else if (nb_run == 2) go(std::integral_constant<int, 2>{}, f8c);
                        else go(std::integral_constant<int, 3>{}, f8c);
                    };
                    if (l == 2) by_nb(std::true_type{}); else by_nb(std::false_type{});
                    return;
                }
            }
            if (nb_run == 1) launch_gru_mx<HS3, DYN, 1>(l, ggrid, st, in, out_, wst[l], m->bias[l], ws->h0buf + l * slab, ws->rows_p, nullptr);
            else if (nb_run == 2) launch_gru_mx<HS3, DYN, 2>(l, ggrid, st, in, out_, wst[l], m->bias[l], ws->h0buf + l * slab, ws->rows_p, nullptr);
            else launch_gru_mx<HS3, DYN>(l, ggrid, st, in, out_, wst[l], m->bias[l], ws->h0buf + l * slab, ws->rows_p, dbg, m->l0_stag);
        };
        layer(0, ws->x0, ws->act[0], dbg_layer == 0 ? ws->dbg : nullptr);
        if (tm) HIP_TRY(hipEventRecord(ws->ev[2], st));
#ifdef CCSM_PWR_L0ONLY          // power diagnostics build (tools/gpu_power.py): layer 0 on its own - results are wrong on purpose
        layer(0, ws->x0, ws->act[1], nullptr);
        if (tm) HIP_TRY(hipEventRecord(ws->ev[3], st));
        layer(0, ws->x0, ws->act[1], nullptr);
#else
        layer(1, ws->act[0], ws->act[1], dbg_layer == 1 ? ws->dbg : nullptr);
        if (tm) HIP_TRY(hipEventRecord(ws->ev[3], st));
        layer(2, ws->act[1], ws->act[0], dbg_layer == 2 ? ws->dbg : nullptr);
#endif
    } else if (m->split3_v2) {                                      // CCSM_SPLIT3_V2=1: the round-1 kernel (A/B runs)
        hipLaunchKernelGGL((gru_layer_v2_kernel<kKB0>), ggrid, dim3(512), gru2_lds(kKB0), st, ws->x0, ws->act[0],
                           m->wst2[0], m->bias[0], ws->h0buf, ws->rows_p);
        if (tm) HIP_TRY(hipEventRecord(ws->ev[2], st));
        hipLaunchKernelGGL((gru_layer_v2_kernel<kKB12>), ggrid, dim3(512), gru2_lds(kKB12), st, ws->act[0], ws->act[1],
                           m->wst2[1], m->bias[1], ws->h0buf + slab, ws->rows_p);
        if (tm) HIP_TRY(hipEventRecord(ws->ev[3], st));
        hipLaunchKernelGGL((gru_layer_v2_kernel<kKB12>), ggrid, dim3(512), gru2_lds(kKB12), st, ws->act[1], ws->act[0],
                           m->wst2[2], m->bias[2], ws->h0buf + 2 * slab, ws->rows_p);
    } else {
        // split3 on the split-mx schedule: layer 0 = the hybrid's layer-0 kernel (three passes in both parts) writing fp16 lo fragments,
        // layers 1-2 = gru_layer12_f3_kernel (ccsm_gru_f3.hip)
        auto layer = [&](auto nbc, int l, const uint4* in, uint4* out_) {
            constexpr int NBF = decltype(nbc)::value;
            if (l == 0 && !m->f3_shape32)
                hipLaunchKernelGGL((gru_layer0_f3s_kernel<NBF>), ggrid, dim3(512), mx0_lds(NBF), st, in, out_, m->wstf3s[0], m->biasn[0], ws->h0buf, ws->rows_p);
            else if (l == 0 && NBF == 3 && m->l0_stag)
                hipLaunchKernelGGL((gru_layer0_mx_kernel<false, true, false, 3, true, true>), ggrid, dim3(512), mx0_lds(3), st, in, out_, m->wstf3[0], m->bias[0],
                                   ws->h0buf, ws->rows_p, nullptr);
            else if (l == 0) hipLaunchKernelGGL((gru_layer0_mx_kernel<false, true, false, NBF, true>), ggrid, dim3(512), mx0_lds(NBF), st, in, out_, m->wstf3[0], m->bias[0],
                                           ws->h0buf, ws->rows_p, nullptr);
#ifdef CCSM_PHASE_STAMPS
            else if (NBF == 3 && ws->dbg && l == (dbg_layer == 2 ? 2 : 1))
                {
                    if (m->f3_shape32) hipLaunchKernelGGL((gru_layer12_f3_kernel<3, true>), ggrid, dim3(512), f3_lds(3), st, in, out_, m->wstf3[l], m->bias[l], ws->h0buf + l * slab, ws->rows_p, ws->dbg);
                    else hipLaunchKernelGGL((gru_layer12_f3s_kernel<3, true>), ggrid, dim3(512), f3_lds(3), st, in, out_, m->wstf3s[l], m->biasn[l], ws->h0buf + l * slab, ws->rows_p, ws->dbg);
                }
#endif
            else if (!m->f3_shape32)
                hipLaunchKernelGGL((gru_layer12_f3s_kernel<NBF>), ggrid, dim3(512), f3_lds(NBF), st, in, out_, m->wstf3s[l], m->biasn[l], ws->h0buf + l * slab, ws->rows_p, nullptr);
            else hipLaunchKernelGGL((gru_layer12_f3_kernel<NBF>), ggrid, dim3(512), f3_lds(NBF), st, in, out_, m->wstf3[l], m->bias[l], ws->h0buf + l * slab, ws->rows_p, nullptr);
        };
        auto all = [&](auto nbc) -> ccsm_status {
            layer(nbc, 0, ws->x0, ws->act[0]);
            if (tm) HIP_TRY(hipEventRecord(ws->ev[2], st));
            layer(nbc, 1, ws->act[0], ws->act[1]);
            if (tm) HIP_TRY(hipEventRecord(ws->ev[3], st));
            layer(nbc, 2, ws->act[1], ws->act[0]);
            return CCSM_OK;
        };
        const ccsm_status rc = nb_run == 1 ? all(std::integral_constant<int, 1>{}) : nb_run == 2 ? all(std::integral_constant<int, 2>{}) : all(std::integral_constant<int, 3>{});
        if (rc != CCSM_OK) return rc;
    }
    if (tm) HIP_TRY(hipEventRecord(ws->ev[4], st));
    SliceTable tab;
    tab.count = ws->n_slices;
    for (int i = 0; i < kMaxSlices; ++i) {
        tab.row_base[i] = i < ws->n_slices ? ws->slice_row[i] : 0;
        tab.n_sites[i] = i < ws->n_slices ? ws->slice_n[i] : 0;
    }
    if constexpr (F8 && !HS3)       // (the hybrid's last layer writes fp16 hi + lo fragments: the three-pass attention pool)
        hipLaunchKernelGGL(attn_fc_f8_kernel, dim3(tiles), dim3(512), kAttF8Lds, st, ws->act[0], m->wa3, m->ua3, m->va, m->fc3,
                           ws->part, tab, m->att_scale[0], m->att_scale[1], m->fc_scale);
    else
        hipLaunchKernelGGL(attn_fc_kernel, dim3(tiles), dim3(512), kAttLds, st, ws->act[0], m->wa, m->ua, m->va, m->fc3s,
                           ws->part, tab);
    if (tm) HIP_TRY(hipEventRecord(ws->ev[5], st));
    for (int i = 0; i < ws->n_slices; ++i)
        hipLaunchKernelGGL(finalize_kernel, dim3((ws->slice_n[i] + 255) / 256), dim3(256), 0, st, ws->part, m->fcb,
                           ws->slice_logits[i], ws->slice_probs[i], ws->slice_n[i], ws->slice_row[i]);
    if (tm) {
        HIP_TRY(hipEventRecord(ws->ev[6], st));
        ws->timed = true;
        ws->ev_runs++;
    }
    ws->n_slices = 0;
    ws->rows_used = 0;
    HIP_TRY(hipGetLastError());
    return CCSM_OK;
}

ccsm_status dispatch_run(const ccsm_model* m, ccsm_workspace* ws, hipStream_t st) {
    if (ws->force_split3) {
        ws->force_split3 = false;
        return launch_run<false>(m, ws, st);
    }
    switch (m->precision) {
        case CCSM_PRECISION_SPLIT_F8: return launch_run<true>(m, ws, st);
        case CCSM_PRECISION_HYBRID: return launch_run<true, true>(m, ws, st);
        case CCSM_PRECISION_SPLIT_MXD: return launch_run<true, false, true>(m, ws, st);
        case CCSM_PRECISION_SPLIT3: return launch_run<false>(m, ws, st);
    }
    return fail(CCSM_ERR_UNSUPPORTED, "precision must be 3 (split-fp16), 4 (split-mx), 5 (hybrid) or 6 (split-mx-d)");
}

// add one slice (device pointers) to the workspace
ccsm_status add_slice(const ccsm_model* m, ccsm_workspace* ws, int n_sites, const StrandDev& s1, const StrandDev& s2,
                      int kmer_is_f32, int npass_per_base, int h0_mode, const float* h0a, const float* h0b, uint64_t seed,
                      uint64_t offset, float* logits, float* probs, hipStream_t st, SiteKeys sk = SiteKeys()) {
    if (ws->n_slices >= kMaxSlices) return fail(CCSM_ERR_CAPACITY, "too many slices in one group (max 16)");
    if (ws->rows_used + 2 * n_sites > 2 * ws->max_sites) return fail(CCSM_ERR_CAPACITY, "group exceeds the workspace's max_sites");
    const int row_base = ws->rows_used;
    ccsm_status rc = launch_prep(m, ws, n_sites, row_base, s1, s2, kmer_is_f32, npass_per_base, h0_mode, h0a, h0b, seed, offset, st, sk);
    if (rc != CCSM_OK) return rc;
    const int i = ws->n_slices++;
    ws->slice_row[i] = row_base;
    ws->slice_n[i] = n_sites;
    ws->slice_logits[i] = logits;
    ws->slice_probs[i] = probs;
    ws->rows_used += 2 * n_sites;
    return CCSM_OK;
}

ccsm_status dispatch_forward(const ccsm_model* m, ccsm_workspace* ws, int n_sites, const StrandDev& s1, const StrandDev& s2,
                             int kmer_is_f32, int npass_per_base, int h0_mode, const float* h0a, const float* h0b,
                             uint64_t seed, uint64_t offset, float* logits, float* probs, hipStream_t st, SiteKeys sk = SiteKeys()) {
    ws->n_slices = 0;
    ws->rows_used = 0;
    ccsm_status rc = add_slice(m, ws, n_sites, s1, s2, kmer_is_f32, npass_per_base, h0_mode, h0a, h0b, seed, offset, logits, probs, st, sk);
    if (rc != CCSM_OK) return rc;
    return dispatch_run(m, ws, st);
}

StrandDev strand_dev(const ccsm_strand& t) {
    StrandDev d{t.kmer, t.ipd, t.pw, t.npass};
    d.ipd_std = t.ipd_std; d.pw_std = t.pw_std; d.sn = t.sn; d.map = t.map;
    return d;
}

ccsm_status check_call(const ccsm_model* m, const ccsm_workspace* ws, int n_sites, const ccsm_batch* b, const ccsm_h0* h0) {
    if (!m || !ws || !b) return fail(CCSM_ERR_INVALID_ARG, "model, workspace and batch must be non-NULL");
    if (n_sites <= 0) return fail(CCSM_ERR_INVALID_ARG, "n_sites must be > 0");
    if (n_sites > ws->max_sites) return fail(CCSM_ERR_CAPACITY, "n_sites exceeds the workspace's max_sites");
    if (ws->device != m->device) return fail(CCSM_ERR_INVALID_ARG, "workspace and model live on different devices");
    for (int s = 0; s < 2; ++s) {
        const ccsm_strand& t = b->strand[s];
        if (!t.kmer || !t.ipd || !t.pw || ((m->feat & kFeatNpass) && !t.npass))
            return fail(CCSM_ERR_INVALID_ARG, "batch strand pointers must be non-NULL");
        if (((m->feat & kFeatStds) && (!t.ipd_std || !t.pw_std)) || ((m->feat & kFeatSn) && !t.sn) || ((m->feat & kFeatMap) && !t.map))
            return fail(CCSM_ERR_INVALID_ARG, "the model was created with is_stds / is_sn / is_map: the batch must carry ipd_std + pw_std / sn / map");
    }
    if (h0) {
        if (h0->mode < 0 || h0->mode > 2) return fail(CCSM_ERR_INVALID_ARG, "unknown h0 mode");
        if (h0->mode == CCSM_H0_EXPLICIT && (!h0->h0[0] || !h0->h0[1]))
            return fail(CCSM_ERR_INVALID_ARG, "explicit h0 needs both strand tensors");
    }
    return CCSM_OK;
}

}  // namespace

extern "C" {

const char* ccsm_last_error(void) { return g_err.c_str(); }
const char* ccsm_version(void) { return "libccsm 0.1.0 (gfx950)"; }
int ccsm_model_precision(const ccsm_model* m) { return m ? m->precision : 0; }
size_t ccsm_workspace_bytes(const ccsm_workspace* ws) { return ws ? ws->bytes : 0; }

}  // extern "C"

namespace {
// The default arithmetic (`precision 0`) is chosen by MEASUREMENT, and conservatively.  profiles/r04_a_tail_study.log (2^20 sites per
// checkpoint, six checkpoints trained with libccsm_train + the synthetic initialisation) shows what decides it: on TRAINED weights the
// per-site error of every block-scaled arithmetic (split-mx, split-mx-d, the hybrid) is heavy-tailed - the hybrid, the most accurate of
// them, still leaves 1-7547 of 2^20 sites beyond the 1e-4 bar on four of six checkpoints, and a batch of 8192 sites inside 2.5e-5 says
// little about the millionth site - while on the synthetic initialisation (the benchmark's weights) split-mx is light-tailed: max 8.5e-6
// over 2^20 sites, 1.6 x its 99.9th percentile.  So ccsm_create serves split-mx only where a probe finds exactly that picture, and the
// fp32-class split-fp16 arithmetic (max 6e-7 against the oracle on every checkpoint) otherwise; split-mx-d and the hybrid are never
// chosen by the probe: they are the caller's explicit choice (`call_mods --arithmetic`), with their tails documented.
//   probe: kProbeBatches batches of kProbeBatch synthetic sites (z-scores with outliers, device-drawn initial states) through split-fp16
//   and split-mx; split-mx is accepted iff over ALL of them  max |dprob| <= kProbeMaxErr (an eighth of the bar)  and  max <=
//   kProbeTailRatio x the 99.9th percentile (a light tail).  The first batch that breaks the first condition ends the probe (a trained
//   checkpoint costs one batch).  Inputs, initial states and both arithmetics are deterministic: the same weights always get the same answer.
constexpr float kProbeTailAt = 1.0e-5f;        // reported: the share of the probe sites beyond it (ccsm_model_probe_tail)
constexpr float kProbeMaxErr = 1.25e-5f;
constexpr float kProbeTailRatio = 3.0f;
constexpr float kMxH0Limit = 6.0f;
#ifndef CCSM_PROBE_SITES
#define CCSM_PROBE_SITES 8192        // sites per probe batch
#endif
#ifndef CCSM_PROBE_BATCHES
#define CCSM_PROBE_BATCHES 8
#endif
constexpr int kProbeSites = CCSM_PROBE_SITES, kProbeBatches = CCSM_PROBE_BATCHES;

// ensure(candidate): packs and uploads that arithmetic's weight streams
ccsm_status probe_arithmetic(ccsm_model* m, const std::function<ccsm_status(int)>& ensure) {
    ccsm_workspace* ws = nullptr;
    ccsm_status st = ccsm_workspace_create(m, kProbeSites, &ws);
    if (st != CCSM_OK) return st;
    st = ensure(CCSM_PRECISION_SPLIT_F8);
    if (st != CCSM_OK) { ccsm_workspace_destroy(ws); return st; }
    std::vector<uint8_t> kmer[2];
    std::vector<float> ipd[2], pw[2], npass[2], isd[2], psd[2], sn[2], mp[2];
    uint32_t sd = 0x9e3779b9u;
    auto rnd = [&]() { sd = sd * 1664525u + 1013904223u; return (float)(sd >> 8) * (1.0f / 16777216.0f); };
    ccsm_batch b;
    std::memset(&b, 0, sizeof(b));
    auto draw = [&]() {                                         // the next batch of the (one) probe sequence
        for (int s = 0; s < 2; ++s) {
            kmer[s].resize((size_t)kProbeSites * kSeqLen);
            ipd[s].resize(kmer[s].size());
            pw[s].resize(kmer[s].size());
            npass[s].resize(kProbeSites);
            for (int i = 0; i < kProbeSites; ++i) {
                npass[s][i] = 3.0f + std::floor(rnd() * 28.0f);
                for (int t = 0; t < kSeqLen; ++t) {
                    const size_t e = (size_t)i * kSeqLen + t;
                    kmer[s][e] = t == 10 ? 1 : t == 11 ? 2 : (uint8_t)(rnd() * 4.0f);
                    const float g0 = rnd() + rnd() + rnd() + rnd() - 2.0f, g1 = rnd() + rnd() + rnd() + rnd() - 2.0f;   // ~N(0, 1/3)
                    ipd[s][e] = 1.7f * g0 + (rnd() < 0.02f ? 8.0f * rnd() : 0.f);                                       // kinetics z-scores: a heavy right tail
                    pw[s][e] = 1.7f * g1 + (rnd() < 0.02f ? 8.0f * rnd() : 0.f);
                }
            }
            b.strand[s].kmer = kmer[s].data(); b.strand[s].ipd = ipd[s].data(); b.strand[s].pw = pw[s].data(); b.strand[s].npass = npass[s].data();
            auto fill = [&](std::vector<float>& v, size_t cnt, float lo, float hi) {
                v.resize(cnt);
                for (float& x : v) x = lo + (hi - lo) * rnd();
                return v.data();
            };
            if (m->feat & kFeatStds) {
                b.strand[s].ipd_std = fill(isd[s], kmer[s].size(), 0.f, 2.f);
                b.strand[s].pw_std = fill(psd[s], kmer[s].size(), 0.f, 2.f);
            }
            if (m->feat & kFeatSn) b.strand[s].sn = fill(sn[s], (size_t)kProbeSites * 4, 3.f, 16.f);
            if (m->feat & kFeatMap) b.strand[s].map = fill(mp[s], kmer[s].size(), 0.f, 1.f);
        }
    };
    ccsm_h0 h0;
    std::memset(&h0, 0, sizeof(h0));
    h0.mode = CCSM_H0_DEVICE_RNG;
    h0.seed = 20260928;
    std::vector<float> lg((size_t)kProbeSites * 2), pa(lg.size()), pb(lg.size()), all;
    all.reserve((size_t)kProbeSites * kProbeBatches);
    const int wanted = m->precision;
    float err = 0.f;
    int beyond = 0;
    const bool forced = std::getenv("CCSM_NO_PRECISION_FALLBACK") != nullptr;
    for (int k = 0; k < kProbeBatches && st == CCSM_OK; ++k) {
        draw();
        h0.offset = (uint64_t)k * kProbeSites;                 // every batch its own initial states
        m->precision = CCSM_PRECISION_SPLIT3;
        st = ccsm_forward_host(m, ws, kProbeSites, &b, &h0, lg.data(), pb.data(), nullptr);
        if (st != CCSM_OK) break;
        m->precision = CCSM_PRECISION_SPLIT_F8;
        st = ccsm_forward_host(m, ws, kProbeSites, &b, &h0, lg.data(), pa.data(), nullptr);
        if (st != CCSM_OK) break;
        for (int i = 0; i < kProbeSites; ++i) {
            const float d = std::isfinite(pa[2 * i + 1]) ? std::fabs(pa[2 * i + 1] - pb[2 * i + 1]) : 1.0f;
            all.push_back(d);
            err = std::fmax(err, d);
            beyond += d > kProbeTailAt;
        }
        if (err > kProbeMaxErr && !forced) break;              // decided: the rest of the probe would not change it
    }
    float q999 = 0.f;
    if (!all.empty()) {
        const size_t r = std::min(all.size() - 1, (size_t)std::floor(0.999 * (double)all.size()));
        std::nth_element(all.begin(), all.begin() + (long)r, all.end());
        q999 = all[r];
    }
    m->probe_err = err;
    m->probe_q999 = q999;
    m->probe_n = (int)all.size();
    m->probe_tail = all.empty() ? -1.f : (float)beyond / (float)all.size();
    const bool ok = (int)all.size() == kProbeSites * kProbeBatches && err <= kProbeMaxErr && err <= kProbeTailRatio * q999;
    if (forced && !ok)
        std::fprintf(stderr, "[libccsm] WARNING: CCSM_NO_PRECISION_FALLBACK is set: serving split-mx although its probe FAILED "
                             "(max |dprob| %.2e over %d probe sites, 99.9 %% %.2e; rule: max <= %.2e and <= %.1f x the 99.9th percentile)\n",
                     (double)err, (int)all.size(), (double)q999, (double)kProbeMaxErr, (double)kProbeTailRatio);
    m->precision = st != CCSM_OK ? wanted : (ok || forced) ? CCSM_PRECISION_SPLIT_F8 : CCSM_PRECISION_SPLIT3;
    ccsm_workspace_destroy(ws);
    return st;
}
}  // namespace

extern "C" {

ccsm_status ccsm_create(const ccsm_config* cfg, const ccsm_weights* w, int device, ccsm_model** out) {
    if (!cfg || !w || !out) return fail(CCSM_ERR_INVALID_ARG, "cfg, weights and out must be non-NULL");
    *out = nullptr;
    if (!cfg->model_type || std::strcmp(cfg->model_type, "attbigru2s") != 0)
        return fail(CCSM_ERR_UNSUPPORTED, "--model_type not right! (this build implements attbigru2s)");
    if (cfg->seq_len != kSeqLen || cfg->num_layers != kLayers || cfg->num_classes != kClasses || cfg->hidden_size != kHidden)
        return fail(CCSM_ERR_UNSUPPORTED, "this build implements seq_len 21, layer_rnn 3, class_num 2, hid_rnn 256");
    // [embedding(8) | ipd | pw | npass? | ipd_std, pw_std? | sn(4)? | map?] (models.py:39-47): the layer-0 kernels take one 16-wide
    // k-block, which holds every variant except is_npass + is_stds + is_sn (17 or 18 columns); for those the embedding is folded into
    // the matrix below ([one-hot(5) | features] = 14 or 15 columns)
    const int feat = (cfg->is_npass ? kFeatNpass : 0) | (cfg->is_stds ? kFeatStds : 0) | (cfg->is_sn ? kFeatSn : 0) | (cfg->is_map ? kFeatMap : 0);
    const int feat0 = kEmbed + 2 + (cfg->is_npass ? 1 : 0) + (cfg->is_stds ? 2 : 0) + (cfg->is_sn ? 4 : 0) + (cfg->is_map ? 1 : 0);
    const bool fold = feat0 > 16;
    const int k0 = fold ? feat0 - kEmbed + kVocab : feat0;         // columns of the layer-0 matrix as the kernels see it
    const int prec = cfg->precision == 0 ? 4 : cfg->precision;
    const bool auto_prec = cfg->precision == 0;
    if (prec != CCSM_PRECISION_SPLIT3 && prec != CCSM_PRECISION_SPLIT_F8 && prec != CCSM_PRECISION_HYBRID && prec != CCSM_PRECISION_SPLIT_MXD)
        return fail(CCSM_ERR_INVALID_ARG, "precision must be 0 (default: chosen by a probe batch), 3 (split-fp16), 4 (split-mx), 5 (hybrid) or 6 (split-mx-d)");
    if (!w->embed_weight || !w->att_wa || !w->att_ua || !w->att_va || !w->fc1_weight || !w->fc1_bias)
        return fail(CCSM_ERR_INVALID_ARG, "weights: NULL tensor");
    for (int l = 0; l < kLayers; ++l)
        for (int d = 0; d < 2; ++d)
            if (!w->weight_ih[l][d] || !w->weight_hh[l][d] || !w->bias_ih[l][d] || !w->bias_hh[l][d])
                return fail(CCSM_ERR_INVALID_ARG, "weights: NULL rnn tensor");
    HIP_TRY(hipSetDevice(device));
    ccsm_model* m = new (std::nothrow) ccsm_model();
    if (!m) return fail(CCSM_ERR_NOMEM, "out of host memory");
    m->device = device;
    m->precision = prec;
    m->feat = feat;
    m->feat0 = k0;
    m->fold = fold ? 1 : 0;
    // W'[:, c] = W[:, 0:8] E[c] (c < 5), then the feature columns: W[:, 0:8] E[code] + W[:, 8:] f = W' [one-hot(code) | f]
    std::vector<float> wfold[2];
    const float* wih_l0[2] = {w->weight_ih[0][0], w->weight_ih[0][1]};
    if (fold)
        for (int d = 0; d < 2; ++d) {
            wfold[d].resize((size_t)kGates * kHidden * k0);
            for (int r = 0; r < kGates * kHidden; ++r) {
                const float* src = w->weight_ih[0][d] + (size_t)r * feat0;
                for (int c = 0; c < kVocab; ++c) {
                    double a = 0.0;
                    for (int j = 0; j < kEmbed; ++j) a += (double)src[j] * (double)w->embed_weight[c * kEmbed + j];
                    wfold[d][(size_t)r * k0 + c] = (float)a;
                }
                for (int c = kVocab; c < k0; ++c) wfold[d][(size_t)r * k0 + c] = src[kEmbed + (c - kVocab)];
            }
            wih_l0[d] = wfold[d].data();
        }
    ccsm_status st = CCSM_OK;
    std::vector<_Float16> hbuf;
    std::vector<float> fbuf;
    for (int l = 0; l < kLayers && st == CCSM_OK; ++l) {
        const float* const* wih = l == 0 ? wih_l0 : w->weight_ih[l];
        pack_wstream_v2(l, m->feat0, wih, w->weight_hh[l], hbuf);
        st = upload(&m->wst2[l], hbuf.data(), hbuf.size() * sizeof(_Float16));
        if (st != CCSM_OK) break;
        pack_bias(w->bias_ih[l], w->bias_hh[l], fbuf);
        st = upload(&m->bias[l], fbuf.data(), fbuf.size() * sizeof(float));
    }
    if (st == CCSM_OK) { pack_att(w->att_wa, hbuf); st = upload(&m->wa, hbuf.data(), hbuf.size() * sizeof(_Float16)); }
    if (st == CCSM_OK) { pack_att(w->att_ua, hbuf); st = upload(&m->ua, hbuf.data(), hbuf.size() * sizeof(_Float16)); }
    if (st == CCSM_OK && prec >= 4) { pack_att_v3(w->att_wa, hbuf, m->att_scale[0]); st = upload(&m->wa3, hbuf.data(), hbuf.size() * sizeof(_Float16)); }
    if (st == CCSM_OK && prec >= 4) { pack_att_v3(w->att_ua, hbuf, m->att_scale[1]); st = upload(&m->ua3, hbuf.data(), hbuf.size() * sizeof(_Float16)); }
    if (st == CCSM_OK) {
        fbuf.assign((size_t)kWaves * 2 * 16, 0.f);
        for (int wave = 0; wave < kWaves; ++wave)
            for (int hh = 0; hh < 2; ++hh)
                for (int r = 0; r < 16; ++r) fbuf[(wave * 2 + hh) * 16 + r] = w->att_va[kUnitTile * wave + crow(r, hh)];
        st = upload(&m->va, fbuf.data(), fbuf.size() * sizeof(float));
    }
    if (st == CCSM_OK) st = upload(&m->fcw, w->fc1_weight, sizeof(float) * kClasses * 4 * kHidden);
    if (st == CCSM_OK) {
        std::vector<uint8_t> fb;
        pack_fc_s3(w->fc1_weight, fb);
        st = upload(&m->fc3s, fb.data(), fb.size());
    }
    if (st == CCSM_OK && prec >= 4) {
        std::vector<uint8_t> fb;
        pack_fc_v3(w->fc1_weight, fb, m->fc_scale);
        st = upload(&m->fc3, fb.data(), fb.size());
    }
    if (st == CCSM_OK) st = upload(&m->fcb, w->fc1_bias, sizeof(float) * kClasses);
    if (st == CCSM_OK) st = upload(&m->embed, w->embed_weight, sizeof(float) * kVocab * kEmbed);
    if (st == CCSM_OK) {
        hipError_t e = hipSuccess;
        auto set_lds = [&](const void* fn, int bytes) {
            if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        };
        set_lds(reinterpret_cast<const void*>(&gru_layer_v2_kernel<kKB0>), gru2_lds(kKB0));
        set_lds(reinterpret_cast<const void*>(&gru_layer_v2_kernel<kKB12>), gru2_lds(kKB12));
        set_lds(reinterpret_cast<const void*>(&attn_fc_kernel), kAttLds);
        set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<false, true, false, 3, true>), mx0_lds(3));
        set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<false, true, false, 3, true, true>), mx0_lds(3));
        set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<false, true, false, 2, true>), mx0_lds(2));
        set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<false, true, false, 1, true>), mx0_lds(1));
        set_lds(reinterpret_cast<const void*>(&gru_layer12_f3_kernel<3>), f3_lds(3));
        set_lds(reinterpret_cast<const void*>(&gru_layer12_f3_kernel<2>), f3_lds(2));
        set_lds(reinterpret_cast<const void*>(&gru_layer12_f3_kernel<1>), f3_lds(1));
        set_lds(reinterpret_cast<const void*>(&gru_layer12_f3s_kernel<3>), f3_lds(3));
        set_lds(reinterpret_cast<const void*>(&gru_layer12_f3s_kernel<2>), f3_lds(2));
        set_lds(reinterpret_cast<const void*>(&gru_layer12_f3s_kernel<1>), f3_lds(1));
        set_lds(reinterpret_cast<const void*>(&gru_layer0_f3s_kernel<3>), mx0_lds(3));
        set_lds(reinterpret_cast<const void*>(&gru_layer0_f3s_kernel<2>), mx0_lds(2));
        set_lds(reinterpret_cast<const void*>(&gru_layer0_f3s_kernel<1>), mx0_lds(1));
#ifdef CCSM_PHASE_STAMPS
        set_lds(reinterpret_cast<const void*>(&gru_layer12_f3_kernel<3, true>), f3_lds(3));
        set_lds(reinterpret_cast<const void*>(&gru_layer12_f3s_kernel<3, true>), f3_lds(3));
#endif
        if (prec >= CCSM_PRECISION_SPLIT_F8) {
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<false, false>), kMx0Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<false, false, false, 3, false, true>), kMx0Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<false, true, false, 3, false, true>), kMx0Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<false, false, true, 3, false, true>), kMx0Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<false, false, false>), kMx12Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<true, false, false>), kMx12Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<false, true>), kMx0Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<false, false, true>), kMx12Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<true, false, true>), kMx12Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<false, false, true>), kMx0Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<false, false, false, true>), kMx12Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<true, false, false, true>), kMx12Lds);
            // the 64-row forms
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<false, false, false, 2>), mx0_lds(2));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<false, false, false, false, 2>), mx12_lds(2));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<true, false, false, false, 2>), mx12_lds(2));
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<false, true, false, 2>), mx0_lds(2));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<false, false, true, false, 2>), mx12_lds(2));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<true, false, true, false, 2>), mx12_lds(2));
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<false, false, true, 2>), mx0_lds(2));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<false, false, false, true, 2>), mx12_lds(2));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<true, false, false, true, 2>), mx12_lds(2));
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<false, false, false, 1>), mx0_lds(1));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<false, false, false, false, 1>), mx12_lds(1));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<true, false, false, false, 1>), mx12_lds(1));
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<false, true, false, 1>), mx0_lds(1));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<false, false, true, false, 1>), mx12_lds(1));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<true, false, true, false, 1>), mx12_lds(1));
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<false, false, true, 1>), mx0_lds(1));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<false, false, false, true, 1>), mx12_lds(1));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<true, false, false, true, 1>), mx12_lds(1));
#ifdef CCSM_PHASE_STAMPS
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<true, false>), kMx0Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<true, false, false, 3, false, true>), kMx0Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<true, true, false, 3, false, true>), kMx0Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<true, false, true, 3, false, true>), kMx0Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<false, true, false>), kMx12Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<true, true, false>), kMx12Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<true, true>), kMx0Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<false, true, true>), kMx12Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<true, true, true>), kMx12Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer0_mx_kernel<true, false, true>), kMx0Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<false, true, false, true>), kMx12Lds);
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx_kernel<true, true, false, true>), kMx12Lds);
#endif
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx16_kernel<false, 3>), mx12_lds(3));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx16_kernel<true, 3>), mx12_lds(3));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx16_kernel<false, 2>), mx12_lds(2));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx16_kernel<true, 2>), mx12_lds(2));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx16_kernel<false, 1>), mx12_lds(1));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx16_kernel<true, 1>), mx12_lds(1));
#ifdef CCSM_PHASE_STAMPS
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx16_kernel<false, 3, true>), mx12_lds(3));
            set_lds(reinterpret_cast<const void*>(&gru_layer12_mx16_kernel<true, 3, true>), mx12_lds(3));
#endif
            set_lds(reinterpret_cast<const void*>(&attn_fc_f8_kernel), kAttF8Lds);
        }
        if (e != hipSuccess) st = fail(CCSM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
    }
    // split3's streams on the split-mx schedule (always: it is the reference of the probe and the arithmetic of explicit large initial states)
    m->split3_v2 = std::getenv("CCSM_SPLIT3_V2") != nullptr;
    m->l0_stag = std::getenv("CCSM_L0_LOCKSTEP") == nullptr;
    m->f3_shape32 = std::getenv("CCSM_F3_SHAPE32") != nullptr;
    m->mx_shape16 = std::getenv("CCSM_MX_SHAPE16") != nullptr;
#ifdef CCSM_STAGGER_DIAG
    {
        const int stagger = std::getenv("CCSM_STAGGER") ? std::atoi(std::getenv("CCSM_STAGGER")) : 0;
        if (st == CCSM_OK && hipMemcpyToSymbol(HIP_SYMBOL(g_ccsm_stagger), &stagger, sizeof(int)) != hipSuccess) st = fail(CCSM_ERR_HIP, "hipMemcpyToSymbol(g_ccsm_stagger)");
    }
#endif
    if (st == CCSM_OK) {
        std::vector<uint8_t> bbuf;
        pack_wstream_mx(0, m->feat0, wih_l0, w->weight_hh[0], true, bbuf);
        st = upload(&m->wstf3[0], bbuf.data(), bbuf.size());
        std::vector<float> nbuf;
        if (!m->f3_shape32 && st == CCSM_OK) {
            pack_wstream_f3s0(m->feat0, wih_l0, w->weight_hh[0], bbuf);
            st = upload(&m->wstf3s[0], bbuf.data(), bbuf.size());
            pack_bias_natural(w->bias_ih[0], w->bias_hh[0], nbuf);
            if (st == CCSM_OK) st = upload(&m->biasn[0], nbuf.data(), nbuf.size() * sizeof(float));
        }
        for (int l = 1; l < kLayers && st == CCSM_OK; ++l) {
            if (m->f3_shape32) {
                pack_wstream_f3(w->weight_ih[l], w->weight_hh[l], bbuf);
                st = upload(&m->wstf3[l], bbuf.data(), bbuf.size());
            } else {
                pack_wstream_f3s(w->weight_ih[l], w->weight_hh[l], bbuf);
                st = upload(&m->wstf3s[l], bbuf.data(), bbuf.size());
                pack_bias_natural(w->bias_ih[l], w->bias_hh[l], nbuf);
                if (st == CCSM_OK) st = upload(&m->biasn[l], nbuf.data(), nbuf.size() * sizeof(float));
            }
        }
    }
    // the weight streams of one arithmetic of the split-mx family (a forced precision: that one; the default: what the probe gets to)
    float qerr[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto ensure_streams = [&](int which) -> ccsm_status {
        uint4** dst = which == CCSM_PRECISION_SPLIT_F8 ? m->wstmx : which == CCSM_PRECISION_HYBRID ? m->wsthy : m->wstmd;
        if (dst[0]) return CCSM_OK;
        std::vector<uint8_t> bbuf;
        for (int l = 0; l < kLayers; ++l) {
            const float* const* wih = l == 0 ? wih_l0 : w->weight_ih[l];
            qerr[which] = std::fmax(qerr[which], pack_wstream_mx(l, m->feat0, wih, w->weight_hh[l], which == CCSM_PRECISION_HYBRID, bbuf,
                                                                which == CCSM_PRECISION_SPLIT_MXD));
            const ccsm_status s2 = upload(&dst[l], bbuf.data(), bbuf.size());
            if (s2 != CCSM_OK) return s2;
        }
        if (which == CCSM_PRECISION_SPLIT_F8 && m->mx_shape16)           // layers 1-2 on the 16-wide instructions (ccsm_gru_mx16.hip): the same values, other fragments
            for (int l = 1; l < kLayers; ++l) {
                qerr[which] = std::fmax(qerr[which], pack_wstream_mx16(w->weight_ih[l], w->weight_hh[l], bbuf));
                ccsm_status s2 = upload(&m->wstmx16[l], bbuf.data(), bbuf.size());
                if (s2 == CCSM_OK && !m->biasn[l]) {
                    std::vector<float> nbuf;
                    pack_bias_natural(w->bias_ih[l], w->bias_hh[l], nbuf);
                    s2 = upload(&m->biasn[l], nbuf.data(), nbuf.size() * sizeof(float));
                }
                if (s2 != CCSM_OK) return s2;
            }
        return CCSM_OK;
    };
    if (st == CCSM_OK && prec >= CCSM_PRECISION_SPLIT_F8 && !auto_prec) st = ensure_streams(prec);
    if (st == CCSM_OK && auto_prec) st = probe_arithmetic(m, ensure_streams);      // leaves split-mx (a clean, light-tailed probe) or split-fp16 in m->precision
    if (st == CCSM_OK && m->precision >= CCSM_PRECISION_SPLIT_F8) m->mx_quant_err = qerr[m->precision];
    // every upload above went over the NULL stream, which a caller's non-blocking streams do not wait for: nothing of this model may still be
    // in flight when the first forward is issued on such a stream (see ccsm_workspace_create)
    if (st == CCSM_OK && hipDeviceSynchronize() != hipSuccess) st = fail(CCSM_ERR_HIP, "hipDeviceSynchronize at the end of ccsm_create");
    if (st != CCSM_OK) {
        ccsm_destroy(m);
        return st;
    }
    *out = m;
    return CCSM_OK;
}

float ccsm_model_probe_error(const ccsm_model* m) { return m ? m->probe_err : -1.f; }
float ccsm_model_probe_error_hybrid(const ccsm_model* m) { return m ? m->probe_err_hybrid : -1.f; }
float ccsm_model_probe_error_of(const ccsm_model* m, int precision) {
    return !m ? -1.f : precision == CCSM_PRECISION_SPLIT_F8 ? m->probe_err : precision == CCSM_PRECISION_HYBRID ? m->probe_err_hybrid
              : precision == CCSM_PRECISION_SPLIT_MXD ? m->probe_err_dyn : -1.f;
}
float ccsm_model_probe_tail(const ccsm_model* m, int precision) {
    return !m ? -1.f : precision == CCSM_PRECISION_SPLIT_F8 ? m->probe_tail : precision == CCSM_PRECISION_HYBRID ? m->probe_tail_hybrid
              : precision == CCSM_PRECISION_SPLIT_MXD ? m->probe_tail_dyn : -1.f;
}
float ccsm_model_probe_q999(const ccsm_model* m) { return m ? m->probe_q999 : -1.f; }
int ccsm_model_probe_sites(const ccsm_model* m) { return m ? m->probe_n : 0; }
float ccsm_model_quant_error(const ccsm_model* m) { return m ? m->mx_quant_err : -1.f; }

// ---- the probe on the caller's OWN data.  ccsm_create's probe runs synthetic sites: nothing ties its inputs to what the
// caller will feed.  A caller that wants the selection rule applied to its real input runs its first batches through the arithmetic in
// use AND through split3 (ccsm_model_set_precision switches between the two; both sets of weight streams are resident), hands both
// results to ccsm_model_data_probe_add, and lets ccsm_model_data_probe_decide apply ccsm_create's rule to them: max |dprob| <= 1.25e-5 and
// (from 8192 sites on) max <= 3 x the 99.9th percentile; a failed candidate is replaced by split3 for the rest of the model's life.
// `call_mods` does this on the first <= 65536 sites of its input (reference call site: call_modifications.py:201-214).
ccsm_status ccsm_model_set_precision(ccsm_model* m, int precision) {
    if (!m) return fail(CCSM_ERR_INVALID_ARG, "model must be non-NULL");
    const bool have = precision == CCSM_PRECISION_SPLIT3 || (precision == CCSM_PRECISION_SPLIT_F8 && m->wstmx[0]) ||
                      (precision == CCSM_PRECISION_HYBRID && m->wsthy[0]) || (precision == CCSM_PRECISION_SPLIT_MXD && m->wstmd[0]);
    if (!have) return fail(CCSM_ERR_INVALID_ARG, "ccsm_model_set_precision: this model holds no weight streams of that arithmetic (split3 always; "
                                                  "of the split-mx family what ccsm_create probed or was asked for)");
    if (m->slices_bound > 0 && precision != m->precision)
        return fail(CCSM_ERR_BUSY, "ccsm_model_set_precision: slices of a group are bound to a workspace of this model and not yet run");
    m->precision = precision;
    return CCSM_OK;
}
ccsm_status ccsm_workspace_force_split3(ccsm_workspace* ws) {
    if (!ws) return fail(CCSM_ERR_INVALID_ARG, "workspace must be non-NULL");
    ws->force_split3 = true;
    return CCSM_OK;
}
ccsm_status ccsm_model_data_probe_add(ccsm_model* m, const float* probs_candidate, const float* probs_split3, int n_sites) {
    if (!m || !probs_candidate || !probs_split3 || n_sites < 0) return fail(CCSM_ERR_INVALID_ARG, "ccsm_model_data_probe_add: NULL argument");
    m->dprobe.reserve(m->dprobe.size() + (size_t)n_sites);
    for (int i = 0; i < n_sites; ++i) {
        const float a = probs_candidate[2 * i + 1], b = probs_split3[2 * i + 1];
        m->dprobe.push_back(std::isfinite(a) && std::isfinite(b) ? std::fabs(a - b) : 1.0f);
    }
    return CCSM_OK;
}
int ccsm_model_data_probe_decide(ccsm_model* m) {
    if (!m) return -1;
    float err = 0.f, q999 = 0.f;
    std::vector<float>& all = m->dprobe;
    for (float d : all) err = std::fmax(err, d);
    if (!all.empty()) {
        const size_t r = std::min(all.size() - 1, (size_t)std::floor(0.999 * (double)all.size()));
        std::nth_element(all.begin(), all.begin() + (long)r, all.end());
        q999 = all[r];
    }
    m->dprobe_err = all.empty() ? -1.f : err;
    m->dprobe_q999 = all.empty() ? -1.f : q999;
    m->dprobe_n = (int)all.size();
    // fewer than one synthetic probe batch of sites: the percentile is not a tail statistic yet, the bound on the maximum alone decides
    const bool ok = all.empty() || (err <= kProbeMaxErr && ((int)all.size() < kProbeSites || err <= kProbeTailRatio * q999));
    m->dprobe_ok = ok ? 1 : 0;
    if (!ok && m->precision != CCSM_PRECISION_SPLIT3 && std::getenv("CCSM_NO_PRECISION_FALLBACK") == nullptr) m->precision = CCSM_PRECISION_SPLIT3;
    all.clear();
    all.shrink_to_fit();
    return m->precision;
}
float ccsm_model_data_probe_error(const ccsm_model* m) { return m ? m->dprobe_err : -1.f; }
float ccsm_model_data_probe_q999(const ccsm_model* m) { return m ? m->dprobe_q999 : -1.f; }
int ccsm_model_data_probe_sites(const ccsm_model* m) { return m ? m->dprobe_n : 0; }
int ccsm_model_data_probe_verdict(const ccsm_model* m) { return m ? m->dprobe_ok : -1; }

void ccsm_destroy(ccsm_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    for (int l = 0; l < kLayers; ++l) {
        (void)hipFree(m->wst2[l]);
        (void)hipFree(m->wstmx[l]);
        (void)hipFree(m->wstmd[l]);
        (void)hipFree(m->wsthy[l]);
        (void)hipFree(m->wstf3[l]);
        (void)hipFree(m->wstf3s[l]);
        (void)hipFree(m->wstmx16[l]);
        (void)hipFree(m->biasn[l]);
        (void)hipFree(m->bias[l]);
    }
    (void)hipFree(m->wa); (void)hipFree(m->ua); (void)hipFree(m->va);
    (void)hipFree(m->wa3); (void)hipFree(m->ua3); (void)hipFree(m->fc3); (void)hipFree(m->fc3s);
    (void)hipFree(m->fcw); (void)hipFree(m->fcb); (void)hipFree(m->embed);
    delete m;
}

ccsm_status ccsm_workspace_create(const ccsm_model* m, int max_sites, ccsm_workspace** out) {
    if (!m || !out) return fail(CCSM_ERR_INVALID_ARG, "model and out must be non-NULL");
    *out = nullptr;
    if (max_sites <= 0 || max_sites > (1 << 22)) return fail(CCSM_ERR_INVALID_ARG, "max_sites must be in [1, 4194304]");
    HIP_TRY(hipSetDevice(m->device));
    ccsm_workspace* ws = new (std::nothrow) ccsm_workspace();
    if (!ws) return fail(CCSM_ERR_NOMEM, "out of host memory");
    ws->device = m->device;
    ws->max_sites = max_sites;
    ws->rows_p = rows_padded(max_sites);
    const size_t tiles = ws->rows_p / 32;
    const size_t x0_b = tiles * kSeqLen * kKB0 * 2 * 1024;
    const size_t act_b = tiles * kSeqLen * kKB12 * 2 * 1024;
    const size_t h0_b = (size_t)2 * kLayers * ws->rows_p * kHidden * sizeof(float);
    const size_t part_b = (size_t)ws->rows_p * 2 * sizeof(float);
    // host-path staging: per strand kmer f32|u8 (N,21) + ipd + pw (N,21) f32 + npass (N,21) f32 worst case
    ws->in_bytes = (size_t)2 * ((size_t)max_sites * kSeqLen * 4 * sizeof(float) + 64);  // + alignment padding
    if (m->feat & kFeatStds) ws->in_bytes += (size_t)2 * 2 * max_sites * kSeqLen * sizeof(float);
    if (m->feat & kFeatSn) ws->in_bytes += (size_t)2 * max_sites * 4 * sizeof(float);
    if (m->feat & kFeatMap) ws->in_bytes += (size_t)2 * max_sites * kSeqLen * sizeof(float);
    const size_t out_b = (size_t)max_sites * 4 * sizeof(float);
    ccsm_status st = CCSM_OK;
    auto dmalloc = [&](void** p, size_t b) -> ccsm_status {
        HIP_TRY(hipMalloc(p, b));
        ws->bytes += b;
        return CCSM_OK;
    };
    if (st == CCSM_OK) st = dmalloc((void**)&ws->x0, x0_b);
    if (st == CCSM_OK) st = dmalloc((void**)&ws->act[0], act_b);
    if (st == CCSM_OK) st = dmalloc((void**)&ws->act[1], act_b);
    if (st == CCSM_OK) st = dmalloc((void**)&ws->h0buf, h0_b);
    if (st == CCSM_OK) st = dmalloc((void**)&ws->part, part_b);
#ifdef CCSM_PHASE_STAMPS
    if (st == CCSM_OK && std::getenv("CCSM_PHASE_DEBUG")) st = dmalloc((void**)&ws->dbg, (kSeqLen * kWaves * 5 + 2 * kSeqLen * kWaves * 8 * 6) * 8);
#endif
    if (st == CCSM_OK) st = dmalloc((void**)&ws->d_in, ws->in_bytes);
    if (st == CCSM_OK) st = dmalloc((void**)&ws->d_out, out_b);
    if (st == CCSM_OK) {   // padding rows are computed (and ignored): give them finite contents once
        hipError_t e = hipMemset(ws->x0, 0, x0_b);
        if (e == hipSuccess) e = hipMemset(ws->h0buf, 0, h0_b);
        // hipMemset of device memory returns before it has run, on the NULL stream - which a caller's non-blocking streams (every
        // torch.cuda.Stream) do not wait for: a workspace created while another one is busy could be zeroed AFTER its first call's
        // extraction kernels had filled x0 (round 5: found by the byte comparison of test_call_mods_probes_the_arithmetic_on_its_own_input -
        // the second chunk of a call_mods run came out wrong in one run of three).  The creation is rare: wait here.
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) st = fail(CCSM_ERR_HIP, std::string("hipMemset: ") + hipGetErrorString(e));
    }
    if (st == CCSM_OK) {
        hipError_t e = hipHostMalloc((void**)&ws->p_in, ws->in_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void**)&ws->p_out, out_b, hipHostMallocDefault);
        if (e != hipSuccess) st = fail(CCSM_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    if (st == CCSM_OK) {
        ws->ev_ok = true;
        for (int k = 0; k < ccsm_workspace::kEvSets; ++k)
            for (int i = 0; i < 8; ++i)
                if (hipEventCreate(&ws->evs[k][i]) != hipSuccess) ws->ev_ok = false;
    }
    if (st != CCSM_OK) {
        ccsm_workspace_destroy(ws);
        return st;
    }
    *out = ws;
    return CCSM_OK;
}

void ccsm_workspace_destroy(ccsm_workspace* ws) {
    if (!ws) return;
    if (ws->bound_to && ws->n_slices > 0) {          // slices added and never run: the model (destroyed after its workspaces) stops counting them
        ws->bound_to->slices_bound -= ws->n_slices;
        if (ws->bound_to->slices_bound < 0) ws->bound_to->slices_bound = 0;
    }
    (void)hipSetDevice(ws->device);
    if (ws->pending_sites) (void)hipStreamSynchronize(ws->pending_stream);
    (void)hipFree(ws->x0); (void)hipFree(ws->act[0]); (void)hipFree(ws->act[1]);
    (void)hipFree(ws->h0buf); (void)hipFree(ws->part); (void)hipFree(ws->dbg); (void)hipFree(ws->d_in); (void)hipFree(ws->d_h0);
    (void)hipFree(ws->d_out);
    if (ws->r_pending) (void)hipStreamSynchronize(ws->r_stream);
    (void)hipFree(ws->r_bytes); (void)hipFree(ws->r_table); (void)hipFree(ws->r_locs);
    if (ws->rp_bytes) (void)hipHostFree(ws->rp_bytes);
    if (ws->rp_table) (void)hipHostFree(ws->rp_table);
    if (ws->rp_locs) (void)hipHostFree(ws->rp_locs);
    (void)hipFree(ws->d_keys);
    if (ws->p_keys) (void)hipHostFree(ws->p_keys);
    if (ws->p_in) (void)hipHostFree(ws->p_in);
    if (ws->p_h0) (void)hipHostFree(ws->p_h0);
    if (ws->p_out) (void)hipHostFree(ws->p_out);
    for (int k = 0; k < ccsm_workspace::kEvSets; ++k)
        for (int i = 0; i < 8; ++i)
            if (ws->evs[k][i]) (void)hipEventDestroy(ws->evs[k][i]);
    delete ws;
}

ccsm_status ccsm_forward_device(const ccsm_model* m, ccsm_workspace* ws, int n_sites, const ccsm_batch* b, const ccsm_h0* h0,
                                float* logits, float* probs, void* stream) {
    ccsm_status st = check_call(m, ws, n_sites, b, h0);
    if (st != CCSM_OK) return st;
    if (!logits || !probs) return fail(CCSM_ERR_INVALID_ARG, "logits and probs must be non-NULL");
    HIP_TRY(hipSetDevice(m->device));
    const StrandDev s1 = strand_dev(b->strand[0]), s2 = strand_dev(b->strand[1]);
    const int mode = h0 ? h0->mode : CCSM_H0_DEVICE_RNG;
    SiteKeys sk;
    if (h0) { sk.key = reinterpret_cast<const unsigned long long*>(h0->site_key); sk.sub = reinterpret_cast<const unsigned int*>(h0->site_sub); }
    return dispatch_forward(m, ws, n_sites, s1, s2, b->kmer_is_f32, b->npass_per_base, mode, h0 ? h0->h0[0] : nullptr,
                            h0 ? h0->h0[1] : nullptr, h0 ? h0->seed : 0, h0 ? h0->offset : 0, logits, probs,
                            static_cast<hipStream_t>(stream), sk);
}

ccsm_status ccsm_group_add_device(const ccsm_model* m, ccsm_workspace* ws, int n_sites, const ccsm_batch* b, const ccsm_h0* h0,
                                  float* logits, float* probs, void* stream) {
    if (!m || !ws || !b) return fail(CCSM_ERR_INVALID_ARG, "model, workspace and batch must be non-NULL");
    if (n_sites <= 0) return fail(CCSM_ERR_INVALID_ARG, "n_sites must be > 0");
    ccsm_status st = check_call(m, ws, std::min(n_sites, ws->max_sites), b, h0);
    if (st != CCSM_OK) return st;
    if (!logits || !probs) return fail(CCSM_ERR_INVALID_ARG, "logits and probs must be non-NULL");
    HIP_TRY(hipSetDevice(m->device));
    const StrandDev s1 = strand_dev(b->strand[0]), s2 = strand_dev(b->strand[1]);
    const int mode = h0 ? h0->mode : CCSM_H0_DEVICE_RNG;
    SiteKeys sk;
    if (h0) { sk.key = reinterpret_cast<const unsigned long long*>(h0->site_key); sk.sub = reinterpret_cast<const unsigned int*>(h0->site_sub); }
    st = add_slice(m, ws, n_sites, s1, s2, b->kmer_is_f32, b->npass_per_base, mode, h0 ? h0->h0[0] : nullptr,
                   h0 ? h0->h0[1] : nullptr, h0 ? h0->seed : 0, h0 ? h0->offset : 0, logits, probs,
                   static_cast<hipStream_t>(stream), sk);
    if (st == CCSM_OK) { m->slices_bound++; ws->bound_to = m; }     // (until ccsm_group_run: ccsm_model_set_precision refuses in between)
    return st;
}

ccsm_status ccsm_group_run(const ccsm_model* m, ccsm_workspace* ws, void* stream) {
    if (!m || !ws) return fail(CCSM_ERR_INVALID_ARG, "model and workspace must be non-NULL");
    if (ws->n_slices == 0) return CCSM_OK;
    HIP_TRY(hipSetDevice(m->device));
    if (ws->bound_to == m) {
        m->slices_bound -= ws->n_slices;
        if (m->slices_bound < 0) m->slices_bound = 0;
        ws->bound_to = nullptr;
    }
    return dispatch_run(m, ws, static_cast<hipStream_t>(stream));
}

int ccsm_group_pending(const ccsm_workspace* ws) { return ws ? ws->n_slices : 0; }

ccsm_status ccsm_submit_host(const ccsm_model* m, ccsm_workspace* ws, int n_sites, const ccsm_batch* b, const ccsm_h0* h0,
                             void* stream) {
    ccsm_status st = check_call(m, ws, n_sites, b, h0);
    if (st != CCSM_OK) return st;
    if (ws->pending_sites) return fail(CCSM_ERR_INVALID_ARG, "workspace already has a batch in flight (call ccsm_wait_host)");
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t hs = static_cast<hipStream_t>(stream);
    // ---- stage the features into one pinned block: per strand [kmer | ipd | pw | npass]
    const size_t n = (size_t)n_sites;
    const size_t kmer_b = n * kSeqLen * (b->kmer_is_f32 ? 4 : 1);
    const size_t kmer_pad = (kmer_b + 15) & ~(size_t)15;
    const size_t f_b = n * kSeqLen * sizeof(float);
    const size_t np_b = b->npass_per_base ? f_b : ((n * sizeof(float) + 15) & ~(size_t)15);
    const size_t sn_b = n * 4 * sizeof(float);
    const size_t strand_b = kmer_pad + 2 * f_b + np_b + ((m->feat & kFeatStds) ? 2 * f_b : 0) + ((m->feat & kFeatSn) ? sn_b : 0) +
                            ((m->feat & kFeatMap) ? f_b : 0);
    if (2 * strand_b > ws->in_bytes) return fail(CCSM_ERR_CAPACITY, "staging buffer too small");
    StrandDev sd[2];
    for (int s = 0; s < 2; ++s) {
        uint8_t* hp = ws->p_in + s * strand_b;
        uint8_t* dp = ws->d_in + s * strand_b;
        std::memcpy(hp, b->strand[s].kmer, kmer_b);
        std::memcpy(hp + kmer_pad, b->strand[s].ipd, f_b);
        std::memcpy(hp + kmer_pad + f_b, b->strand[s].pw, f_b);
        if (m->feat & kFeatNpass) std::memcpy(hp + kmer_pad + 2 * f_b, b->strand[s].npass, b->npass_per_base ? f_b : n * sizeof(float));
        sd[s].kmer = dp;
        sd[s].ipd = reinterpret_cast<const float*>(dp + kmer_pad);
        sd[s].pw = reinterpret_cast<const float*>(dp + kmer_pad + f_b);
        sd[s].npass = reinterpret_cast<const float*>(dp + kmer_pad + 2 * f_b);
        size_t o = kmer_pad + 2 * f_b + np_b;                       // the variant's optional planes
        auto plane = [&](const float* src, size_t bytes) {
            std::memcpy(hp + o, src, bytes);
            const float* d = reinterpret_cast<const float*>(dp + o);
            o += bytes;
            return d;
        };
        if (m->feat & kFeatStds) { sd[s].ipd_std = plane(b->strand[s].ipd_std, f_b); sd[s].pw_std = plane(b->strand[s].pw_std, f_b); }
        if (m->feat & kFeatSn) sd[s].sn = plane(b->strand[s].sn, sn_b);
        if (m->feat & kFeatMap) sd[s].map = plane(b->strand[s].map, f_b);
    }
    HIP_TRY(hipMemcpyAsync(ws->d_in, ws->p_in, 2 * strand_b, hipMemcpyHostToDevice, hs));
    const int mode = h0 ? h0->mode : CCSM_H0_DEVICE_RNG;
    const float *h0a = nullptr, *h0b = nullptr;
    if (mode == CCSM_H0_EXPLICIT) {
        // 12 KiB per site: parity/test path only (never stream h0 in production)
        const size_t cap = (size_t)2 * 2 * kLayers * ws->max_sites * kHidden * sizeof(float);
        if (!ws->d_h0) {
            HIP_TRY(hipMalloc((void**)&ws->d_h0, cap));
            ws->bytes += cap;
            HIP_TRY(hipHostMalloc((void**)&ws->p_h0, cap, hipHostMallocDefault));
        }
        const size_t hb = (size_t)2 * kLayers * n * kHidden * sizeof(float);
        std::memcpy(ws->p_h0, h0->h0[0], hb);
        std::memcpy(reinterpret_cast<uint8_t*>(ws->p_h0) + hb, h0->h0[1], hb);
        // split-mx's correction operands cover |h| up to ~8 (|h_t| <= max(1, |h0|)); a call with larger explicit initial states is
        // served in the split-fp16 arithmetic (host-pointer entry points only: device-resident initial states are not inspected)
        if (m->precision >= CCSM_PRECISION_SPLIT_F8) {       // split-mx and the hybrid (its activation blobs have the same range)
            float mx = 0.f;
            const float* p = ws->p_h0;
            for (size_t i = 0; i < 2 * hb / sizeof(float); ++i) mx = std::fmax(mx, std::fabs(p[i]));
            if (!(mx <= kMxH0Limit)) ws->force_split3 = true;
        }
        HIP_TRY(hipMemcpyAsync(ws->d_h0, ws->p_h0, 2 * hb, hipMemcpyHostToDevice, hs));
        h0a = ws->d_h0;
        h0b = ws->d_h0 + hb / sizeof(float);
    }
    SiteKeys sk;
    if (mode == CCSM_H0_DEVICE_RNG && h0 && h0->site_key) {          // 12 bytes per site next to the 680 of the features
        if (!ws->d_keys) {
            HIP_TRY(hipMalloc((void**)&ws->d_keys, (size_t)ws->max_sites * 12 + 64));
            HIP_TRY(hipHostMalloc((void**)&ws->p_keys, (size_t)ws->max_sites * 12 + 64, hipHostMallocDefault));
        }
        std::memcpy(ws->p_keys, h0->site_key, n * 8);
        if (h0->site_sub) std::memcpy(ws->p_keys + (size_t)ws->max_sites * 8, h0->site_sub, n * 4);
        HIP_TRY(hipMemcpyAsync(ws->d_keys, ws->p_keys, n * 8, hipMemcpyHostToDevice, hs));
        if (h0->site_sub)
            HIP_TRY(hipMemcpyAsync(ws->d_keys + (size_t)ws->max_sites * 8, ws->p_keys + (size_t)ws->max_sites * 8, n * 4, hipMemcpyHostToDevice, hs));
        sk.key = reinterpret_cast<const unsigned long long*>(ws->d_keys);
        sk.sub = h0->site_sub ? reinterpret_cast<const unsigned int*>(ws->d_keys + (size_t)ws->max_sites * 8) : nullptr;
    }
    float* d_logits = ws->d_out;
    float* d_probs = ws->d_out + n * 2;
    st = dispatch_forward(m, ws, n_sites, sd[0], sd[1], b->kmer_is_f32, b->npass_per_base, mode, h0a, h0b, h0 ? h0->seed : 0,
                          h0 ? h0->offset : 0, d_logits, d_probs, hs, sk);
    if (st != CCSM_OK) return st;
    HIP_TRY(hipMemcpyAsync(ws->p_out, ws->d_out, n * 4 * sizeof(float), hipMemcpyDeviceToHost, hs));
    ws->pending_sites = n_sites;
    ws->pending_stream = hs;
    return CCSM_OK;
}

ccsm_status ccsm_wait_host(ccsm_workspace* ws, float* logits, float* probs) {
    if (!ws || !logits || !probs) return fail(CCSM_ERR_INVALID_ARG, "workspace, logits and probs must be non-NULL");
    if (!ws->pending_sites) return fail(CCSM_ERR_INVALID_ARG, "no batch in flight on this workspace");
    HIP_TRY(hipSetDevice(ws->device));
    const size_t n = (size_t)ws->pending_sites;
    ws->pending_sites = 0;
    HIP_TRY(hipStreamSynchronize(ws->pending_stream));
    std::memcpy(logits, ws->p_out, n * 2 * sizeof(float));
    std::memcpy(probs, ws->p_out + n * 2, n * 2 * sizeof(float));
    return CCSM_OK;
}

ccsm_status ccsm_forward_host(const ccsm_model* m, ccsm_workspace* ws, int n_sites, const ccsm_batch* b, const ccsm_h0* h0,
                              float* logits, float* probs, void* stream) {
    if (!logits || !probs) return fail(CCSM_ERR_INVALID_ARG, "logits and probs must be non-NULL");
    ccsm_status st = ccsm_submit_host(m, ws, n_sites, b, h0, stream);
    if (st != CCSM_OK) return st;
    return ccsm_wait_host(ws, logits, probs);
}

// ---- read-level entry: raw CCS read arrays in, per-site calls out (feature extraction on the GPU) -----------------------
// Enqueue everything for a chunk of reads on `stream`: H2D of the raw arrays (through the workspace's pinned block), per-read
// statistics, h0, fragment packing, the model, D2H of locs / logits / probs.  With site_counts (the caller's per-read kept-site
// counts, e.g. ccsm_bam_batch.n_sites) nothing waits on the GPU in here; without them one round trip fetches the counts.
ccsm_status ccsm_submit_reads_host(const ccsm_model* m, ccsm_workspace* ws, const ccsm_reads* rd, const int32_t* site_counts,
                                   const ccsm_h0* h0, void* stream) {
    if (!m || !ws || !rd) return fail(CCSM_ERR_INVALID_ARG, "model, workspace and reads must be non-NULL");
    if (rd->n_reads <= 0) return fail(CCSM_ERR_INVALID_ARG, "n_reads must be > 0");
    if (m->feat != kFeatNpass)
        return fail(CCSM_ERR_UNSUPPORTED, "the read-level entry points build the default features (is_npass only); a model with is_stds / "
                                          "is_sn / is_map takes per-site features (ccsm_forward_host / _device)");
    if (!rd->offset || !rd->length || !rd->seq || !rd->fi || !rd->ri || !rd->fp || !rd->rp || !rd->fn || !rd->rn)
        return fail(CCSM_ERR_INVALID_ARG, "read arrays must be non-NULL");
    if (ws->device != m->device) return fail(CCSM_ERR_INVALID_ARG, "workspace and model live on different devices");
    if (ws->pending_sites || ws->n_slices || ws->r_pending) return fail(CCSM_ERR_INVALID_ARG, "workspace has work in flight");
    if (h0 && (h0->mode < 0 || h0->mode > 2)) return fail(CCSM_ERR_INVALID_ARG, "unknown h0 mode");
    const int nr = rd->n_reads;
    size_t total = 0;
    for (int r = 0; r < nr; ++r) {
        if (rd->length[r] <= 0 || rd->offset[r] < 0) return fail(CCSM_ERR_INVALID_ARG, "read lengths must be > 0 and offsets >= 0");
        total = std::max(total, (size_t)rd->offset[r] + (size_t)rd->length[r]);
    }
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t hs = static_cast<hipStream_t>(stream);
    if (total > ws->r_cap_bases) {
        (void)hipFree(ws->r_bytes);
        if (ws->rp_bytes) (void)hipHostFree(ws->rp_bytes);
        ws->r_bytes = ws->rp_bytes = nullptr;
        ws->r_cap_bases = 0;
        const size_t cap = ((total * 5 / 4 + 4095) / 4096) * 4096;
        HIP_TRY(hipMalloc((void**)&ws->r_bytes, cap * 5));
        HIP_TRY(hipHostMalloc((void**)&ws->rp_bytes, cap * 5, hipHostMallocDefault));
        ws->r_cap_bases = cap;
    }
    if (nr > ws->r_cap_reads) {
        (void)hipFree(ws->r_table);
        if (ws->rp_table) (void)hipHostFree(ws->rp_table);
        ws->r_table = ws->rp_table = nullptr;
        ws->r_cap_reads = 0;
        const int cap = ((nr * 5 / 4 + 63) / 64) * 64;
        HIP_TRY(hipMalloc((void**)&ws->r_table, (size_t)cap * kReadTableBytes + 64));
        HIP_TRY(hipHostMalloc((void**)&ws->rp_table, (size_t)cap * kReadTableBytes + 64, hipHostMallocDefault));
        ws->r_cap_reads = cap;
    }
    if (!ws->r_locs) {
        HIP_TRY(hipMalloc((void**)&ws->r_locs, (size_t)ws->max_sites * sizeof(int) + 64));
        HIP_TRY(hipHostMalloc((void**)&ws->rp_locs, (size_t)ws->max_sites * sizeof(int) + 64, hipHostMallocDefault));
    }
    const size_t cb = ws->r_cap_bases;
    uint8_t* d_arr[5];
    const uint8_t* h_arr[5] = {rd->seq, rd->fi, rd->ri, rd->fp, rd->rp};
    for (int a = 0; a < 5; ++a) {
        d_arr[a] = ws->r_bytes + a * cb;
        std::memcpy(ws->rp_bytes + a * cb, h_arr[a], total);
        HIP_TRY(hipMemcpyAsync(d_arr[a], ws->rp_bytes + a * cb, total, hipMemcpyHostToDevice, hs));
    }
    // per-read table, device and pinned mirror: offset i64 | stats f64 x8 | length | fn | rn | nsites | first_site (+1) | flag
    const size_t cr = (size_t)ws->r_cap_reads;
    auto at = [&](uint8_t* base, size_t off) { return base + cr * off; };
    long long* d_off = reinterpret_cast<long long*>(at(ws->r_table, 0));
    double* d_stats = reinterpret_cast<double*>(at(ws->r_table, 8));
    int* d_len = reinterpret_cast<int*>(at(ws->r_table, 72));
    float* d_fn = reinterpret_cast<float*>(at(ws->r_table, 76));
    float* d_rn = reinterpret_cast<float*>(at(ws->r_table, 80));
    int* d_ns = reinterpret_cast<int*>(at(ws->r_table, 84));
    int* d_first = reinterpret_cast<int*>(at(ws->r_table, 88));        // nr + 1 entries (capacity: cr + 16)
    std::memcpy(at(ws->rp_table, 0), rd->offset, (size_t)nr * 8);
    std::memcpy(at(ws->rp_table, 72), rd->length, (size_t)nr * 4);
    std::memcpy(at(ws->rp_table, 76), rd->fn, (size_t)nr * 4);
    std::memcpy(at(ws->rp_table, 80), rd->rn, (size_t)nr * 4);
    HIP_TRY(hipMemcpyAsync(d_off, at(ws->rp_table, 0), (size_t)nr * 8, hipMemcpyHostToDevice, hs));
    HIP_TRY(hipMemcpyAsync(d_len, at(ws->rp_table, 72), (size_t)nr * 4, hipMemcpyHostToDevice, hs));
    HIP_TRY(hipMemcpyAsync(d_fn, at(ws->rp_table, 76), (size_t)nr * 4, hipMemcpyHostToDevice, hs));
    HIP_TRY(hipMemcpyAsync(d_rn, at(ws->rp_table, 80), (size_t)nr * 4, hipMemcpyHostToDevice, hs));
    ccsm_extract::ReadTable rt{d_off, d_len, d_fn, d_rn};
    hipLaunchKernelGGL(ccsm_extract::extract_stats_kernel, dim3(nr), dim3(256), 0, hs, rt, d_arr[0], d_arr[1], d_arr[2], d_arr[3],
                       d_arr[4], d_stats, d_ns);
    HIP_TRY(hipGetLastError());
    int32_t* h_first = reinterpret_cast<int32_t*>(at(ws->rp_table, 88));
    h_first[0] = 0;
    if (site_counts) {
        for (int r = 0; r < nr; ++r) h_first[r + 1] = site_counts[r];
    } else {   // the number of rows is data dependent: one round trip for the per-read site counts
        HIP_TRY(hipMemcpyAsync(h_first + 1, d_ns, (size_t)nr * 4, hipMemcpyDeviceToHost, hs));
        HIP_TRY(hipStreamSynchronize(hs));
    }
    long long acc = 0;
    for (int r = 0; r < nr; ++r) {
        if (h_first[r + 1] < 0) return fail(CCSM_ERR_INVALID_ARG, "negative site count");
        acc += h_first[r + 1];
        if (acc > ws->max_sites) return fail(CCSM_ERR_CAPACITY, "reads hold more sites than the workspace's max_sites");
        h_first[r + 1] = (int32_t)acc;
    }
    const int n_sites = (int)acc;
    ws->r_pending = true;
    ws->r_nreads = nr;
    ws->r_nsites = n_sites;
    ws->r_checked = site_counts != nullptr;
    ws->r_stream = hs;
    if (n_sites == 0) return CCSM_OK;
    HIP_TRY(hipMemcpyAsync(d_first, h_first, (size_t)(nr + 1) * 4, hipMemcpyHostToDevice, hs));
    const int mode = h0 ? h0->mode : CCSM_H0_DEVICE_RNG;
    const float *h0a = nullptr, *h0b = nullptr;
    if (mode == CCSM_H0_EXPLICIT) {   // host tensors (6, n_sites, 256) per strand: parity/test path only
        if (!h0->h0[0] || !h0->h0[1]) { ws->r_pending = false; return fail(CCSM_ERR_INVALID_ARG, "explicit h0 needs both strand tensors"); }
        const size_t cap = (size_t)2 * 2 * kLayers * ws->max_sites * kHidden * sizeof(float);
        if (!ws->d_h0) {
            HIP_TRY(hipMalloc((void**)&ws->d_h0, cap));
            ws->bytes += cap;
            HIP_TRY(hipHostMalloc((void**)&ws->p_h0, cap, hipHostMallocDefault));
        }
        const size_t hb = (size_t)2 * kLayers * n_sites * kHidden * sizeof(float);
        std::memcpy(ws->p_h0, h0->h0[0], hb);
        std::memcpy(reinterpret_cast<uint8_t*>(ws->p_h0) + hb, h0->h0[1], hb);
        // split-mx's correction operands cover |h| up to ~8 (|h_t| <= max(1, |h0|)); a call with larger explicit initial states is
        // served in the split-fp16 arithmetic (host-pointer entry points only: device-resident initial states are not inspected)
        if (m->precision >= CCSM_PRECISION_SPLIT_F8) {       // split-mx and the hybrid (its activation blobs have the same range)
            float mx = 0.f;
            const float* p = ws->p_h0;
            for (size_t i = 0; i < 2 * hb / sizeof(float); ++i) mx = std::fmax(mx, std::fabs(p[i]));
            if (!(mx <= kMxH0Limit)) ws->force_split3 = true;
        }
        HIP_TRY(hipMemcpyAsync(ws->d_h0, ws->p_h0, 2 * hb, hipMemcpyHostToDevice, hs));
        h0a = ws->d_h0;
        h0b = ws->d_h0 + hb / sizeof(float);
    }
    // per-read keys of the device-drawn initial states: the extraction writes each site's key next to its location, and the states
    // are then drawn from (key, location)
    unsigned long long* d_rkey = nullptr;
    unsigned long long* d_skey = nullptr;
    if (mode == CCSM_H0_DEVICE_RNG && rd->h0_key) {
        if (!ws->d_keys) {
            HIP_TRY(hipMalloc((void**)&ws->d_keys, (size_t)ws->max_sites * 12 + 64));
            HIP_TRY(hipHostMalloc((void**)&ws->p_keys, (size_t)ws->max_sites * 12 + 64, hipHostMallocDefault));
        }
        // the read keys ride in the per-read table (slot at byte 96 of every entry block)
        std::memcpy(at(ws->rp_table, 96), rd->h0_key, (size_t)nr * 8);
        d_rkey = reinterpret_cast<unsigned long long*>(at(ws->r_table, 96));
        HIP_TRY(hipMemcpyAsync(d_rkey, at(ws->rp_table, 96), (size_t)nr * 8, hipMemcpyHostToDevice, hs));
        d_skey = reinterpret_cast<unsigned long long*>(ws->d_keys);
    }
    hipLaunchKernelGGL(ccsm_extract::extract_pack_kernel, dim3(nr, ccsm_extract::kPackParts), dim3(256), 0, hs, rt, d_arr[0], d_arr[1], d_arr[2], d_arr[3],
                       d_arr[4], d_stats, d_first, m->embed, ws->x0, ws->r_locs, n_sites, 0, d_rkey, d_skey);
    const size_t total4 = (size_t)2 * kLayers * 2 * n_sites * (kHidden / 4);
    hipLaunchKernelGGL(prep_h0_kernel, dim3((int)std::min<size_t>((total4 + 255) / 256, 4096)), dim3(256), 0, hs, ws->h0buf, h0a,
                       h0b, n_sites, 0, ws->rows_p, mode, h0 ? h0->seed : 0, h0 ? h0->offset : 0, d_skey,
                       d_skey ? reinterpret_cast<const unsigned int*>(ws->r_locs) : nullptr);
    HIP_TRY(hipGetLastError());
    ws->n_slices = 1;
    ws->rows_used = 2 * n_sites;
    ws->slice_row[0] = 0;
    ws->slice_n[0] = n_sites;
    ws->slice_logits[0] = ws->d_out;
    ws->slice_probs[0] = ws->d_out + (size_t)n_sites * 2;
    // CCSM_NULL_MODEL=1 (diagnostics: tools/host_feed_probe.py): everything but the BiGRU / attention launches - transfers, extraction,
    // initial states, result copies - with constant outputs (p0 = p1), to measure what the HOST side of call_mods sustains
    static const bool null_model = []() { const char* e = std::getenv("CCSM_NULL_MODEL"); return e && e[0] == '1'; }();
    ccsm_status st = CCSM_OK;
    if (null_model) {
        ws->n_slices = 0;
        ws->rows_used = 0;
        HIP_TRY(hipMemsetAsync(ws->d_out, 0x3f, (size_t)n_sites * 4 * sizeof(float), hs));
    } else {
        st = dispatch_run(m, ws, hs);
    }
    if (st != CCSM_OK) { ws->r_pending = false; return st; }
    HIP_TRY(hipMemcpyAsync(ws->p_out, ws->d_out, (size_t)n_sites * 4 * sizeof(float), hipMemcpyDeviceToHost, hs));
    HIP_TRY(hipMemcpyAsync(ws->rp_locs, ws->r_locs, (size_t)n_sites * sizeof(int), hipMemcpyDeviceToHost, hs));
    if (ws->r_checked) HIP_TRY(hipMemcpyAsync(at(ws->rp_table, 84), d_ns, (size_t)nr * 4, hipMemcpyDeviceToHost, hs));
    return CCSM_OK;
}

ccsm_status ccsm_wait_reads_host(ccsm_workspace* ws, int32_t* first_site, int32_t* locs, float* logits, float* probs,
                                 int32_t* n_sites_out) {
    if (!ws || !first_site || !locs || !logits || !probs || !n_sites_out)
        return fail(CCSM_ERR_INVALID_ARG, "workspace and every output must be non-NULL");
    if (!ws->r_pending) return fail(CCSM_ERR_INVALID_ARG, "no read chunk in flight on this workspace");
    HIP_TRY(hipSetDevice(ws->device));
    ws->r_pending = false;
    const int nr = ws->r_nreads, n_sites = ws->r_nsites;
    const size_t cr = (size_t)ws->r_cap_reads;
    const int32_t* h_first = reinterpret_cast<const int32_t*>(ws->rp_table + cr * 88);
    HIP_TRY(hipStreamSynchronize(ws->r_stream));
    std::memcpy(first_site, h_first, (size_t)(nr + 1) * 4);
    *n_sites_out = n_sites;
    if (n_sites == 0) return CCSM_OK;
    if (ws->r_checked) {   // the caller's counts against the device's own scan
        const int32_t* h_ns = reinterpret_cast<const int32_t*>(ws->rp_table + cr * 84);
        for (int r = 0; r < nr; ++r)
            if (h_ns[r] != h_first[r + 1] - h_first[r])
                return fail(CCSM_ERR_INVALID_ARG, "site_counts disagree with the reads (read " + std::to_string(r) + ")");
    }
    std::memcpy(locs, ws->rp_locs, (size_t)n_sites * sizeof(int));
    std::memcpy(logits, ws->p_out, (size_t)n_sites * 2 * sizeof(float));
    std::memcpy(probs, ws->p_out + (size_t)n_sites * 2, (size_t)n_sites * 2 * sizeof(float));
    return CCSM_OK;
}

ccsm_status ccsm_forward_reads_host(const ccsm_model* m, ccsm_workspace* ws, const ccsm_reads* rd, const ccsm_h0* h0,
                                    int32_t* first_site, int32_t* locs, float* logits, float* probs, int32_t* n_sites_out,
                                    void* stream) {
    if (!first_site || !locs || !logits || !probs || !n_sites_out)
        return fail(CCSM_ERR_INVALID_ARG, "model, workspace, reads and every output must be non-NULL");
    *n_sites_out = 0;
    ccsm_status st = ccsm_submit_reads_host(m, ws, rd, nullptr, h0, stream);
    if (st != CCSM_OK) return st;
    return ccsm_wait_reads_host(ws, first_site, locs, logits, probs, n_sites_out);
}

ccsm_status ccsm_workspace_set_timing(ccsm_workspace* ws, int enable) {
    if (!ws) return fail(CCSM_ERR_INVALID_ARG, "workspace must be non-NULL");
    ws->timing = enable != 0;
    ws->timed = false;
    ws->ev_runs = 0;
    return CCSM_OK;
}

ccsm_status ccsm_workspace_last_timing(ccsm_workspace* ws, float out_ms[5]) {
    if (!ws || !out_ms) return fail(CCSM_ERR_INVALID_ARG, "workspace and out must be non-NULL");
    if (!ws->timed) return fail(CCSM_ERR_INVALID_ARG, "no timed forward on this workspace");
    HIP_TRY(hipSetDevice(ws->device));
    HIP_TRY(hipEventSynchronize(ws->ev[6]));
    float fin = 0.f;
    HIP_TRY(hipEventElapsedTime(&out_ms[0], ws->ev[1], ws->ev[2]));
    HIP_TRY(hipEventElapsedTime(&out_ms[1], ws->ev[2], ws->ev[3]));
    HIP_TRY(hipEventElapsedTime(&out_ms[2], ws->ev[3], ws->ev[4]));
    HIP_TRY(hipEventElapsedTime(&out_ms[3], ws->ev[4], ws->ev[5]));
    HIP_TRY(hipEventElapsedTime(&fin, ws->ev[5], ws->ev[6]));
    out_ms[4] = fin;
    return CCSM_OK;
}

int ccsm_debug_fp8_e4m3(float v) { return fp8_e4m3_host(v); }
int ccsm_debug_rows_padded(int n_sites) { return n_sites > 0 ? rows_padded(n_sites) : 0; }
int ccsm_debug_rows_capacity(const ccsm_workspace* ws) { return ws ? ws->rows_p : 0; }

ccsm_status ccsm_workspace_timing_mean(ccsm_workspace* ws, float out_ms[5], int* n_runs) {
    if (!ws || !out_ms || !n_runs) return fail(CCSM_ERR_INVALID_ARG, "workspace, out and n_runs must be non-NULL");
    HIP_TRY(hipSetDevice(ws->device));
    const int n = std::min(ws->ev_runs, (int)ccsm_workspace::kEvSets);
    *n_runs = n;
    for (int j = 0; j < 5; ++j) out_ms[j] = 0.f;
    for (int k = 0; k < n; ++k) {
        hipEvent_t* e = ws->evs[k];
        HIP_TRY(hipEventSynchronize(e[6]));
        for (int j = 0; j < 5; ++j) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, e[1 + j], e[2 + j]));
            out_ms[j] += ms / n;
        }
    }
    return CCSM_OK;
}

ccsm_status ccsm_debug_read(ccsm_workspace* ws, int which, void* host_dst, size_t bytes) {
    if (!ws || !host_dst) return fail(CCSM_ERR_INVALID_ARG, "workspace and dst must be non-NULL");
    HIP_TRY(hipSetDevice(ws->device));
    const size_t tiles = ws->rows_p / 32;
    const void* src = nullptr;
    size_t cap = 0;
    switch (which) {
        case 0: src = ws->x0; cap = tiles * kSeqLen * kKB0 * 2 * 1024; break;
        case 1: src = ws->act[0]; cap = tiles * kSeqLen * kKB12 * 2 * 1024; break;
        case 2: src = ws->act[1]; cap = tiles * kSeqLen * kKB12 * 2 * 1024; break;
        case 3: src = ws->h0buf; cap = (size_t)2 * kLayers * ws->rows_p * kHidden * sizeof(float); break;
        case 4: src = ws->part; cap = (size_t)ws->rows_p * 2 * sizeof(float); break;
        case 5: src = ws->dbg; cap = ws->dbg ? (kSeqLen * kWaves * 5 + 2 * kSeqLen * kWaves * 8 * 6) * 8 : 0; break;
        default: return fail(CCSM_ERR_INVALID_ARG, "unknown buffer id");
    }
    if (bytes > cap) return fail(CCSM_ERR_CAPACITY, "debug read larger than the buffer");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_dst, src, bytes, hipMemcpyDeviceToHost));
    return CCSM_OK;
}

ccsm_status ccsm_selftest_split_f8(int device, float* err_corr, float* err_main_only) {
    if (!err_corr || !err_main_only) return fail(CCSM_ERR_INVALID_ARG, "outputs must be non-NULL");
    HIP_TRY(hipSetDevice(device));
    // W: 32 units x 32 k, X: 32 rows x 32 k, deterministic, non-symmetric, weights ~ U(-0.07, 0.07), activations in (-1, 1)
    std::vector<float> w(32 * 32), x(32 * 32);
    uint32_t sd = 12345u;
    auto rnd = [&]() { sd = sd * 1664525u + 1013904223u; return (float)(sd >> 8) * (1.0f / 16777216.0f); };
    for (auto& v : w) v = (rnd() - 0.5f) * 0.14f;
    for (auto& v : x) v = (rnd() - 0.5f) * 1.9f;
    const int lg = corr_scale_log2(w.data(), w.size());
    std::vector<_Float16> frag(2 * 2 * 512, (_Float16)0.f);
    for (int kb = 0; kb < 2; ++kb) {
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) frag[(kb * 2) * 512 + lane * 8 + j] = (_Float16)w[(lane & 31) * 32 + 16 * kb + 8 * (lane >> 5) + j];
        emit_corr_frag(reinterpret_cast<uint8_t*>(frag.data()) + (kb * 2 + 1) * 1024, kb, lg,
                       [&](int i, int k) { return w[i * 32 + k]; });
    }
    uint4* dw = nullptr;
    float *dx = nullptr, *dc = nullptr;
    HIP_TRY(hipMalloc((void**)&dw, 4096));
    HIP_TRY(hipMalloc((void**)&dx, 4096));
    HIP_TRY(hipMalloc((void**)&dc, 4096));
    HIP_TRY(hipMemcpy(dw, frag.data(), 4096, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dx, x.data(), 4096, hipMemcpyHostToDevice));
    float* outs[2] = {err_corr, err_main_only};
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(corr_selftest_kernel, dim3(1), dim3(64), 0, 0, dw, dx, dc, 127 - 11 - lg, pass == 0 ? 1 : 0);
        HIP_TRY(hipGetLastError());
        std::vector<float> c(1024);
        HIP_TRY(hipMemcpy(c.data(), dc, 4096, hipMemcpyDeviceToHost));
        double err = 0.0;
        for (int u = 0; u < 32; ++u)
            for (int r = 0; r < 32; ++r) {
                double ref = 0.0;
                for (int k = 0; k < 32; ++k) ref += (double)w[u * 32 + k] * (double)x[r * 32 + k];
                err = std::fmax(err, std::fabs(ref - (double)c[u * 32 + r]));
            }
        *outs[pass] = (float)err;
    }
    (void)hipFree(dw); (void)hipFree(dx); (void)hipFree(dc);
    return CCSM_OK;
}

ccsm_status ccsm_selftest_split_mx(int device, int fmt, float* err_corr, float* err_main_only, int* blob_mismatch) {
    if (fmt != 2 && fmt != 4) return fail(CCSM_ERR_INVALID_ARG, "fmt must be 2 (fp6 weight blob) or 4 (fp4)");
    if (!err_corr || !err_main_only || !blob_mismatch) return fail(CCSM_ERR_INVALID_ARG, "outputs must be non-NULL");
    HIP_TRY(hipSetDevice(device));
    // W: 32 units x 32 k, X: 32 rows x 32 k, deterministic, non-symmetric, weights ~ U(-0.07, 0.07), activations in (-1, 1)
    std::vector<float> w(32 * 32), x(32 * 32);
    uint32_t sd = 12345u;
    auto rnd = [&]() { sd = sd * 1664525u + 1013904223u; return (float)(sd >> 8) * (1.0f / 16777216.0f); };
    for (auto& v : w) v = (rnd() - 0.5f) * 0.14f;
    for (auto& v : x) v = (rnd() - 0.5f) * 1.9f;
    std::vector<_Float16> frag(4 * 512, (_Float16)0.f);       // hi kb0, hi kb1, blob, scale dwords (byte 0)
    auto get = [&](int i, int k) { return w[i * 32 + k]; };
    emit_hi_frag(frag.data(), 0, get);
    emit_hi_frag(frag.data() + 512, 1, get);
    BlobErr be;
    emit_blob_frag(reinterpret_cast<uint8_t*>(frag.data()) + 2048, reinterpret_cast<uint8_t*>(frag.data()) + 3072 + 256,
                   reinterpret_cast<uint8_t*>(frag.data()) + 3072, 0, fmt, get, &be);
    uint4* dw = nullptr;
    float *dx = nullptr, *dc = nullptr;
    uint32_t* db = nullptr;
    HIP_TRY(hipMalloc((void**)&dw, 4096));
    HIP_TRY(hipMalloc((void**)&dx, 4096));
    HIP_TRY(hipMalloc((void**)&dc, 4096));
    HIP_TRY(hipMalloc((void**)&db, 64 * 24));
    HIP_TRY(hipMemcpy(dw, frag.data(), 4096, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dx, x.data(), 4096, hipMemcpyHostToDevice));
    float* outs[2] = {err_corr, err_main_only};
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(mx_selftest_kernel, dim3(1), dim3(64), 0, 0, dw, dx, dc, db, pass == 0 ? fmt : 0, 0.25f, kMxScaleHi, kMxScaleLo);
        HIP_TRY(hipGetLastError());
        std::vector<float> c(1024);
        HIP_TRY(hipMemcpy(c.data(), dc, 4096, hipMemcpyDeviceToHost));
        double err = 0.0;
        for (int u = 0; u < 32; ++u)
            for (int r = 0; r < 32; ++r) {
                double ref = 0.0;
                for (int k = 0; k < 32; ++k) ref += (double)w[u * 32 + k] * (double)x[r * 32 + k];
                err = std::fmax(err, std::fabs(ref - (double)c[u * 32 + r]));
            }
        *outs[pass] = (float)err;
    }
    // the host's fp6 encoder against the instruction's: the same activations packed here (x_hi * 4 | x_lo * 2^14, kMxPerm order)
    std::vector<uint8_t> dev_blob(64 * 24), host_blob(64 * 24, 0);
    HIP_TRY(hipMemcpy(dev_blob.data(), db, 64 * 24, hipMemcpyDeviceToHost));
    for (int lane = 0; lane < 64; ++lane) {
        const int n = lane & 31, g = lane >> 5;
        uint8_t bits[24] = {0};
        for (int j = 0; j < 32; ++j) {
            const float v = x[n * 32 + mx_perm(j)];
            const float hi = (float)(_Float16)v;
            const float lo12 = (float)(_Float16)((v - hi) * 4096.0f);
            const uint8_t code = mx_code((g ? lo12 : hi) * 4.0f, 2);
            const int bit = j * 6;
            bits[bit >> 3] |= (uint8_t)(code << (bit & 7));
            if ((bit & 7) + 6 > 8) bits[(bit >> 3) + 1] |= (uint8_t)(code >> (8 - (bit & 7)));
        }
        std::memcpy(host_blob.data() + lane * 24, bits, 24);
    }
    int mism = 0;
    for (size_t i = 0; i < dev_blob.size(); ++i) mism += dev_blob[i] != host_blob[i];
    *blob_mismatch = mism;
    (void)hipFree(dw); (void)hipFree(dx); (void)hipFree(dc); (void)hipFree(db);
    return CCSM_OK;
}

ccsm_status ccsm_selftest_mfma(int device, float* max_abs_err) {
    if (!max_abs_err) return fail(CCSM_ERR_INVALID_ARG, "max_abs_err must be non-NULL");
    HIP_TRY(hipSetDevice(device));
    std::vector<_Float16> a(32 * 16), b(16 * 32);
    std::vector<float> c(32 * 32), ref(32 * 32, 0.f);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 20) - 2048) / 512.0f; };
    for (auto& v : a) v = (_Float16)rnd();
    for (auto& v : b) v = (_Float16)rnd();   // asymmetric B (transpose-detecting)
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j)
            for (int k = 0; k < 16; ++k) ref[i * 32 + j] += (float)a[i * 16 + k] * (float)b[k * 32 + j];
    _Float16 *da = nullptr, *db = nullptr;
    float* dc = nullptr;
    HIP_TRY(hipMalloc((void**)&da, a.size() * 2));
    HIP_TRY(hipMalloc((void**)&db, b.size() * 2));
    HIP_TRY(hipMalloc((void**)&dc, c.size() * 4));
    HIP_TRY(hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, 0, da, db, dc);
    HIP_TRY(hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc);
    float me = 0.f;
    for (int i = 0; i < 32 * 32; ++i) me = std::max(me, std::abs(c[i] - ref[i]));
    *max_abs_err = me;
    return CCSM_OK;
}

}  // extern "C"


// =========================================================================================================
// Aggregate mode (config 5)
// =========================================================================================================
struct ccsm_aggr_model {
    int device = 0;
    uint4 *frag_w = nullptr, *frag_att = nullptr;      // MFMA A fragments (ccsm_aggr.hip: Frags)
    float* vec = nullptr;
    float* normals = nullptr;      // seeded torch.randn stream replica
    int64_t n_normals = 0;
    bool only_close = false;       // --only_close: adjacency indicator instead of the distance feature
    // staging for the host-pointer path
    int64_t cap_sites = 0;
    long long* d_pos = nullptr;
    float *d_hist = nullptr, *d_out = nullptr;
};

namespace {
// torch CPU randn after manual_seed(seed) + `skip` 32-bit draws: mt19937 -> 24-bit uniforms -> Box-Muller in blocks of 16
// (ATen/native/cpu/DistributionTemplates.h:140-149, 208-229; ATen/core/TransformationHelper.h:85-88).
void torch_randn_stream(uint64_t seed, int64_t skip, int64_t n, std::vector<float>& out) {
    std::mt19937 gen(static_cast<uint32_t>(seed));       // init_genrand(seed & 0xffffffff)
    gen.discard(static_cast<unsigned long long>(skip));
    const int64_t blocks = (n + 15) / 16;
    out.resize(static_cast<size_t>(blocks) * 16);
    for (int64_t b = 0; b < blocks; ++b) {
        float u[16];
        for (int j = 0; j < 16; ++j) u[j] = static_cast<float>(gen() & ((1u << 24) - 1)) * (1.0f / 16777216.0f);
        for (int j = 0; j < 8; ++j) {
            const float u1 = 1.0f - u[j], u2 = u[j + 8];
            const float radius = std::sqrt(-2.0f * std::log(u1));
            const float theta = 6.283185307179586f * u2;
            out[b * 16 + j] = radius * std::cos(theta);
            out[b * 16 + j + 8] = radius * std::sin(theta);
        }
    }
}
}  // namespace

extern "C" {

ccsm_status ccsm_aggr_create(const ccsm_aggr_weights* w, int device, uint64_t seed, int64_t stream_sites, ccsm_aggr_model** out) {
    if (!w || !out) return fail(CCSM_ERR_INVALID_ARG, "weights and out must be non-NULL");
    *out = nullptr;
    for (int d = 0; d < 2; ++d)
        if (!w->weight_ih[d] || !w->weight_hh[d] || !w->bias_ih[d] || !w->bias_hh[d]) return fail(CCSM_ERR_INVALID_ARG, "NULL rnn tensor");
    if (!w->att_wa || !w->att_ua || !w->att_va || !w->fc1_weight || !w->fc1_bias) return fail(CCSM_ERR_INVALID_ARG, "NULL tensor");
    if (stream_sites <= 0 || stream_sites > (int64_t)1 << 27) return fail(CCSM_ERR_INVALID_ARG, "stream_sites must be in [1, 2^27]");
    HIP_TRY(hipSetDevice(device));
    ccsm_aggr_model* m = new (std::nothrow) ccsm_aggr_model();
    if (!m) return fail(CCSM_ERR_NOMEM, "out of host memory");
    m->device = device;
    using namespace ccsm_aggr;
    ccsm_status st = CCSM_OK;
    {
        // A fragment of a 32 x 16 block of a row-major matrix: lane i + 32 q holds M[i][8 q + 0..7] as eight halfs; every value as
        // fp16 hi + lo, one fragment each
        std::vector<_Float16> fw((size_t)kWFrags * 512), fa((size_t)kAttFrags * 512);
        auto put = [](std::vector<_Float16>& dst, int frag_hi, const std::function<float(int, int)>& at) {
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const HalfPair p = split_host(at(lane & 31, 8 * (lane >> 5) + j));
                    dst[(size_t)frag_hi * 512 + lane * 8 + j] = p.hi;
                    dst[(size_t)(frag_hi + 1) * 512 + lane * 8 + j] = p.lo;
                }
        };
        for (int d = 0; d < 2; ++d)
            for (int g = 0; g < 3; ++g)
                for (int kb = 0; kb < 2; ++kb) {
                    // input part: columns 0..19 = histogram weights, column 20 = the bias the constant-1 input column carries
                    // (b_ih + b_hh for r and z; b_in alone for n, whose b_hn sits inside r * (...))
                    put(fw, ((((d * 2 + 0) * 3 + g) * 2 + kb) * 2), [&](int i, int k8) {
                        const int row = g * H + i, k = 16 * kb + k8;
                        if (k < NB) return w->weight_ih[d][row * F + k];
                        if (k == NB) return w->bias_ih[d][row] + (g < 2 ? w->bias_hh[d][row] : 0.0f);
                        return 0.0f;
                    });
                    put(fw, ((((d * 2 + 1) * 3 + g) * 2 + kb) * 2), [&](int i, int k8) { return w->weight_hh[d][(g * H + i) * H + 16 * kb + k8]; });
                }
        for (int which = 0; which < 2; ++which)
            for (int half = 0; half < 2; ++half)
                for (int kb = 0; kb < 2; ++kb)
                    put(fa, (((which * 2 + half) * 2 + kb) * 2), [&](int i, int k8) {
                        return (which == 0 ? w->att_ua : w->att_wa)[i * 64 + 32 * half + 16 * kb + k8];
                    });
        std::vector<float> vec(kVecFloats, 0.f);
        for (int d = 0; d < 2; ++d) {
            for (int g = 0; g < 3; ++g)
                for (int i = 0; i < H; ++i) vec[d * 128 + g * 32 + i] = w->weight_ih[d][(g * H + i) * F + NB];   // the position feature's column
            for (int i = 0; i < H; ++i) vec[d * 128 + 96 + i] = w->bias_hh[d][2 * H + i];
        }
        std::memcpy(&vec[256], w->att_va, sizeof(float) * H);
        std::memcpy(&vec[288], w->fc1_weight, sizeof(float) * 64);
        vec[352] = w->fc1_bias[0];
        st = upload(&m->frag_w, fw.data(), fw.size() * sizeof(_Float16));
        if (st == CCSM_OK) st = upload(&m->frag_att, fa.data(), fa.size() * sizeof(_Float16));
        if (st == CCSM_OK) st = upload(&m->vec, vec.data(), vec.size() * sizeof(float));
        if (st == CCSM_OK && hipFuncSetAttribute(reinterpret_cast<const void*>(&ccsm_aggr::aggr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)kLdsBytes) != hipSuccess)
            st = fail(CCSM_ERR_HIP, "hipFuncSetAttribute(aggr_kernel)");
    }
    if (st == CCSM_OK) {
        // the reference seeds, THEN constructs AggrAttRNN (call_mods_freq_bam.py:313-322): its parameter initialisation
        // consumes one 32-bit draw per parameter (14 753) before the first h0 is drawn
        const int64_t n_params = 2 * (96 * F + 96 * H + 96 + 96) + 2 * (H * 64) + H + 64 + 1;
        std::vector<float> nrm;
        torch_randn_stream(seed, n_params, stream_sites * 64, nrm);
        m->n_normals = (int64_t)nrm.size();
        st = upload(&m->normals, nrm.data(), nrm.size() * sizeof(float));
    }
    if (st == CCSM_OK && hipDeviceSynchronize() != hipSuccess) st = fail(CCSM_ERR_HIP, "hipDeviceSynchronize at the end of ccsm_aggr_create");   // (as ccsm_create)
    if (st != CCSM_OK) { ccsm_aggr_destroy(m); return st; }
    *out = m;
    return CCSM_OK;
}

void ccsm_aggr_destroy(ccsm_aggr_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipFree(m->frag_w); (void)hipFree(m->frag_att); (void)hipFree(m->vec);
    (void)hipFree(m->normals); (void)hipFree(m->d_pos); (void)hipFree(m->d_hist); (void)hipFree(m->d_out);
    delete m;
}

ccsm_status ccsm_aggr_set_only_close(ccsm_aggr_model* m, int only_close) {
    if (!m) return fail(CCSM_ERR_INVALID_ARG, "model must be non-NULL");
    m->only_close = only_close != 0;
    return CCSM_OK;
}

ccsm_status ccsm_aggr_forward_device(ccsm_aggr_model* m, int64_t n_sites, const int64_t* refposes, const float* histos,
                                     int64_t stream_pos, float* out, void* stream) {
    if (!m || !refposes || !histos || !out) return fail(CCSM_ERR_INVALID_ARG, "NULL argument");
    if (n_sites <= 0 || n_sites > (int64_t)1 << 30) return fail(CCSM_ERR_INVALID_ARG, "n_sites out of range");
    if (stream_pos < 0 || stream_pos + n_sites * 64 > m->n_normals)
        return fail(CCSM_ERR_CAPACITY, "random stream exhausted: create the model with a larger stream_sites");
    HIP_TRY(hipSetDevice(m->device));
    ccsm_aggr::Frags fr{m->frag_w, m->frag_att, m->vec};
    // one wave per tile of 32 sites, four waves (one per SIMD) per workgroup; at most four workgroups per CU's worth of tiles queued
    const int64_t tiles = (n_sites + ccsm_aggr::TILE - 1) / ccsm_aggr::TILE;
    const int grid = (int)std::min<int64_t>((tiles + ccsm_aggr::WAVES - 1) / ccsm_aggr::WAVES, 256 * 4);
    hipLaunchKernelGGL(ccsm_aggr::aggr_kernel, dim3(grid), dim3(256), ccsm_aggr::kLdsBytes, static_cast<hipStream_t>(stream), fr,
                       reinterpret_cast<const long long*>(refposes), histos, m->normals, (long long)stream_pos, (int)n_sites, out,
                       m->only_close ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return CCSM_OK;
}

ccsm_status ccsm_aggr_forward_host(ccsm_aggr_model* m, int64_t n_sites, const int64_t* refposes, const float* histos,
                                   int64_t stream_pos, float* out, void* stream) {
    if (!m || !refposes || !histos || !out) return fail(CCSM_ERR_INVALID_ARG, "NULL argument");
    if (n_sites <= 0) return fail(CCSM_ERR_INVALID_ARG, "n_sites must be > 0");
    HIP_TRY(hipSetDevice(m->device));
    if (n_sites > m->cap_sites) {
        (void)hipFree(m->d_pos); (void)hipFree(m->d_hist); (void)hipFree(m->d_out);
        m->d_pos = nullptr; m->d_hist = nullptr; m->d_out = nullptr; m->cap_sites = 0;
        HIP_TRY(hipMalloc((void**)&m->d_pos, n_sites * sizeof(long long)));
        HIP_TRY(hipMalloc((void**)&m->d_hist, n_sites * 20 * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&m->d_out, n_sites * sizeof(float)));
        m->cap_sites = n_sites;
    }
    hipStream_t hs = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(m->d_pos, refposes, n_sites * sizeof(long long), hipMemcpyHostToDevice, hs));
    HIP_TRY(hipMemcpyAsync(m->d_hist, histos, n_sites * 20 * sizeof(float), hipMemcpyHostToDevice, hs));
    ccsm_status st = ccsm_aggr_forward_device(m, n_sites, reinterpret_cast<const int64_t*>(m->d_pos), m->d_hist, stream_pos, m->d_out, stream);
    if (st != CCSM_OK) return st;
    HIP_TRY(hipMemcpyAsync(out, m->d_out, n_sites * sizeof(float), hipMemcpyDeviceToHost, hs));
    HIP_TRY(hipStreamSynchronize(hs));
    return CCSM_OK;
}

// See include/ccsm.h.  Launches run back to back for `seconds`; only the second half is averaged (the power governor needs
// about a second to settle at the cap).
ccsm_status ccsm_measure_mfma_ceiling(int device, int mode, double seconds, float* tflops, float* issue_gcycles) {
    if (!tflops || mode < 0 || mode > 4 || !(seconds > 0.0) || seconds > 60.0)
        return fail(CCSM_ERR_INVALID_ARG, "mode must be 0 (f16), 1 (mix), 2 (mix, LDS-fed), 3 (f16 on 16x16x32) or 4 (mix on 16-wide instructions), 0 < seconds <= 60, tflops non-NULL");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    const int grid = prop.multiProcessorCount, iters = 20000;
    const std::vector<uint4> h = ccsm_ceiling::random_operands();
    uint4* rnd = nullptr;
    float* out = nullptr;
    HIP_TRY(hipMalloc((void**)&rnd, h.size() * sizeof(uint4)));
    if (hipMalloc((void**)&out, (size_t)grid * 512 * sizeof(float)) != hipSuccess) { (void)hipFree(rnd); return fail(CCSM_ERR_NOMEM, "hipMalloc"); }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ccsm_status st = CCSM_OK;
    auto launch = [&](int it) {
        switch (mode) {
            case 0: hipLaunchKernelGGL((ccsm_ceiling::k<512, 0>), dim3(grid), dim3(512), 0, 0, rnd, out, it); break;
            case 1: hipLaunchKernelGGL((ccsm_ceiling::k<512, 1>), dim3(grid), dim3(512), 0, 0, rnd, out, it); break;
            case 3: hipLaunchKernelGGL((ccsm_ceiling::k16<512, 0>), dim3(grid), dim3(512), 0, 0, rnd, out, it); break;
            case 4: hipLaunchKernelGGL((ccsm_ceiling::k16<512, 1>), dim3(grid), dim3(512), 0, 0, rnd, out, it); break;
            default: hipLaunchKernelGGL((ccsm_ceiling::k<512, 2>), dim3(grid), dim3(512), 0, 0, rnd, out, it); break;
        }
    };
    double ms_sum = 0.0;
    long launches = 0;
    do {
        if (hipMemcpy(rnd, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
            hipEventCreate(&e1) != hipSuccess) { st = fail(CCSM_ERR_HIP, "ceiling probe set-up"); break; }
        launch(200);
        if (hipDeviceSynchronize() != hipSuccess) { st = fail(CCSM_ERR_HIP, "ceiling probe launch"); break; }
        const auto t0 = std::chrono::steady_clock::now();
        auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
        while (elapsed() < seconds) {
            (void)hipEventRecord(e0, 0);
            for (int r = 0; r < 4; ++r) launch(iters);
            (void)hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess) { st = fail(CCSM_ERR_HIP, "ceiling probe run"); break; }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (elapsed() > 0.5 * seconds) { ms_sum += ms; launches += 4; }
        }
    } while (false);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(rnd); (void)hipFree(out);
    if (st != CCSM_OK) return st;
    if (launches == 0) return fail(CCSM_ERR_INVALID_ARG, "seconds too short for one averaged launch");
    const double waves = (double)grid * 8, sec = ms_sum * 1e-3;
    // per iteration of either kernel: the flops of 8 v_mfma_f32_32x32x16_f16 (k16: 16 v_mfma_f32_16x16x32_f16 on 8 accumulators x 2 groups)
    *tflops = (float)(launches * waves * iters * (8.0 * 2 * 32 * 32 * 16) / sec * 1e-12);
    const bool mixed = mode == 1 || mode == 2 || mode == 4;
    if (issue_gcycles) *issue_gcycles = (float)(launches * (waves / (grid * 4.0)) * iters * (8 * 32 + (mixed ? 4 * 35 : 0)) / sec * 1e-9);
    return CCSM_OK;
}

}  // extern "C"
