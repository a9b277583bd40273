// libccsm GRU layers in "split-mx" arithmetic: f16 main product + ONE block-scaled MX correction product per pair of k-blocks.
// Included by ccsm_api.hip after ccsm_kernels.hip and ccsm_gru_f8.hip (fp8 helpers, the attention kernel).
//
// Arithmetic.  Every fp32 operand v is carried as hi = fp16(v) plus the residual lo = v - hi.  The product
//     W x  =  W_hi x_hi  +  W_lo x_hi  +  W_hi x_lo  (+ W_lo x_lo, dropped: 2^-22 relative)
// is issued per pair of k-blocks (32 k) as
//     main : 2 x v_mfma_f32_32x32x16_f16 on (W_hi, x_hi)
//     corr : 1 x v_mfma_scale_f32_32x32x64_f8f6f4, K = 64 = [32 k of W_lo x_hi | 32 k of W_hi x_lo], activations (B) in
//            fp6 e2m3, weights (A) in fp4 e2m1 (recurrent part; input part of the r and z gates) or fp6 e2m3 (input part of the n
//            gate: fp4 there shifts every site the same way, ~1e-5), fp32 accumulation into the SAME accumulator.
// Why not fp8 operands (round 1): the GRU kernels run at the 1400 W package power cap (profiles/r02_c_power_attribution.md), so
// their time is their energy; the fp8 x fp8 correction MFMAs cost 16 % of it, fp6 x fp6 half of that, fp4 weights x fp6
// activations almost none, at 4-7e-6 max |dprob| (tests/diag/emulate_corr_formats.py; fp8: 4e-6, bar 1e-4).
//
// Correction operands ("blobs").  One MX operand of one lane = 32 values = one block with one E8M0 scale:
//   lane (n, g = 0): W_lo[n][k] * 2^11 (A)  /  x_hi[n][k] (B)          lane (n, g = 1): W_hi[n][k] (A)  /  x_lo[n][k] (B)
//   for the 32 k of the pair in the order kMxPerm[j] = 8 ((j & 15) >> 2) + (j & 3) + 4 (j >> 4)  — what eight
//   v_permlane32_swap and one v_cvt_scalef32_pk32_fp6_f16 make of the MFMA C layout in the step epilogue; the host packs the
//   weights in the same order.  An ACTIVATION blob (fp6: 24 bytes per lane) travels in the two 1 KiB "corr" fragments the fp8
//   scheme had (nothing else about fragments, transfers or LDS changes): bytes 0-15 of a lane in the fragment of the pair's first
//   k-block, bytes 16-23 in the first half of the second one's.  A WEIGHT blob (fp4: 16 bytes per lane) is one 1 KiB fragment = the
//   instruction's A operand as loaded (a 24-byte fp6 weight blob would be a 6-register operand assembled from two loads: the
//   compiler copies it together behind an s_waitcnt vmcnt(0) per load); the lane's E8M0 scale bytes of a pair's gates share one
//   dword (256 B per pair, byte g = gate g, picked by the instruction's op_sel), chosen by the host PER (row, 32-k block) from the
//   block's largest magnitude (heavy-tailed matrices keep their small entries' corrections).  Activations use fixed scales: GRU
//   outputs lie in (-1, 1) (x_hi * 4, x_lo * 2^14); the initial states (any magnitude up to 15) are packed 8x coarser and multiplied
//   with their own scale in the first step.
// The recurrent state is carried as fp16 hi + fp8 lo: a wave keeps the fp8 residuals of its own 32 units (MFMA C layout) in LDS
// next to the fragments (the blend h' = n + z (h - n) needs h_{t-1} itself, and an fp6 blob cannot be picked apart cheaply).
#include <hip/hip_runtime.h>

namespace ccsm {

constexpr int kMxWFmtH = 4;                        // A (weight) operand format of the correction MFMA, recurrent part (phase B): fp4 e2m1
constexpr int kMxWFmtX = 2;                        // ... input part of the n gate (phase C): fp6 e2m3 — fp4 THERE biases every site the same way (~1e-5);
                                                   // the input part of the r and z gates (phase A) takes fp4 like the recurrent part
                                                   // (tests/diag/emulate_corr_formats.py: 4.9-6.4e-6 max |dprob| either way)
constexpr int kMxBFmt = 2;                         // B (activation) operand: fp6 e2m3
constexpr int kMxScaleHi = 127 - 2;                // x_hi blob holds x_hi * 4
constexpr int kMxScaleLo = 127 - 14;               // x_lo blob holds x_lo * 2^14
constexpr int kMxScaleHi0 = 127 + 2;               // initial states: x_hi / 4 (clamped to |h0| <= 30)
constexpr int kMxScaleLo0 = 127 - 10;              //                 x_lo * 2^10
constexpr float kMxH0Div = 4.0f;                   // what the initial-state blobs are divided by (GRU outputs: 0.25)
constexpr float kMxLoScale = 65536.0f;             // a wave's private fp8 residuals carry lo * 2^16: |lo| <= 2^-8 (|h| < 16) stays below e4m3's 448.
                                                   // |h_t| <= max(1, |h0|): the state is NOT confined to (-1, 1) when the initial states are not

// Diagnostic builds of the layer-1/2 kernel (results wrong on purpose; tools/ab_build.sh <name> -DCCSM_MX_DIAG=<bits>): what one consumer costs -
// the decomposition of DESIGN 10's "everything else".  1: the gate math's sigmoids and tanh replaced by multiplies; 2: no barrier in the pairs of
// phases A and C (the MFMAs, requests and waits stay); 4: no LDS reads of B operands in any phase (the MFMAs take what the registers hold);
// 16: no ring refills; 32: no counted waits.  (Weight requests on L1-resident fragments: -DCCSM_PWR_W1, round 4.)
#ifndef CCSM_MX_DIAG
#define CCSM_MX_DIAG 0
#endif
constexpr int kMxDiag = CCSM_MX_DIAG;
__device__ __forceinline__ float diag_sigmoid(float x) { return (kMxDiag & 1) ? x * 0.25f : sigmoid_f(x); }
typedef _Float16 half32 __attribute__((ext_vector_type(32)));
typedef int i32x6 __attribute__((ext_vector_type(6)));

// ---- "split-mx-d" (DYN): fp6 instead of fp4 recurrent weight blobs, and activation blobs with one E8M0 scale per (row, 32-k block)
// instead of the fixed x_hi * 4.  On TRAINED checkpoints the fixed scale is what gives split-mx its tail (state values below 0.25 are fp6
// subnormals there; together with the fp4 recurrent weights: 0.2-1.7e-4 max |dprob|; with fp6 weights and per-block scales 1.3-4.7e-5 over
// 8192 sites, profiles/r03_y_dynamic_scale_emulation.log and r03_w_split_mx_d_trained_checkpoints.log).
//   * The STATE's block scale needs no storage (the LDS is full): it is a function of the block's largest |x_hi| as fp16, and every lane
//     that reads a blob also holds 16 of the block's 32 fp16 hi values (its hi fragments of the pair's two k-blocks), its half-wave
//     partner the other 16 - the writer (step tail) and every reader derive the same exponent E (below).
//   * For the NEXT layer's input part (XD = DYN or the hybrid: every arithmetic but plain split-mx) the same blob goes to HBM with the
//     lane's scale byte in the spare bytes 8-11 of the pair's second corr fragment - byte bt = row tile bt of the writing workgroup, bytes
//     0..bt filled by tile bt, so the reader takes ONE dword (the last tile's) per pair and the instruction's op_sel picks the byte.
// The exponent:
//     m = max |x_hi| (fp16 bits e, mantissa f):  E = max(e, 1) - 17 + (f > 0x380)      so that  |x_hi| / 2^E <= 7.5  (fp6 e2m3's top)
//     hi lanes (g = 0) hold x_hi / 2^E, scale byte 127 + E;   lo lanes (g = 1) hold x_lo / 2^(E - 11), scale byte 127 + E - 11
//     (|x_lo| <= half an ulp of the block's largest value = 2^(e - 26): at most 4 after the division).
// max |.| of the eight fp16 of one fragment lane and of `prev` (an fp16 in the low half): four v_max3_f16 with |.| source modifiers
// and half selectors (masking and a packed-max tree cost 7 + 3 for the final unpack).  Writer and readers use the same routine.
__device__ __forceinline__ uint32_t absmax8(uint4 f, uint32_t prev) {
    uint32_t a, b, c, m;
    asm("v_max3_f16 %0, |%1|, |%1|, |%2| op_sel:[0,1,0,0]" : "=v"(a) : "v"(f.x), "v"(f.y));     // x.lo x.hi y.lo
    asm("v_max3_f16 %0, |%1|, |%2|, |%2| op_sel:[1,0,1,0]" : "=v"(b) : "v"(f.y), "v"(f.z));     // y.hi z.lo z.hi
    asm("v_max3_f16 %0, |%1|, |%1|, %2 op_sel:[0,1,0,0]" : "=v"(c) : "v"(f.w), "v"(a));         // w.lo w.hi a
    asm("v_max3_f16 %0, %1, %2, %3" : "=v"(m) : "v"(b), "v"(c), "v"(prev));
    return m;                                                          // (the high half is not defined)
}
// m = max |.| of this lane's 16 values as fp16 bits (low half); returns E of the (row, block): the partner lane (n, 1 - g) holds the other 16
__device__ __forceinline__ int dyn_block_exp(uint32_t m16) {
    uint32_t m = m16 & 0xffffu;                                        // (positive fp16 bit patterns order like unsigned integers)
    uint32_t x = m, y = m;
    swap32(x, y);                                                      // lower lanes: y = partner's m; upper lanes: x = partner's m
    m = x > y ? x : y;
    const int e = (int)(m >> 10);
    return (e > 1 ? e : 1) - 17 + ((m & 0x3ffu) > 0x380u ? 1 : 0);
}

// correction MFMA of one pair: fp4 weight blob w (16 bytes per lane), its scale = byte G of ws, activation blob (x0, x1)
// (SBB: the byte of scale_b the instruction takes)
template <int G, int SBB = 0>
__device__ __forceinline__ f32x16 mfma_corr_mx(uint4 w, uint32_t ws, uint4 x0, uint2 x1, f32x16 c, int scale_b) {
    const i32x8 a = {(int)w.x, (int)w.y, (int)w.z, (int)w.w, 0, 0, 0, 0};
    const i32x8 b = {(int)x0.x, (int)x0.y, (int)x0.z, (int)x0.w, (int)x1.x, (int)x1.y, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, kMxWFmtH, kMxBFmt, G, (int)ws, SBB, scale_b);
}
// the same with an fp6 weight blob (24 bytes per lane: w0 | w1)
template <int G, int SBB = 0>
__device__ __forceinline__ f32x16 mfma_corr_mx6(uint4 w0, uint2 w1, uint32_t ws, uint4 x0, uint2 x1, f32x16 c, int scale_b) {
    const i32x8 a = {(int)w0.x, (int)w0.y, (int)w0.z, (int)w0.w, (int)w1.x, (int)w1.y, 0, 0};
    const i32x8 b = {(int)x0.x, (int)x0.y, (int)x0.z, (int)x0.w, (int)x1.x, (int)x1.y, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, kMxWFmtX, kMxBFmt, G, (int)ws, SBB, scale_b);
}

// 32 fp16 values (16 packed registers) / scale -> one fp6 blob
__device__ __forceinline__ void blob_of(const uint32_t (&p)[16], float scale, uint4& c0, uint2& c1) {
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    const u32x16 v = {p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15]};
    const i32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(half32, v), scale);
    c0 = make_uint4((uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3]);
    c1 = make_uint2((uint32_t)r[4], (uint32_t)r[5]);
}

// The wave's 32 units of one batch row in MFMA C layout, as lane (n, hh) holds them: v[4q + e] = unit 8q + 4hh + e.
//   hi0, hi1 : this lane's 16 bytes of the hi fragments of the wave's two k-blocks
//   c0, c1   : this lane's activation blob (lower lanes: x_hi of all 32 units, upper lanes: x_lo)
//   lo8      : fp8 (x 2^16) residuals of this lane's own 16 values, C-layout order (the wave's private copy)
// SCALE = what the blob's values are divided by (0.25 for GRU outputs; kMxH0Div for initial states).
// DYNB: the blob carries the block's own scale (dyn_block_exp of the 32 hi values) instead: *dscale = this lane's E8M0 scale (byte 0 of the
// dword the MX instruction takes: 127 + E in the hi lanes, 127 + E - 11 in the lo lanes); `scale` is then unused.
template <bool CLAMP, bool DYNB = false>
__device__ __forceinline__ void pack_pair_mx(const float (&v)[16], float scale, uint4& hi0, uint4& hi1, uint4& c0, uint2& c1, uint4& lo8,
                                             uint32_t* dscale = nullptr, int upper = 0 /* lane >> 5 (DYNB) */) {
    static_assert(!(CLAMP && DYNB), "a block-scaled blob needs no clamp");
    typedef _Float16 half2p __attribute__((ext_vector_type(2)));
    uint32_t hp[8], hq[8], lp[8];     // hi (fragments), hi as it goes into the blob, lo * 2^12
    float lf[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const half2p h = {(_Float16)v[2 * j], (_Float16)v[2 * j + 1]};
        hp[j] = __builtin_bit_cast(uint32_t, h);
        lf[2 * j] = v[2 * j] - (float)h[0];
        lf[2 * j + 1] = v[2 * j + 1] - (float)h[1];
        float l0 = lf[2 * j] * 4096.0f, l1 = lf[2 * j + 1] * 4096.0f;                            // |lo| <= 2^-12 |v|: no fp16 underflow
        if constexpr (CLAMP) {      // nothing may exceed the fp6 range after the division by `scale`: an overflowing conversion poisons the
                                    // product (NaN probabilities were observed for |h0| >= 8 with the residual at the top of its range)
            const float top = 7.5f * scale;
            const half2p hc = {(_Float16)fminf(fmaxf(v[2 * j], -top), top), (_Float16)fminf(fmaxf(v[2 * j + 1], -top), top)};
            hq[j] = __builtin_bit_cast(uint32_t, hc);
            l0 = fminf(fmaxf(l0, -top), top);
            l1 = fminf(fmaxf(l1, -top), top);
        } else {
            hq[j] = hp[j];
        }
        lp[j] = pack2((_Float16)l0, (_Float16)l1);
    }
    {   // residuals as fp8: clamped first (the conversion does not saturate: an overflow is a NaN that would stay in the state)
        uint32_t r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float c[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = __builtin_amdgcn_fmed3f(lf[4 * q + e], -kF8Clamp / kMxLoScale, kF8Clamp / kMxLoScale);
            short2v t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_bit_cast(short2v, hp[2 * q]), c[0], c[1], 1.0f / kMxLoScale, false);
            t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, c[2], c[3], 1.0f / kMxLoScale, true);
            r[q] = __builtin_bit_cast(uint32_t, t);
        }
        lo8 = make_uint4(r[0], r[1], r[2], r[3]);
    }
    {   // hi fragments: lane (n, g) <- k = 16 kb + 8 g + j
        uint32_t a0 = hp[0], a1 = hp[1], b0 = hp[2], b1 = hp[3];
        swap32(a0, b0);
        swap32(a1, b1);
        hi0 = make_uint4(a0, a1, b0, b1);
        a0 = hp[4]; a1 = hp[5]; b0 = hp[6]; b1 = hp[7];
        swap32(a0, b0);
        swap32(a1, b1);
        hi1 = make_uint4(a0, a1, b0, b1);
    }
    if constexpr (DYNB) {
        // the block's own scale from its 32 fp16 hi values; the lo lanes hold lo * 2^12 here and want x_lo / 2^(E - 11) = that / 2^(E + 1)
        // (the blob conversion divides by `scale`)
        const uint32_t pm = absmax8(make_uint4(hp[4], hp[5], hp[6], hp[7]), absmax8(make_uint4(hp[0], hp[1], hp[2], hp[3]), 0u));
        const int E = dyn_block_exp(pm);
        scale = __builtin_bit_cast(float, (uint32_t)(127 + E + upper) << 23);
        *dscale = (uint32_t)(127 + E - 11 * upper);
    }
    // blob: lower lanes end up with (own hi | partner's hi), upper lanes with (partner's lo | own lo): kMxPerm order
#pragma unroll
    for (int j = 0; j < 8; ++j) swap32(hq[j], lp[j]);
    const uint32_t p[16] = {hq[0], hq[1], hq[2], hq[3], hq[4], hq[5], hq[6], hq[7], lp[0], lp[1], lp[2], lp[3], lp[4], lp[5], lp[6], lp[7]};
    blob_of(p, scale, c0, c1);
}

// The same 16 values as fp16 hi and fp16 lo fragments of the wave's two k-blocks (hybrid arithmetic: the recurrent operand).
__device__ __forceinline__ void pack_pair_hl(const float (&v)[16], uint4& hi0, uint4& hi1, uint4& lo0, uint4& lo1) {
    typedef _Float16 half2p __attribute__((ext_vector_type(2)));
    uint32_t hp[8], lp[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const half2p h = {(_Float16)v[2 * j], (_Float16)v[2 * j + 1]};
        hp[j] = __builtin_bit_cast(uint32_t, h);
        lp[j] = pack2((_Float16)(v[2 * j] - (float)h[0]), (_Float16)(v[2 * j + 1] - (float)h[1]));
    }
    auto frag = [](uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1) {      // lane (n, g) <- k = 16 kb + 8 g + j
        swap32(a0, b0);
        swap32(a1, b1);
        return make_uint4(a0, a1, b0, b1);
    };
    hi0 = frag(hp[0], hp[1], hp[2], hp[3]);
    hi1 = frag(hp[4], hp[5], hp[6], hp[7]);
    lo0 = frag(lp[0], lp[1], lp[2], lp[3]);
    lo1 = frag(lp[4], lp[5], lp[6], lp[7]);
}

// LDS byte offsets shared by the two kernels: h fragments [kb 16][bt 3][hi | corr] x 1 KiB at 0 (96 KiB).  The fp8 residuals of a
// wave's own units (16 B per lane and batch tile) live half in the unused second half of the lane's slot in the corr fragment
// of the wave's second k-block (bytes 8-15) and half in a 12 KiB region at LO_OFF: [wave][bt][lane] x 8 B.
// NB = row tiles per workgroup: 3 (96 rows) in full launches; 2 (64 rows) where 96-row workgroups would leave compute units idle (a lone
// batch, a ragged group): the same code with two thirds of the rows per weight stream (ccsm_api.hip picks per launch).
constexpr int kMxNB = 3;
// Phase C's second pass over x_t runs in REVERSE pair order (15 .. 0): a cyclic pass over more bytes than the XCD's L2 holds (32
// workgroups x 192 KiB of x_t beside 2.4 MB of weights in 4 MiB) finds none of them again under LRU, a zig-zag pass finds the most
// recent ones.  +0.7 % (three alternating A/B runs, profiles/r03_s_zigzag.log); the host packs the n gate's input weights in the same
// order.  -DCCSM_NO_ZIGZAG builds the forward order.
#ifdef CCSM_NO_ZIGZAG
constexpr bool kMxZigZag = false;
#else
constexpr bool kMxZigZag = true;
#endif
constexpr int mx_hbytes(int nb) { return kKBH * nb * 2 * 1024; }
template <int NB = kMxNB>
__device__ __forceinline__ int mx_hfrag(int kb, int bt, int f) { return ((kb * NB + bt) * 2 + f) << 10; }

// ---- h0 -> LDS: hi fragments, blobs (coarse scale) and residuals of this wave's own two k-blocks, every batch tile
template <bool HS3, bool DYN = false, int NB = kMxNB>
__device__ __forceinline__ void mx_h0_to_lds(char* smem, int lo_off, const float* __restrict__ h0d, int tile0, int wave, int lane) {
    const int n = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
        const float* src = h0d + ((size_t)(tile0 + bt) * 32 + n) * kHidden + 32 * wave;
        float v[16];                                                // C layout: v[4q + e] = unit 8q + 4hh + e
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(src + 8 * q + 4 * hh);
            v[4 * q + 0] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
        if constexpr (HS3) {            // exact: fp16 hi and fp16 lo fragments, any magnitude
            uint4 hi0, hi1, lo0, lo1;
            pack_pair_hl(v, hi0, hi1, lo0, lo1);
            *reinterpret_cast<uint4*>(smem + mx_hfrag<NB>(2 * wave, bt, 0) + lane * 16) = hi0;
            *reinterpret_cast<uint4*>(smem + mx_hfrag<NB>(2 * wave + 1, bt, 0) + lane * 16) = hi1;
            *reinterpret_cast<uint4*>(smem + mx_hfrag<NB>(2 * wave, bt, 1) + lane * 16) = lo0;
            *reinterpret_cast<uint4*>(smem + mx_hfrag<NB>(2 * wave + 1, bt, 1) + lane * 16) = lo1;
            continue;
        }
        uint4 hi0, hi1, c0, lo8;
        uint2 c1;
        if constexpr (DYN) {            // the blob carries its own scale: any magnitude, no coarse first-step scale
            uint32_t sc;
            pack_pair_mx<false, true>(v, 0.25f, hi0, hi1, c0, c1, lo8, &sc, hh);
        } else
        pack_pair_mx<true>(v, kMxH0Div, hi0, hi1, c0, c1, lo8);
        *reinterpret_cast<uint4*>(smem + mx_hfrag<NB>(2 * wave, bt, 0) + lane * 16) = hi0;
        *reinterpret_cast<uint4*>(smem + mx_hfrag<NB>(2 * wave + 1, bt, 0) + lane * 16) = hi1;
        *reinterpret_cast<uint4*>(smem + mx_hfrag<NB>(2 * wave, bt, 1) + lane * 16) = c0;
        *reinterpret_cast<uint4*>(smem + mx_hfrag<NB>(2 * wave + 1, bt, 1) + lane * 16) = make_uint4(c1.x, c1.y, lo8.x, lo8.y);
        *reinterpret_cast<uint2*>(smem + lo_off + ((wave * NB + bt) * 64 + lane) * 8) = make_uint2(lo8.z, lo8.w);
    }
}

// ---- step tail: n = tanh(N); h' = n + z (h_{t-1} - n) for this wave's own units; fragments and blobs for the next step (LDS)
// and the next layer (HBM).  accz = sigmoid(Z) already, accn = N.  OUT_FP8: the layer feeding the attention kernel writes fp8 corr
// fragments (attn_fc_f8_kernel's format) instead of blobs.  t16 = lane * 16 (an opaque copy: see lane16_here in the kernels).
template <bool OUT_FP8, bool HS3, bool DYN = false, int NB = kMxNB>
__device__ __forceinline__ void mx_tail(char* smem, int lo_off, const f32x16 (&accz)[NB], const f32x16 (&accn)[NB], uint4* __restrict__ out,
                                        int tile0, int t, int dir, int wave, int t16) {
    constexpr bool XD = HS3 || DYN;     // block-scaled blobs for the next layer's input part (every arithmetic but plain split-mx)
    const int own_off = wave * (2 * NB * 2 * 1024);                          // mx_hfrag(2 wave, 0, 0)
    char* t_wr = smem + (own_off + t16);                                        // + lane * 16       (fragment writes)
    const char* t_rd = smem + (own_off + (t16 & 0x1f0) + ((t16 >> 9) << 3));    // + n * 16 + hh * 8 (own-unit reads, C layout)
    char* t_lo = smem + (lo_off + wave * (NB * 64 * 8) + (t16 >> 1));        // + lane * 8
    auto own_frag = [&](int kbl, int bt, int f) -> int { return ((kbl * NB + bt) * 2 + f) << 10; };
    uint32_t bsc_all = 0;
#ifdef CCSM_PWR_TAIL1       // diagnostic build (results wrong on purpose): the tail's arithmetic (tanh, blend, packing) for row tile 0 only, its results
                           // written for all three row tiles - LDS writes and output stores as shipped: what the tail's vector work costs
    float hn[16];
    uint4 hi0, hi1, c0, lo8;
    uint2 c1;
    uint32_t bsc = 0;
#endif
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
#ifdef CCSM_PWR_TAIL1
        if (bt == 0) {
#else
        float hn[16];
#endif
        uint2 la = make_uint2(0, 0), lb = make_uint2(0, 0);
        if constexpr (!HS3) {
            la = *reinterpret_cast<const uint2*>(t_wr + own_frag(1, bt, 1) + 8);            // residuals of values 0..7
            lb = *reinterpret_cast<const uint2*>(t_lo + bt * (64 * 8));                     // 8..15
        }
        const uint32_t l8[4] = {la.x, la.y, lb.x, lb.y};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const half4 hi = as_half4(*reinterpret_cast<const uint2*>(t_rd + own_frag(q >> 1, bt, 0) + 512 * (q & 1)));
            float hp[4];
            if constexpr (HS3) {        // h_{t-1} = fp16 hi + fp16 lo, both from this wave's own fragments
                const half4 lo = as_half4(*reinterpret_cast<const uint2*>(t_rd + own_frag(q >> 1, bt, 1) + 512 * (q & 1)));
#pragma unroll
                for (int e = 0; e < 4; ++e) hp[e] = (float)hi[e] + (float)lo[e];
            } else {
                const int lo4 = (int)l8[q];
                hp[0] = (float)hi[0] + __builtin_amdgcn_cvt_f32_fp8(lo4, 0) * (1.0f / kMxLoScale);
                hp[1] = (float)hi[1] + __builtin_amdgcn_cvt_f32_fp8(lo4, 1) * (1.0f / kMxLoScale);
                hp[2] = (float)hi[2] + __builtin_amdgcn_cvt_f32_fp8(lo4, 2) * (1.0f / kMxLoScale);
                hp[3] = (float)hi[3] + __builtin_amdgcn_cvt_f32_fp8(lo4, 3) * (1.0f / kMxLoScale);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float nn = (kMxDiag & 1) ? accn[bt][4 * q + e] * 0.001f : tanh_fold(accn[bt][4 * q + e]);
                hn[4 * q + e] = (hp[e] - nn) * accz[bt][4 * q + e] + nn;
            }
        }
#ifndef CCSM_PWR_TAIL1
        uint4 hi0, hi1, c0, lo8;
        uint2 c1;
        uint32_t bsc = 0;               // XD: this lane's E8M0 scale of the blob (the next layer reads it beside the blob)
#endif
        // XD: ONE block-scaled blob, for the own state (DYN) and for the next layer (`upper` from the opaque lane copy: the hoisted
        // 110 + upper got spilled)
        if constexpr (XD) pack_pair_mx<false, true>(hn, 0.25f, hi0, hi1, c0, c1, lo8, &bsc, t16 >> 9);
        else pack_pair_mx<false>(hn, 0.25f, hi0, hi1, c0, c1, lo8);
#ifdef CCSM_PWR_TAIL1
        } else {                    // (the accumulators of the other row tiles stay alive: their MFMAs and sigmoids must not be optimised away)
#pragma unroll
            for (int r = 0; r < 16; r += 8) {
                asm volatile("" :: "v"(accn[bt][r]), "v"(accn[bt][r + 1]), "v"(accn[bt][r + 2]), "v"(accn[bt][r + 3]), "v"(accn[bt][r + 4]), "v"(accn[bt][r + 5]), "v"(accn[bt][r + 6]), "v"(accn[bt][r + 7]));
                asm volatile("" :: "v"(accz[bt][r]), "v"(accz[bt][r + 1]), "v"(accz[bt][r + 2]), "v"(accz[bt][r + 3]), "v"(accz[bt][r + 4]), "v"(accz[bt][r + 5]), "v"(accz[bt][r + 6]), "v"(accz[bt][r + 7]));
            }
        }
#endif
        const uint4 c1w = make_uint4(c1.x, c1.y, lo8.x, lo8.y);
        *reinterpret_cast<uint4*>(t_wr + own_frag(0, bt, 0)) = hi0;
        *reinterpret_cast<uint4*>(t_wr + own_frag(1, bt, 0)) = hi1;
        if constexpr (HS3) {            // the recurrent operand of the next step: fp16 lo fragments (the blob only travels to the next layer)
            uint4 h0_, h1_, lo0, lo1;
            pack_pair_hl(hn, h0_, h1_, lo0, lo1);
            *reinterpret_cast<uint4*>(t_wr + own_frag(0, bt, 1)) = lo0;
            *reinterpret_cast<uint4*>(t_wr + own_frag(1, bt, 1)) = lo1;
        } else {
            *reinterpret_cast<uint4*>(t_wr + own_frag(0, bt, 1)) = c0;
            *reinterpret_cast<uint4*>(t_wr + own_frag(1, bt, 1)) = c1w;
            *reinterpret_cast<uint2*>(t_lo + bt * (64 * 8)) = make_uint2(lo8.z, lo8.w);
        }
        // streaming stores: the next reader is another kernel 0.5 GB later, keep the L2 for the weight stream
        char* o = reinterpret_cast<char*>(out + (((size_t)(tile0 + bt) * kSeqLen + t) * kKB12 + (dir * kKBH + 2 * wave)) * 2 * kFragU4) + (uint32_t)t16;   // uniform base + 32-bit lane offset
        nt_store(hi0, reinterpret_cast<uint4*>(o));
        nt_store(hi1, reinterpret_cast<uint4*>(o + 2048));
        if constexpr (OUT_FP8) {
            // For the attention pool: ONE compact residual fragment per pair instead of two [fp8 x_hi | fp8 x_lo] corr fragments - lane
            // (n, g) holds the 16 fp8 residuals (x 2^17) of k-block g of the pair, byte j <-> k = kCorrPerm[j]; the fp8 copy of x_hi is
            // derivable from the hi fragments and attn_fc_f8_kernel re-derives it in LDS: 3 instead of 4 bytes per element cross HBM twice.
            // This lane's values: hn[4q + e] = unit 8q + 4hh + e, i.e. q = 0, 1 -> k-block 0 (units 0-3, 8-11 (+ 4hh)), q = 2, 3 -> k-block 1.
            typedef _Float16 half2p __attribute__((ext_vector_type(2)));
            uint32_t lq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float lf[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = hn[4 * q + e];
                    lf[e] = __builtin_amdgcn_fmed3f(v - (float)(_Float16)v, -kF8Clamp / kCorrActLo, kF8Clamp / kCorrActLo);
                }
                lq[q] = cvt4_fp8_l(hi0.x, lf[0], lf[1], lf[2], lf[3]);
            }
            // lower lanes keep k-block 0 (own units 0-3, 8-11; the partner's 4-7, 12-15), upper lanes k-block 1
            swap32(lq[0], lq[2]);       // lower: lq0 = own kb0 0-3, lq2 = partner's kb0 4-7 ; upper: lq0 = partner's kb1 0-3, lq2 = own kb1 4-7
            swap32(lq[1], lq[3]);       // lower: lq1 = own kb0 8-11, lq3 = partner's kb0 12-15 ; upper: lq1 = partner's kb1 8-11, lq3 = own kb1 12-15
            nt_store(make_uint4(lq[0], lq[1], lq[2], lq[3]), reinterpret_cast<uint4*>(o + 1024));
        } else {
            nt_store(c0, reinterpret_cast<uint4*>(o + 1024));
            // (bytes 8-15 are spare in HBM: the scales of row tiles 0..bt in bytes 0..bt of one dword - the reader takes the last tile's, one
            // register for the three row tiles, the instruction's op_sel picks the byte)
            if constexpr (XD) bsc_all |= bsc << (8 * bt);
            nt_store(XD ? make_uint4(c1.x, c1.y, bsc_all, 0u) : c1w, reinterpret_cast<uint4*>(o + 3072));
        }
    }
}

// Three-pass split-fp16 arithmetic (F3: ccsm_gru_f3.hip, and gru_layer0_mx_kernel<.., HS3, .., F3>).  Step tail: n = tanh(N); h' = n + z (h_{t-1} - n) for this wave's own 32 units; fp16 hi + lo fragments for the next step (LDS) and the
// next layer / the attention pool (HBM).  accz = sigmoid(Z) already, accn = N.
template <int NB>
__device__ __forceinline__ void f3_tail(char* smem, const f32x16 (&accz)[NB], const f32x16 (&accn)[NB], uint4* __restrict__ out, int tile0,
                                        int t, int dir, int wave, int t16) {
    const int own_off = wave * (2 * NB * 2 * 1024);                          // mx_hfrag(2 wave, 0, 0)
    char* t_wr = smem + (own_off + t16);                                        // + lane * 16       (fragment writes)
    const char* t_rd = smem + (own_off + (t16 & 0x1f0) + ((t16 >> 9) << 3));    // + n * 16 + hh * 8 (own-unit reads, C layout)
    auto own_frag = [&](int kbl, int bt, int f) -> int { return ((kbl * NB + bt) * 2 + f) << 10; };
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
        float hn[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const half4 hi = as_half4(*reinterpret_cast<const uint2*>(t_rd + own_frag(q >> 1, bt, 0) + 512 * (q & 1)));
            const half4 lo = as_half4(*reinterpret_cast<const uint2*>(t_rd + own_frag(q >> 1, bt, 1) + 512 * (q & 1)));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float hp = (float)hi[e] + (float)lo[e];
                const float nn = tanh_fold(accn[bt][4 * q + e]);
                hn[4 * q + e] = (hp - nn) * accz[bt][4 * q + e] + nn;
            }
        }
        uint4 hi0, hi1, lo0, lo1;
        pack_pair_hl(hn, hi0, hi1, lo0, lo1);
        *reinterpret_cast<uint4*>(t_wr + own_frag(0, bt, 0)) = hi0;
        *reinterpret_cast<uint4*>(t_wr + own_frag(1, bt, 0)) = hi1;
        *reinterpret_cast<uint4*>(t_wr + own_frag(0, bt, 1)) = lo0;
        *reinterpret_cast<uint4*>(t_wr + own_frag(1, bt, 1)) = lo1;
        // streaming stores: the next reader is another kernel 0.5 GB later, keep the L2 for the weight stream
        char* o = reinterpret_cast<char*>(out + (((size_t)(tile0 + bt) * kSeqLen + t) * kKB12 + (dir * kKBH + 2 * wave)) * 2 * kFragU4) + (uint32_t)t16;
        nt_store(hi0, reinterpret_cast<uint4*>(o));
        nt_store(lo0, reinterpret_cast<uint4*>(o + 1024));
        nt_store(hi1, reinterpret_cast<uint4*>(o + 2048));
        nt_store(lo1, reinterpret_cast<uint4*>(o + 3072));
    }
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#define CCSM_FENCE asm volatile("" ::: "memory")
// -DCCSM_MX_PIN_READS (experiment, not measured: the pool was closed): the layer-1/2 kernel's B-operand reads are written one MFMA group AHEAD of
// their first use, but its MFMAs are builtins - pure values the scheduler hoists over the loads, which the memory fence holds in place: in the
// code object every k-block's reads stand right in front of their first MFMA.  This pins them where they are written.
#ifdef CCSM_MX_PIN_READS
#define CCSM_PIN_READS __builtin_amdgcn_sched_barrier(0)
#else
#define CCSM_PIN_READS ((void)0)
#endif

// Weight streams, per (direction, wave), in bytes.  hi / lo fragments and blobs are 1 KiB (lane * 16), the scale dwords of a pair
// 256 B (lane * 4; byte g = gate g):
//   phase-A pair (r, z)    : hi (kbl, g) at (2 kbl + g) KiB | fp4 blob (g) at (4 + g) KiB | scales at 6 KiB                      = 6400 B
//   phase-B pair (r, z, n) : hi (kbl, g) at (3 kbl + g) KiB | fp4 blob (g) at (6 + g) KiB | scales at 9 KiB                  = 9472 B
//   phase-C pair (n)       : hi (kbl) at kbl KiB | fp6 blob: bytes 0-15 at 2 KiB, 16-23 at 3 KiB | scales at 3.5 KiB (byte 0) = 3840 B
constexpr int kMxPairA = 6 * 1024 + 256, kMxPairB = 9 * 1024 + 256, kMxPairC = 3 * 1024 + 512 + 256;
// "hybrid" arithmetic (HS3): the recurrent part in three fp16 passes (hi*hi + lo*hi + hi*lo, the state carried as fp16 hi + fp16 lo),
// the input part in split-mx.  On a TRAINED checkpoint the recurrent part is where the 4-bit corrections and the fp8 state residual cost
// accuracy (it is applied 21 times per layer): emulated on trained weights, max |dprob| 1.8e-5 with the hybrid against 1.2e-4 with split-mx
// throughout.  Its phase-B pair: hi (kbl, g) at (3 kbl + g) KiB | fp16 lo (kbl, g) at (6 + 3 kbl + g) KiB = 12 KiB.
constexpr int kMxPairBH = 12 * 1024;
// split-mx-d (DYN): the recurrent weight blobs in fp6.  Its phase-B pair: hi (kbl, g) at (3 kbl + g) KiB | fp6 blob (g): bytes 0-15 at (6 + g) KiB,
// bytes 16-23 at 9 KiB + g * 512 | scales at 10.5 KiB = 11008 B
constexpr int kMxPairBD = 10 * 1024 + 512 + 256;
constexpr int mx_pair_b(bool hs3, bool dyn = false) { return hs3 ? kMxPairBH : dyn ? kMxPairBD : kMxPairB; }
constexpr int mx0_wbytes(bool hs3, bool dyn = false) { return 4 * 1024 + (kKBH / 2) * mx_pair_b(hs3, dyn) + 2 * 1024; }   // layer 0: [r hi, r lo, z hi, z lo] [B] [n hi, n lo]
constexpr int kMx12OffB = (kKB12 / 2) * kMxPairA;
constexpr int mx12_off_c(bool hs3, bool dyn = false) { return kMx12OffB + (kKBH / 2) * mx_pair_b(hs3, dyn); }
constexpr int mx12_wbytes(bool hs3, bool dyn = false) { return mx12_off_c(hs3, dyn) + (kKB12 / 2) * kMxPairC; }

// G gates of the pair's correction product: weight blobs W[g] with scale bytes g of WS, activation blobs xc0 / xc1
#define CCSM_CORR_G(G, W, WS, SB)                                                                             \
    do {                                                                                                      \
        CCSM_FENCE;                                                                                           \
        _Pragma("unroll") for (int bt = 0; bt < NB; ++bt) {                                                   \
            acc[0][bt] = mfma_corr_mx<0>(W[0], WS, xc0[bt], xc1[bt], acc[0][bt], SB);                         \
            acc[1][bt] = mfma_corr_mx<1>(W[1], WS, xc0[bt], xc1[bt], acc[1][bt], SB);                         \
            if constexpr (G == 3) acc[2][bt] = mfma_corr_mx<2>(W[2], WS, xc0[bt], xc1[bt], acc[2][bt], SB);   \
        }                                                                                                     \
        CCSM_FENCE;                                                                                           \
    } while (0)

// ---------------------------------------------------------------------------------------------------------
// Layer 0: K = 11 input features padded to one k-block, [hi | lo] fp16 fragments on both sides, three f16 passes for the x-part
// (raw z-scores reach hundreds); the recurrent part in split-mx.  96 batch rows of one direction per workgroup, wave w owns hidden
// units [32w, 32w + 32) of all gates; phases: A x-part of r, z - B h-part of r, z, n - C x-part of n - tail.
//   xin : [tile][t][hi|lo][64] uint4          out : [tile][t][32 kb][hi | corr][64] uint4 (activation blobs in the corr fragments)
// LDS : h fragments 96 KiB | x ring 2 x 6 KiB | residuals 12 KiB | biases 4 KiB
// ---------------------------------------------------------------------------------------------------------
constexpr int mx0_xoff(int nb) { return mx_hbytes(nb); }
constexpr int mx0_looff(int nb) { return mx0_xoff(nb) + 2 * nb * 2 * 1024; }
constexpr int mx0_biasoff(int nb) { return mx0_looff(nb) + kWaves * nb * 64 * 8; }
constexpr int mx0_lds(int nb) { return mx0_biasoff(nb) + kWaves * 4 * 32 * 4; }
constexpr int kMx0Lds = mx0_lds(kMxNB);

template <bool DBG, bool HS3, bool DYN = false, int NB_ = kMxNB, bool F3 = false, bool STAG = false>
__global__ __launch_bounds__(512, 2) void gru_layer0_mx_kernel(const uint4* __restrict__ xin, uint4* __restrict__ out,
                                                                const uint4* __restrict__ wst, const float* __restrict__ bias,
                                                                const float* __restrict__ h0, int rows_p,
                                                                unsigned long long* __restrict__ dbg) {
    constexpr int NB = NB_;
    constexpr int kMx0XOff = mx0_xoff(NB), kMx0LoOff = mx0_looff(NB), kMx0BiasOff = mx0_biasoff(NB);
    static_assert(!(HS3 && DYN), "one or the other");
    static_assert(!F3 || HS3, "F3 = the hybrid's layer 0 (three passes in both parts) writing fp16 lo fragments instead of blobs");
    constexpr int PB = mx_pair_b(HS3, DYN);
    constexpr int OFF_B = 4 * 1024, OFF_C = OFF_B + (kKBH / 2) * PB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dir = blockIdx.x & 1;
    const int tile0 = (blockIdx.x >> 1) * NB;
    const int hh = lane >> 5;
    const int lane16 = lane * 16;
    const int sb = hh ? kMxScaleLo : kMxScaleHi;
    static_assert(kMxScaleLo0 - kMxScaleLo == kMxScaleHi0 - kMxScaleHi, "one offset turns the steady scales into the initial-state scales");

    if (threadIdx.x < kWaves * 4 * 32 / 4)
        reinterpret_cast<float4*>(smem + kMx0BiasOff)[threadIdx.x] = reinterpret_cast<const float4*>(bias + (size_t)dir * kWaves * 4 * 32)[threadIdx.x];
    mx_h0_to_lds<HS3, DYN, NB>(smem, kMx0LoOff, h0 + (size_t)dir * rows_p * kHidden, tile0, wave, lane);

    // x staging: 6 fragments per step (bt x hi|lo), waves 0-5 move one each
    const u32x4_t xrs = dma_rsrc(xin + (size_t)tile0 * kSeqLen * 2 * kFragU4);     // per-workgroup base: see gru_layer12_mx_kernel
    // (the 32-row form casts first: its offset otherwise trips a backend check, "Illegal instruction detected: V_CMP_NE_U32 0, $src_shared_base";
    //  the other forms keep the expression their tuned code was generated from: cast-first gives them 10 % fewer instructions and, A/B on
    //  MI355X, the same 2.47 M sites/s - the kernels are bound by energy, not by issue slots)
    const unsigned sx_base = NB == 1 ? (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)kMx0XOff
                                     : (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + kMx0XOff);
    auto stage_load = [&](int t, int buf) {
        const int f = wave < 2 * NB ? wave : 2 * NB - 1;            // the other waves re-stage the last fragment (same bytes, same place)
        const int hl = f & 1, bt = f >> 1;
        const int soff = (((bt * kSeqLen + t) * 2 + hl) << 10);
        dma16_buf(xrs, lane16, __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane((int)(sx_base + ((buf * 2 * NB + f) << 10))));
    };
    // STAG: the transfers are group 1's (waves 0-3), two per wave and step: fragments w and w + 4 (clamped: re-staged duplicates)
    auto stage_load2 = [&](int t, int buf) {
        if (wave < 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int f = wave + 4 * i < 2 * NB ? wave + 4 * i : 2 * NB - 1;
                const int hl = f & 1, bt = f >> 1;
                const int soff = (((bt * kSeqLen + t) * 2 + hl) << 10);
                dma16_buf(xrs, lane16, __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane((int)(sx_base + ((buf * 2 * NB + f) << 10))));
            }
        }
    };
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const char*>(wst) + (size_t)(dir * kWaves + wave) * mx0_wbytes(HS3, DYN));
    const int bias_off = kMx0BiasOff + wave * 4 * 32 * 4;
    auto w_at = [&](int off) -> uint4 { return buf_load(wrs, lane16, off); };
    auto ws_at = [&](int off) -> uint32_t { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, lane * 4, off, 0); };
    auto w8_at = [&](int off) -> uint2 {            // bytes 16-23 of an fp6 blob: lane * 8
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(wrs, lane * 8, off, 0);
        return make_uint2(v[0], v[1]);
    };

    uint4 wxa[2][2];                                                // phase A: [gate r,z][hi, lo]
    uint4 wxc[2];                                                   // phase C: n gate [hi, lo]
    uint4 wbh[2][3], wbb[3];                                        // phase B resident pair: [kb in pair][gate] hi ; [gate] blob
    uint32_t wbs = 0;                                               //                        scale bytes
    uint2 wbb1[3];                                                  // split-mx-d: bytes 16-23 of the fp6 blobs
    uint4 wbl[2][3];                                                // hybrid arithmetic: [kb in pair][gate] fp16 lo instead of the blobs
    auto ld_first = [&]() {                                         // everything a step needs before its second phase-B pair: 14 (16) requests
#pragma unroll
        for (int g = 0; g < 2; ++g) { wxa[g][0] = w_at((2 * g) << 10); wxa[g][1] = w_at((2 * g + 1) << 10); }
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            wbh[0][g] = w_at(OFF_B + (g << 10)); wbh[1][g] = w_at(OFF_B + ((3 + g) << 10));
            if constexpr (HS3) { wbl[0][g] = w_at(OFF_B + ((6 + g) << 10)); wbl[1][g] = w_at(OFF_B + ((9 + g) << 10)); }
            else wbb[g] = w_at(OFF_B + ((6 + g) << 10));
            if constexpr (DYN) wbb1[g] = w8_at(OFF_B + (9 << 10) + g * 512);
        }
        if constexpr (!HS3) wbs = ws_at(OFF_B + (DYN ? (10 << 10) + 512 : (9 << 10)));
    };
    if constexpr (STAG) stage_load2(dir ? kSeqLen - 1 : 0, 0);
    else stage_load(dir ? kSeqLen - 1 : 0, 0);
    ld_first();
    if constexpr (HS3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (DYN) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");          // the first transfer (older than the 14 weight requests)
    if constexpr (STAG) {
        __syncthreads();                                            // x_0, the initial states and the biases in LDS
        if (wave >= 4) __syncthreads();                             // group 2 runs one slot behind group 1 from here on
    }

    for (int s = 0; s < kSeqLen; ++s) {
        const int t = dir ? (kSeqLen - 1 - s) : s;
        auto stamp = [&](int k) {
            if constexpr (DBG) {
                if (dbg != nullptr && blockIdx.x == 0 && lane == 0) dbg[(s * kWaves + wave) * 5 + k] = __builtin_readcyclecounter();
            }
        };
        stamp(0);
        const int tn = s + 1 < kSeqLen ? (dir ? t - 1 : t + 1) : t;
        f32x16 acc[3][NB];                                          // R, Z, N
        auto lane16_here = [&]() -> int {                           // opaque copy: per-lane addresses are rebuilt where a phase needs them
            int v = lane16;
            asm volatile("" : "+v"(v));
            return v;
        };
        auto bias_set = [&](int set) {
            f32x16 b;
            const char* bp = smem + (bias_off + ((lane16_here() >> 9) << 6));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(bp + set * 128 + q * 16);
                b[4 * q + 0] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
            }
            return b;
        };
        {
            const f32x16 b0 = bias_set(0), b1 = bias_set(1);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) { acc[0][bt] = b0; acc[1][bt] = b1; }
        }
        // ---------------- phase A: R, Z += W_i{r,z} x_t (three fp16 passes) -------------------------------------------------
        // the transfer of this step's x (issued one step ago) is older than the 14 weight requests and 4 NB output stores of the tail
        // (4 NB stores; the hybrid has 16 weight requests there, split-mx-d 17)
        if constexpr (!STAG) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((HS3 ? 16 : DYN ? 17 : 14) + 4 * NB) : "memory");
            __syncthreads();                                        // x_t in LDS; everybody's h_{t-1} fragments written
            stage_load(tn, (s + 1) & 1);
        }
        auto rd_x0 = [&](uint4 (&x0)[NB][2]) {
            const int l16 = lane16_here();
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                x0[bt][0] = *reinterpret_cast<const uint4*>(smem + kMx0XOff + ((((s & 1) * NB + bt) * 2 + 0) << 10) + l16);
                x0[bt][1] = *reinterpret_cast<const uint4*>(smem + kMx0XOff + ((((s & 1) * NB + bt) * 2 + 1) << 10) + l16);
            }
        };
        {
            uint4 x0[NB][2];
            rd_x0(x0);
            CCSM_FENCE;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    acc[g][bt] = mfma16(wxa[g][0], x0[bt][0], acc[g][bt]);
                    acc[g][bt] = mfma16(wxa[g][0], x0[bt][1], acc[g][bt]);
                    acc[g][bt] = mfma16(wxa[g][1], x0[bt][0], acc[g][bt]);
                }
            CCSM_FENCE;
        }
        if constexpr (!STAG) { wxc[0] = w_at(OFF_C); wxc[1] = w_at(OFF_C + 1024); }
        if constexpr (!STAG) stamp(1);
        // ---------------- phase B: R, Z, N += W_h{r,z,n} h_{t-1}  (N starts at b_hn) ---------------------------------
        {
            const f32x16 b3 = bias_set(3);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) acc[2][bt] = b3;
        }
        const int sbh = sb + (s == 0 ? kMxScaleHi0 - kMxScaleHi : 0);
        uint4 xh[NB], xc0[NB];
        uint2 xc1[NB];
        if constexpr (HS3) {
            // hybrid arithmetic: three fp16 passes per k-block (W_hi h_hi + W_lo h_hi + W_hi h_lo); one pair resident (hi and lo fragments
            // of its two k-blocks), each k-block's six fragments refilled with the next pair's right behind its MFMAs
            uint4 xl[NB];
            static_for<0, kKBH>([&](auto KC) {
                constexpr int KB = decltype(KC)::value;
                constexpr int KBL = KB & 1, Q = KB >> 1;
                constexpr int NXT = OFF_B + (Q + 1) * PB;
                if constexpr (STAG && KB == kKBH - 2) stage_load2(tn, (s + 1) & 1);     // behind the phase's last weight request
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) {
                    xh[bt] = *reinterpret_cast<const uint4*>(smem + mx_hfrag<NB>(KB, bt, 0) + lane * 16);
                    xl[bt] = *reinterpret_cast<const uint4*>(smem + mx_hfrag<NB>(KB, bt, 1) + lane * 16);
                }
                CCSM_FENCE;
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        acc[g][bt] = mfma16(wbh[KBL][g], xh[bt], acc[g][bt]);
                        acc[g][bt] = mfma16(wbl[KBL][g], xh[bt], acc[g][bt]);
                        acc[g][bt] = mfma16(wbh[KBL][g], xl[bt], acc[g][bt]);
                    }
                CCSM_FENCE;
                if constexpr (Q + 1 < kKBH / 2) {
#pragma unroll
                    for (int g = 0; g < 3; ++g) { wbh[KBL][g] = w_at(NXT + ((3 * KBL + g) << 10)); wbl[KBL][g] = w_at(NXT + ((6 + 3 * KBL + g) << 10)); }
                }
                CCSM_FENCE;
                // alpha: the other group's units of h_{t-1} are written.  (The scheduling barrier keeps this k-block's MFMAs in front of
                // the workgroup barrier: moved behind it, the refilled fragments needed registers of their own - spills behind vmcnt(0).)
                if constexpr (STAG && KB == kKBH / 2 - 1) { __builtin_amdgcn_sched_barrier(0); stamp(1); __syncthreads(); CCSM_FENCE; }
            });
        } else
        static_for<0, kKBH / 2>([&](auto QC) {
            constexpr int Q = decltype(QC)::value;
            constexpr bool LAST = Q == kKBH / 2 - 1;
            constexpr int NXT = OFF_B + (Q + 1) * PB;
            uint32_t pm[NB];                                            // split-mx-d: running max |x_hi| of the pair's block, this lane's values
            if constexpr (STAG && LAST) stage_load2(tn, (s + 1) & 1);   // behind the phase's last weight request
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                xh[bt] = *reinterpret_cast<const uint4*>(smem + mx_hfrag<NB>(2 * Q, bt, 0) + lane * 16);
                xc0[bt] = *reinterpret_cast<const uint4*>(smem + mx_hfrag<NB>(2 * Q, bt, 1) + lane * 16);
                xc1[bt] = *reinterpret_cast<const uint2*>(smem + mx_hfrag<NB>(2 * Q + 1, bt, 1) + lane * 16);
            }
            CCSM_FENCE;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g][bt] = mfma16(wbh[0][g], xh[bt], acc[g][bt]);
            CCSM_FENCE;
            if constexpr (DYN) {
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) pm[bt] = absmax8(xh[bt], 0u);
            }
            if constexpr (!LAST) {
#pragma unroll
                for (int g = 0; g < 3; ++g) wbh[0][g] = w_at(NXT + (g << 10));
            }
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) xh[bt] = *reinterpret_cast<const uint4*>(smem + mx_hfrag<NB>(2 * Q + 1, bt, 0) + lane * 16);
            CCSM_FENCE;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g][bt] = mfma16(wbh[1][g], xh[bt], acc[g][bt]);
            CCSM_FENCE;
            if constexpr (!LAST) {
#pragma unroll
                for (int g = 0; g < 3; ++g) wbh[1][g] = w_at(NXT + ((3 + g) << 10));
            }
            if constexpr (DYN) {
                // the block scales of the three row tiles (vector ALU in the shadow of the main MFMAs above), then the correction products
                int sbd[NB];
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) sbd[bt] = 127 + dyn_block_exp(absmax8(xh[bt], pm[bt])) - (hh ? 11 : 0);
                CCSM_FENCE;
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) {
                    acc[0][bt] = mfma_corr_mx6<0>(wbb[0], wbb1[0], wbs, xc0[bt], xc1[bt], acc[0][bt], sbd[bt]);
                    acc[1][bt] = mfma_corr_mx6<1>(wbb[1], wbb1[1], wbs, xc0[bt], xc1[bt], acc[1][bt], sbd[bt]);
                    acc[2][bt] = mfma_corr_mx6<2>(wbb[2], wbb1[2], wbs, xc0[bt], xc1[bt], acc[2][bt], sbd[bt]);
                }
                CCSM_FENCE;
            } else {
                CCSM_CORR_G(3, wbb, wbs, sbh);
            }
            if constexpr (!LAST) {
#pragma unroll
                for (int g = 0; g < 3; ++g) wbb[g] = w_at(NXT + ((6 + g) << 10));
                if constexpr (DYN) {
#pragma unroll
                    for (int g = 0; g < 3; ++g) wbb1[g] = w8_at(NXT + (9 << 10) + g * 512);
                }
                wbs = ws_at(NXT + (DYN ? (10 << 10) + 512 : (9 << 10)));
            }
            CCSM_FENCE;
            if constexpr (STAG && Q == kKBH / 4 - 1) { __builtin_amdgcn_sched_barrier(0); stamp(1); __syncthreads(); CCSM_FENCE; }   // alpha: the other group's units of h_{t-1} are written
        });
        if constexpr (STAG) {
            __builtin_amdgcn_sched_barrier(0);
            stamp(2);
            __syncthreads();                                        // beta: both groups have read units 0-127 of h_{t-1}; group 1 may overwrite them
            CCSM_FENCE;
            wxc[0] = w_at(OFF_C); wxc[1] = w_at(OFF_C + 1024);      // (not held through phase B: in flight under the sigmoids)
            stamp(3);
        }
        // r = sigmoid(R) ; N = b_in + r * N
        {
            const f32x16 b2 = bias_set(2);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[2][bt][r] = b2[r] + sigmoid_f(acc[0][bt][r]) * acc[2][bt][r];
        }
        if constexpr (!STAG) stamp(2);
        // ---------------- phase C: N += W_in x_t (x_t is still in its ring buffer) -------------------------------------------
        {
            uint4 x0[NB][2];
            rd_x0(x0);
            CCSM_FENCE;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                acc[2][bt] = mfma16(wxc[0], x0[bt][0], acc[2][bt]);
                acc[2][bt] = mfma16(wxc[0], x0[bt][1], acc[2][bt]);
                acc[2][bt] = mfma16(wxc[1], x0[bt][0], acc[2][bt]);
            }
            CCSM_FENCE;
        }
        if constexpr (!STAG) stamp(3);
        ld_first();                                                 // the next step's first weight fragments: in flight during the tail
        CCSM_FENCE;
#pragma unroll
        for (int bt = 0; bt < NB; ++bt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[1][bt][r] = sigmoid_f(acc[1][bt][r]);
        if constexpr (!STAG) __syncthreads();                       // every wave has read h_{t-1} (phase B) before anybody overwrites its fragments
        else __builtin_amdgcn_sched_barrier(0);                     // (keeps the tail's operand reads behind the sigmoids, as the barrier did)
        if constexpr (F3) f3_tail<NB>(smem, acc[1], acc[2], out, tile0, t, dir, wave, lane16_here());
        else mx_tail<false, HS3, DYN, NB>(smem, kMx0LoOff, acc[1], acc[2], out, tile0, t, dir, wave, lane16_here());
        if constexpr (STAG) {
            stamp(4);
            // gamma: this group's units of h_t are written; group 1's transfer of x_{t+1} (older than the 2 + 14 weight requests and the
            // 4 NB output stores issued since) has landed
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 + (HS3 ? 16 : DYN ? 17 : 14) + 4 * NB) : "memory");
            __syncthreads();
        }
        if constexpr (!STAG) stamp(4);
    }
    if constexpr (STAG) {
        if (wave < 4) __syncthreads();                              // group 1 keeps group 2's last barrier company
    }
    if constexpr (DBG) {            // where the wave ran: HW_ID (wave slot, SIMD, CU ...) behind the stamps
        if (dbg != nullptr && blockIdx.x == 0 && lane == 0) dbg[kSeqLen * kWaves * 5 + wave] = __builtin_amdgcn_s_getreg((4 /* HW_REG_HW_ID */) | (0 << 6) | (31 << 11));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // no transfer may still be writing LDS when the workgroup retires
}

// ---------------------------------------------------------------------------------------------------------
// Layers 1-2 (KX = 32 k-blocks of input = 16 pairs), on a schedule built around two facts measured on the chunked round-1 kernel
// (DESIGN.md 7): (i) x_t comes from HBM and a transfer takes ~3.5-4 k cycles to land, (ii) a wave's vector-memory operations
// retire IN ORDER, so a weight fragment requested after a transfer cannot be used before that transfer has landed.  Hence
//   * the x ring holds FOUR pairs of k-blocks (4 x 12 KiB); the slot of the pair consumed by iteration g is refilled with the pair of
//     iteration g + 4 right behind iteration g's barrier (measured +1.4 % against refilling it as the iteration's youngest operation);
//   * weights are requested three pairs ahead in phase A (three register slots), one pair ahead in phase B, four pairs ahead
//     in phase C (four slots of the n gate);
//   * ONE barrier per pair, in the middle of the pair (after the main MFMAs, before the correction MFMAs whose operands are
//     already in registers): it publishes pair g + 1 (each wave first waits for its own part of that transfer with a COUNTED
//     s_waitcnt) and retires pair g (all its operand reads are done), so the first operand of pair g + 1 is read from LDS
//     while pair g's correction MFMAs run.
// Iterations ("consumptions") per step: 16 pairs of phase A, then 16 of phase C; phase B touches no x.  The pair code is
// straight-line (static_for over compile-time pair indices): a rolled pair loop with peeled ends made the compiler shuffle weight
// slots between register sets at the loop exits, each shuffle behind an s_waitcnt vmcnt(0).  s_waitcnt immediates count the
// vector-memory operations a wave issues between a transfer and the barrier that needs it: per phase-A pair 7 weight requests
// (2 + 2 hi fragments, 2 fp4 blobs + 1 scale dword) + d transfer instructions, per phase-C pair 5 + d, d = 2 for waves 0-3
// (fragments w and w + 8 of the pair) and 1 for waves 4-7; every load below is therefore UNCONDITIONAL.
//   xin : [tile][t][32 kb][hi | corr][64] uint4      out : the same (OUT_FP8: fp8 corr fragments for the attention kernel)
// LDS : h fragments 96 KiB | x ring 4 x 12 KiB | residuals 12 KiB | biases 4 KiB = 160 KiB
// ---------------------------------------------------------------------------------------------------------
// diagnostic build -DCCSM_PWR_LDS1 (results wrong on purpose): the layer-1/2 kernel reads the B operands of row tile 0 for all three row
// tiles - one LDS read where there are three: what the activations' LDS reads cost
//   = 1: every B operand; = 3: the hi fragments only (main MFMAs); = 2: all reads as shipped, but the main MFMAs take row tile 0's operand
//   (the reads of tiles 1, 2 pinned behind them): 3 against 2 isolates the LDS reads from the operand toggling of the MFMAs
#ifndef CCSM_PWR_LDS1
#define CCSM_PWR_LDS1 0
#endif
constexpr int lds_bt(int bt) { return CCSM_PWR_LDS1 == 1 || CCSM_PWR_LDS1 == 3 ? 0 : bt; }        // hi fragments
constexpr int lds_btc(int bt) { return CCSM_PWR_LDS1 == 1 ? 0 : bt; }                              // blob fragments
constexpr int kMxRS = 4;
// Round 6 (profiles/r06_h: stamps of the 16-wide form of this kernel): the CU's vector-memory path takes one 1-KiB request per 16 cycles from all
// eight waves together, and a wave that waits for its slot issues no MFMA - so requests are issued ONE at a time, each behind the last use of the
// fragment it replaces (gate by gate: three MFMAs), instead of two or three in a row behind six.  Same requests in the same order among
// themselves (the counted waits stand), same products into every accumulator in the same order (same bits).  -DCCSM_MX_NO_ILV: the round-5 order.
#ifdef CCSM_MX_NO_ILV
constexpr bool kMxIlv = false;
#else
constexpr bool kMxIlv = true;
#endif
constexpr int mx_slot_bytes(int nb) { return 2 * nb * 2 * 1024; }
constexpr int mx12_xoff(int nb) { return mx_hbytes(nb); }
constexpr int mx12_looff(int nb) { return mx12_xoff(nb) + kMxRS * mx_slot_bytes(nb); }
constexpr int mx12_biasoff(int nb) { return mx12_looff(nb) + kWaves * nb * 64 * 8; }
constexpr int mx12_lds(int nb) { return mx12_biasoff(nb) + kWaves * 4 * 32 * 4; }
constexpr int kMx12Lds = mx12_lds(kMxNB);

template <bool OUT_FP8, bool DBG, bool HS3, bool DYN = false, int NB_ = kMxNB>
__global__ __launch_bounds__(512, 2) void gru_layer12_mx_kernel(const uint4* __restrict__ xin, uint4* __restrict__ out,
                                                                 const uint4* __restrict__ wst, const float* __restrict__ bias,
                                                                 const float* __restrict__ h0, int rows_p,
                                                                 unsigned long long* __restrict__ dbg) {
    constexpr int NB = NB_, KX = kKB12, NPAIR = KX / 2, RS = kMxRS, SLOT_BYTES = mx_slot_bytes(NB);
    constexpr int kMx12XOff = mx12_xoff(NB), kMx12LoOff = mx12_looff(NB), kMx12BiasOff = mx12_biasoff(NB);
    static_assert(!(HS3 && DYN), "one or the other");
    constexpr int PA = kMxPairA, PB = mx_pair_b(HS3, DYN), PC = kMxPairC, OFF_B = kMx12OffB, OFF_C = mx12_off_c(HS3, DYN);
    constexpr int OFF_BS = DYN ? (10 << 10) + 512 : (9 << 10);      // scale dwords of a phase-B pair
    constexpr int X_OFF = kMx12XOff;
    constexpr bool XD = HS3 || DYN;                                 // the layer below wrote block-scaled blobs (mx_tail)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dir = blockIdx.x & 1;
    const int tile0 = (blockIdx.x >> 1) * NB;
    const int hh = lane >> 5;
    const int lane16 = lane * 16;
    const int sb = hh ? kMxScaleLo : kMxScaleHi;
    start_stagger((blockIdx.x >> 1) & 3);

    if (threadIdx.x < kWaves * 4 * 32 / 4)
        reinterpret_cast<float4*>(smem + kMx12BiasOff)[threadIdx.x] = reinterpret_cast<const float4*>(bias + (size_t)dir * kWaves * 4 * 32)[threadIdx.x];
    mx_h0_to_lds<HS3, DYN, NB>(smem, kMx12LoOff, h0 + (size_t)dir * rows_p * kHidden, tile0, wave, lane);

    // ---- x transfers: fragment f = (kbl * NB + bt) * 2 + hl of a ring slot; wave w moves fragment w, waves 0-3 also w + 8
    // the descriptor starts at THIS workgroup's first tile (64-bit address arithmetic): its 2 GiB range and the 32-bit offsets below
    // then never see more than three tiles, whatever the size of the launch (a layer output passes 2 GiB at 24960 sites)
    const u32x4_t xrs = dma_rsrc(xin + (size_t)tile0 * kSeqLen * KX * 2 * kFragU4);
    const unsigned sx_base = NB == 1 ? (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)X_OFF      // (see gru_layer0_mx_kernel)
                                     : (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + X_OFF);
    auto dma_pair = [&](int slot, int sd, int jd, bool last_read = false) {   // wave-uniform: ring slot, step (clamped), pair of x_t(sd)
        const int sc_ = sd < kSeqLen ? sd : kSeqLen - 1;
        const int td = dir ? kSeqLen - 1 - sc_ : sc_;
        auto one = [&](int f) {
            const int hl = f & 1, bt = (f >> 1) % NB, kbl = (f >> 1) / NB;
            const int soff = ((((bt * kSeqLen + td) * KX + (2 * jd + kbl)) * 2 + hl) << 10);
#ifdef CCSM_DMA_C_NT
            if (last_read) { dma16_buf_nt(xrs, lane16, __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane((int)(sx_base + slot * SLOT_BYTES + (f << 10)))); return; }
#endif
#ifdef CCSM_PWR_NOXC        // diagnostic build (results wrong on purpose): phase C's transfers (the SECOND read of x_t) run with an empty exec mask - the
                           // instruction issues and counts, nothing is fetched: the upper bound of what one pass over x_t could save in this schedule
            if (last_read) {
                asm volatile("s_mov_b32 m0, %3\n\ts_mov_b64 exec, 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_mov_b64 exec, -1"
                             : : "v"(lane16), "s"(xrs), "s"(__builtin_amdgcn_readfirstlane(soff)),
                                 "s"(__builtin_amdgcn_readfirstlane((int)(sx_base + slot * SLOT_BYTES + (f << 10)))) : "memory");
                return;
            }
#endif
#ifdef CCSM_PWR_HALFCORR    // diagnostic build (results wrong on purpose): the blob fragments' transfers move their upper lanes only - what the layer
                           // input costs without the derivable fp6 copy of x_hi, before the cost of re-deriving it
            if (hl) {
                asm volatile("s_mov_b32 m0, %3\n\ts_mov_b32 exec_lo, 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_mov_b64 exec, -1"
                             : : "v"(lane16), "s"(xrs), "s"(__builtin_amdgcn_readfirstlane(soff)),
                                 "s"(__builtin_amdgcn_readfirstlane((int)(sx_base + slot * SLOT_BYTES + (f << 10)))) : "memory");
                return;
            }
#endif
            dma16_buf(xrs, lane16, __builtin_amdgcn_readfirstlane(soff),
                      __builtin_amdgcn_readfirstlane((int)(sx_base + slot * SLOT_BYTES + (f << 10))));
        };
        if (4 * NB >= kWaves || wave < 4 * NB) one(wave);           // 4 NB fragments per pair (NB = 1: waves 4-7 move nothing)
        if (4 * NB > kWaves && wave < 4 * NB - kWaves) one(wave + 8);
    };
    // the transfer of consumption jj + RS of step s (jj: 0-15 phase A pairs, 16-31 phase C pairs) goes into the slot consumption jj
    // just vacated
    auto dma_ahead = [&](int slot, int s, int jj) {
        if constexpr (kMxDiag & 16) return;
        const int g = jj + RS;
        const int c = g & (2 * NPAIR - 1);                           // consumption within its step: 0-15 phase A, 16-31 phase C
        dma_pair(slot, s + (g >> 5), kMxZigZag && c >= NPAIR ? 2 * NPAIR - 1 - c : c & (NPAIR - 1), c >= NPAIR);
    };
    // wait until this wave's part of a transfer has landed: at most NLO (waves 4-7) / NHI (waves 0-3) younger operations
#define CCSM_WAIT_XFER(NLO, NHI)                                                        \
    do {                                                                                \
        if constexpr (kMxDiag & 32) break;                                              \
        if (4 * NB > kWaves && wave < 4 * NB - kWaves) asm volatile("s_waitcnt vmcnt(" #NHI ")" ::: "memory"); \
        else if (4 * NB >= kWaves || wave < 4 * NB) asm volatile("s_waitcnt vmcnt(" #NLO ")" ::: "memory");   \
    } while (0)

    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const char*>(wst) + (size_t)(dir * kWaves + wave) * mx12_wbytes(HS3, DYN));
    const int bias_off = kMx12BiasOff + wave * 4 * 32 * 4;
#ifdef CCSM_PWR_W1          // diagnostic build (results wrong on purpose): every weight request of the layer-1/2 kernel reads one of the stream's first four
                           // fragments (L1-resident): same requests, no L2 -> CU weight traffic - what the weight stream costs beyond its instructions
    auto w_at = [&](int off) -> uint4 { return buf_load(wrs, lane16, off & 0xc00); };
#else
    auto w_at = [&](int off) -> uint4 { return buf_load(wrs, lane16, off); };
#endif
    auto ws_at = [&](int off) -> uint32_t { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, lane * 4, off, 0); };
    auto w8_at = [&](int off) -> uint2 {            // bytes 16-23 of an fp6 blob: lane * 8
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        int v8 = lane16;            // only phases B / C need lane * 8 and no register is free to keep it through phase A (it was spilled: 8-16 B
        asm volatile("" : "+v"(v8));// of scratch, profiles/r03_z_isa.md): rebuilt from an opaque copy of lane * 16 at every use instead
        v8 >>= 1;
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(wrs, v8, off, 0);
        return make_uint2(v[0], v[1]);
    };

    // weight registers: phase A three pair slots: [slot][kb in pair][gate r,z] hi, [slot][gate] fp6 blobs (16 + 8 bytes), [slot] scale
    // bytes; phase B one resident pair (fp4 blobs); phase C four pair slots of the n gate: [slot][kb in pair] hi, [slot] fp6 blob, scale
    // (-DCCSM_MX_A_SLOTS=4 gives plain split-mx a fourth phase-A slot: measured in round 4, phase A -3 %, the step +3 % (the slot's
    // load behind the tail and 8 B of scratch cost more than the look-ahead gains - phase A is bound by the CU's vector-memory path,
    // not by latency: profiles/r04_j_a_slots.log); the default stays three)
#ifndef CCSM_MX_A_SLOTS
#define CCSM_MX_A_SLOTS 3
#endif
    constexpr int NSA = (HS3 || DYN) ? 3 : CCSM_MX_A_SLOTS;
    uint4 wah[NSA][2][2], wab[NSA][2];
    uint32_t was[NSA];
    uint4 wbh[2][3], wbb[3];
    uint32_t wbs = 0;
    uint2 wbb1[3];                                  // split-mx-d: bytes 16-23 of the resident phase-B pair's fp6 blobs
    uint4 wbl[2][3];                                // hybrid arithmetic: fp16 lo fragments of the resident phase-B pair instead of the blobs
    uint4 wch[4][2], wcb[4];
    uint2 wcb1[4];
    uint32_t wcs[4];
    auto ldAh = [&](uint4 (&d)[2], int p, int kbl) {
#pragma unroll
        for (int g = 0; g < 2; ++g) d[g] = w_at(p * PA + ((2 * kbl + g) << 10));
    };
    auto ldAb = [&](int ws, int p) {                // 3 requests
#pragma unroll
        for (int g = 0; g < 2; ++g) wab[ws][g] = w_at(p * PA + ((4 + g) << 10));
        was[ws] = ws_at(p * PA + (6 << 10));
    };
    auto ldA_slot = [&](int ws, int p) { ldAh(wah[ws][0], p, 0); ldAh(wah[ws][1], p, 1); ldAb(ws, p); };   // 7 requests

    // ---- prologue: the ring's first pairs, the first three weight slots
#pragma unroll
    for (int g = 0; g < RS; ++g) dma_pair(g, 0, g);
    ldA_slot(0, 0);
    ldA_slot(1, 1);
    ldA_slot(2, 2);
    if constexpr (NSA == 4) ldA_slot(3, 3);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(7 * NSA) : "memory");   // all ring transfers (older than the 7 NSA weight requests)
    __syncthreads();                                                // ring, h0 fragments and biases are in LDS

    int slot = 0;                                                   // ring slot of the next consumption (wave-uniform)
    for (int s = 0; s < kSeqLen; ++s) {
        const int t = dir ? (kSeqLen - 1 - s) : s;
        auto stamp = [&](int k) {
            if constexpr (DBG) {
                if (dbg != nullptr && blockIdx.x == 0 && lane == 0) dbg[(s * kWaves + wave) * 5 + k] = __builtin_readcyclecounter();
            }
        };
        stamp(0);
        f32x16 acc[3][NB];                                          // R, Z, N
        auto lane16_here = [&]() -> int {                           // opaque copy: per-lane addresses are rebuilt where a phase needs them
            int v = lane16;
            asm volatile("" : "+v"(v));
            return v;
        };
        auto bias_set = [&](int set) {
            f32x16 b;
            const char* bp = smem + (bias_off + ((lane16_here() >> 9) << 6));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(bp + set * 128 + q * 16);
                b[4 * q + 0] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
            }
            return b;
        };
        {
            const f32x16 b0 = bias_set(0), b1 = bias_set(1);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) { acc[0][bt] = b0; acc[1][bt] = b1; }
        }

        uint4 xh[NB], xh1[NB], xc0[NB];
        uint2 xc1[NB];
        int xsc = 0;                                                // XD: the x blobs' E8M0 scales, byte bt = row tile bt (written beside the blob by the layer below)
        auto rdx = [&](uint4 (&x)[NB], int xs, int kbl, int f) {    // xs = byte offset of the slot + lane * 16
            if constexpr (kMxDiag & 4) return;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) x[bt] = *reinterpret_cast<const uint4*>(smem + xs + (((kbl * NB + lds_bt(bt)) * 2 + f) << 10));
        };
        auto rdx_blob = [&](int xs) {
            if constexpr (kMxDiag & 4) return;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                xc0[bt] = *reinterpret_cast<const uint4*>(smem + xs + (((0 * NB + lds_btc(bt)) * 2 + 1) << 10));
                xc1[bt] = *reinterpret_cast<const uint2*>(smem + xs + (((1 * NB + lds_btc(bt)) * 2 + 1) << 10));
            }
            if constexpr (XD) xsc = *reinterpret_cast<const int*>(smem + xs + (((1 * NB + (NB - 1)) * 2 + 1) << 10) + 8);   // bytes 0..2: row tiles 0..2
        };
        auto slot_off = [&](int sl) -> int { return X_OFF + sl * SLOT_BYTES + lane16; };
#define CCSM_MAIN(W, X, G, S0)                                                                                \
    do {                                                                                                      \
        CCSM_FENCE;                                                                                           \
        _Pragma("unroll") for (int bt = 0; bt < NB; ++bt) _Pragma("unroll") for (int g = 0; g < G; ++g)       \
            acc[S0 + g][bt] = mfma16(W[g], X[CCSM_PWR_LDS1 == 2 ? 0 : bt], acc[S0 + g][bt]);                  \
        if constexpr (CCSM_PWR_LDS1 == 2) {                                                                   \
            _Pragma("unroll") for (int bt = 1; bt < NB; ++bt)                                                 \
                asm volatile("" :: "v"(X[bt].x), "v"(X[bt].y), "v"(X[bt].z), "v"(X[bt].w));                   \
        }                                                                                                     \
        CCSM_FENCE;                                                                                           \
    } while (0)

        // ---------------- phase A: R, Z += W_i{r,z} x_t, pairs 0..15 ---------------------------------------------------------
        // pair P lives in weight slot P % 3; behind its three MFMA groups its slot is refilled with pair P + 3 (2 + 2 + 3
        // requests); pairs 13 and 14 take phase B's first pair instead (9 + 1 requests), pair 15 has nothing left to request and
        // its ring refill is deferred to the end of phase B: issued here it would sit in front of phase B's one-pair-ahead
        // weight requests.
        rdx(xh, slot_off(slot), 0, 0);
        int slot_a15 = 0;
        static_for<0, NPAIR>([&](auto PC_) {
            constexpr int P = decltype(PC_)::value;
            constexpr int WS = P % NSA;
            const int xs = slot_off(slot);
            const int slot_n = slot == RS - 1 ? 0 : slot + 1;
            rdx(xh1, xs, 1, 0);
            CCSM_PIN_READS;
            if constexpr (kMxIlv && CCSM_PWR_LDS1 == 0 && P + NSA < NPAIR) {
                static_for<0, 2>([&](auto GC) {
                    constexpr int g = decltype(GC)::value;
                    CCSM_FENCE;
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt) acc[g][bt] = mfma16(wah[WS][0][g], xh[bt], acc[g][bt]);
                    CCSM_FENCE;
                    wah[WS][0][g] = w_at((P + NSA) * PA + ((2 * 0 + g) << 10));
                });
            } else {
            CCSM_MAIN(wah[WS][0], xh, 2, 0);
            if constexpr (P + NSA < NPAIR) ldAh(wah[WS][0], P + NSA, 0);
            }
            if constexpr (P + NSA < NPAIR) {}
            else if constexpr (P == 13) { wbh[0][0] = w_at(OFF_B + (0 << 10)); wbh[0][1] = w_at(OFF_B + (1 << 10)); }
            else if constexpr (P == 14) {
                if constexpr (HS3) wbl[1][0] = w_at(OFF_B + (9 << 10)); else wbs = ws_at(OFF_B + OFF_BS);
            }
            rdx_blob(xs);
            CCSM_PIN_READS;
            if constexpr (kMxIlv && CCSM_PWR_LDS1 == 0 && P + NSA < NPAIR) {
                static_for<0, 2>([&](auto GC) {
                    constexpr int g = decltype(GC)::value;
                    CCSM_FENCE;
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt) acc[g][bt] = mfma16(wah[WS][1][g], xh1[bt], acc[g][bt]);
                    CCSM_FENCE;
                    wah[WS][1][g] = w_at((P + NSA) * PA + ((2 * 1 + g) << 10));
                });
            } else {
            CCSM_MAIN(wah[WS][1], xh1, 2, 0);
            if constexpr (P + NSA < NPAIR) ldAh(wah[WS][1], P + NSA, 1);
            }
            if constexpr (P + NSA < NPAIR) {}
            else if constexpr (P == 13) { wbh[0][2] = w_at(OFF_B + (2 << 10)); wbh[1][0] = w_at(OFF_B + (3 << 10)); }
            else if constexpr (P == 14 && HS3) { wbl[1][1] = w_at(OFF_B + (10 << 10)); wbl[1][2] = w_at(OFF_B + (11 << 10)); }
            else if constexpr (P == 14 && DYN) { wbb1[0] = w8_at(OFF_B + (9 << 10)); wbb1[1] = w8_at(OFF_B + (9 << 10) + 512); wbb1[2] = w8_at(OFF_B + (9 << 10) + 1024); }
            // this wave's part of the next pair's transfer has landed.  Operations the wave has issued since that refill (it sits right
            // behind its pair's barrier): the 3 blob / scale requests of that pair, two pairs of 7 + d, 4 of this pair; pair 13 requests
            // phase B's first pair instead (9), pair 14 one more of it, pair 15 nothing
            // (hybrid arithmetic: pair 14 requests three fragments of phase B's first pair instead of one; split-mx-d: four - the scale
            // dword and the three 8-byte ends of the fp6 blobs)
            // (four slots: pair 12 has nothing left to request, so the windows of pairs 12..15 hold 7 requests fewer each:
            //  pair 12: 3 + 2 (7 + d) + 0; 13: 3 + (7 + d) + d + 4; 14: 3 + d + (9 + d) + 1; 15: (9 + d) + (1 + d))
            if constexpr (NSA == 4 && P == NPAIR - 4) CCSM_WAIT_XFER(19, 21);
            else if constexpr (NSA == 4 && P == NPAIR - 3) CCSM_WAIT_XFER(16, 18);
            else if constexpr (NSA == 4 && P == NPAIR - 2) CCSM_WAIT_XFER(15, 17);
            else if constexpr (NSA == 4 && P == NPAIR - 1) CCSM_WAIT_XFER(12, 14);
            else if constexpr (P == NPAIR - 1) { if constexpr (HS3) CCSM_WAIT_XFER(17, 19); else if constexpr (DYN) CCSM_WAIT_XFER(18, 20); else CCSM_WAIT_XFER(15, 17); }
            else if constexpr (P == NPAIR - 2) { if constexpr (HS3) CCSM_WAIT_XFER(24, 26); else if constexpr (DYN) CCSM_WAIT_XFER(25, 27); else CCSM_WAIT_XFER(22, 24); }
            else CCSM_WAIT_XFER(23, 25);
            if constexpr (!(kMxDiag & 2)) __syncthreads();             // the next pair is in LDS; every wave has read this pair's operands
            if constexpr (P + 1 < NPAIR) dma_ahead(slot, s, P); else slot_a15 = slot;   // the vacated slot is refilled at once (0.4 pair more
                                                                                      // lead than as the pair's youngest operation: +1.4 %)
            if constexpr (P + 1 < NPAIR) rdx(xh, slot_off(slot_n), 0, 0);
            CCSM_PIN_READS;
            CCSM_FENCE;
            if constexpr (XD) {
                static_for<0, NB>([&](auto BC) {
                    constexpr int bt = decltype(BC)::value;
                    acc[0][bt] = mfma_corr_mx<0, bt>(wab[WS][0], was[WS], xc0[bt], xc1[bt], acc[0][bt], xsc);
                    acc[1][bt] = mfma_corr_mx<1, bt>(wab[WS][1], was[WS], xc0[bt], xc1[bt], acc[1][bt], xsc);
                });
            } else if constexpr (kMxIlv && P + NSA < NPAIR) {
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) acc[0][bt] = mfma_corr_mx<0>(wab[WS][0], was[WS], xc0[bt], xc1[bt], acc[0][bt], sb);
                CCSM_FENCE;
                wab[WS][0] = w_at((P + NSA) * PA + ((4 + 0) << 10));
                CCSM_FENCE;
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) acc[1][bt] = mfma_corr_mx<1>(wab[WS][1], was[WS], xc0[bt], xc1[bt], acc[1][bt], sb);
                CCSM_FENCE;
                wab[WS][1] = w_at((P + NSA) * PA + ((4 + 1) << 10));
                was[WS] = ws_at((P + NSA) * PA + (6 << 10));
            } else {
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) {
                    acc[0][bt] = mfma_corr_mx<0>(wab[WS][0], was[WS], xc0[bt], xc1[bt], acc[0][bt], sb);
                    acc[1][bt] = mfma_corr_mx<1>(wab[WS][1], was[WS], xc0[bt], xc1[bt], acc[1][bt], sb);
                }
            }
            CCSM_FENCE;
            if constexpr (P + NSA < NPAIR) { if constexpr (!(kMxIlv && !XD)) ldAb(WS, P + NSA); }
            else if constexpr (P == 13) {       // (hybrid: fragments 6-8 are the fp16 lo of the pair's first k-block)
                wbh[1][1] = w_at(OFF_B + (4 << 10)); wbh[1][2] = w_at(OFF_B + (5 << 10));
                if constexpr (HS3) { wbl[0][0] = w_at(OFF_B + (6 << 10)); wbl[0][1] = w_at(OFF_B + (7 << 10)); wbl[0][2] = w_at(OFF_B + (8 << 10)); }
                else { wbb[0] = w_at(OFF_B + (6 << 10)); wbb[1] = w_at(OFF_B + (7 << 10)); wbb[2] = w_at(OFF_B + (8 << 10)); }
            }
            CCSM_FENCE;
            slot = slot_n;
        });

        stamp(1);
        // ---------------- phase B: R, Z, N += W_h{r,z,n} h_{t-1}  (N starts at b_hn) ---------------------------------
        {
            const f32x16 b3 = bias_set(3);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) acc[2][bt] = b3;
        }
        const int sbh = sb + (s == 0 ? kMxScaleHi0 - kMxScaleHi : 0);
        if constexpr (HS3) {
            // hybrid arithmetic: three fp16 passes per k-block (W_hi h_hi + W_lo h_hi + W_hi h_lo); one pair resident (hi and lo fragments
            // of its two k-blocks), each k-block's six fragments refilled with the next pair's right behind its MFMAs; the last pair's
            // positions take phase C's first pair slot and the hi fragments of its second
            uint4 (&xl)[NB] = xc0;
            static_for<0, kKBH>([&](auto KC) {
                constexpr int KB = decltype(KC)::value;
                constexpr int KBL = KB & 1, Q = KB >> 1;
                constexpr int NXT = OFF_B + (Q + 1) * PB;
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) {
                    xh[bt] = *reinterpret_cast<const uint4*>(smem + mx_hfrag<NB>(KB, bt, 0) + lane * 16);
                    xl[bt] = *reinterpret_cast<const uint4*>(smem + mx_hfrag<NB>(KB, bt, 1) + lane * 16);
                }
                CCSM_FENCE;
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        acc[g][bt] = mfma16(wbh[KBL][g], xh[bt], acc[g][bt]);
                        acc[g][bt] = mfma16(wbl[KBL][g], xh[bt], acc[g][bt]);
                        acc[g][bt] = mfma16(wbh[KBL][g], xl[bt], acc[g][bt]);
                    }
                CCSM_FENCE;
                if constexpr (Q + 1 < kKBH / 2) {
#pragma unroll
                    for (int g = 0; g < 3; ++g) { wbh[KBL][g] = w_at(NXT + ((3 * KBL + g) << 10)); wbl[KBL][g] = w_at(NXT + ((6 + 3 * KBL + g) << 10)); }
                } else if constexpr (KBL == 0) {
                    wch[0][0] = w_at(OFF_C + 0 * PC + (0 << 10)); wch[0][1] = w_at(OFF_C + 0 * PC + (1 << 10)); wcb[0] = w_at(OFF_C + 0 * PC + (2 << 10));
                } else {
                    wcb1[0] = w8_at(OFF_C + 0 * PC + (3 << 10)); wcs[0] = ws_at(OFF_C + 0 * PC + (3 << 10) + 512);
                    wch[1][0] = w_at(OFF_C + 1 * PC + (0 << 10)); wch[1][1] = w_at(OFF_C + 1 * PC + (1 << 10));
                }
                CCSM_FENCE;
            });
        } else
        // pair Q = k-blocks 2Q, 2Q + 1: one pair resident, refilled with the next pair behind each MFMA group (3 + 3 + 4
        // requests); the last pair's positions take phase C's first pair slot
        static_for<0, kKBH / 2>([&](auto QC) {
            constexpr int Q = decltype(QC)::value;
            constexpr bool LAST = Q == kKBH / 2 - 1;
            constexpr int NXT = OFF_B + (Q + 1) * PB;
            uint32_t pm[NB];                                            // split-mx-d: running max |x_hi| of the pair's block, this lane's values
            if constexpr (!(kMxDiag & 4)) {
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                xh[bt] = *reinterpret_cast<const uint4*>(smem + mx_hfrag<NB>(2 * Q, lds_bt(bt), 0) + lane * 16);
                xc0[bt] = *reinterpret_cast<const uint4*>(smem + mx_hfrag<NB>(2 * Q, lds_btc(bt), 1) + lane * 16);
                xc1[bt] = *reinterpret_cast<const uint2*>(smem + mx_hfrag<NB>(2 * Q + 1, lds_btc(bt), 1) + lane * 16);
            }
            }
            constexpr bool ILVB = kMxIlv && !DYN && !LAST && CCSM_PWR_LDS1 == 0;       // (one request behind each gate's three MFMAs: see kMxIlv)
            if constexpr (ILVB) {
                static_for<0, 3>([&](auto GC) {
                    constexpr int g = decltype(GC)::value;
                    CCSM_FENCE;
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt) acc[g][bt] = mfma16(wbh[0][g], xh[bt], acc[g][bt]);
                    CCSM_FENCE;
                    wbh[0][g] = w_at(NXT + (g << 10));
                });
            } else
            CCSM_MAIN(wbh[0], xh, 3, 0);
            if constexpr (DYN) {
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) pm[bt] = absmax8(xh[bt], 0u);
            }
            if constexpr (ILVB) {
            } else if constexpr (!LAST) {
#pragma unroll
                for (int g = 0; g < 3; ++g) wbh[0][g] = w_at(NXT + (g << 10));
            } else {
                wch[0][0] = w_at(OFF_C + 0 * PC + (0 << 10)); wch[0][1] = w_at(OFF_C + 0 * PC + (1 << 10)); wcb[0] = w_at(OFF_C + 0 * PC + (2 << 10));
            }
            if constexpr (!(kMxDiag & 4)) {
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) xh[bt] = *reinterpret_cast<const uint4*>(smem + mx_hfrag<NB>(2 * Q + 1, lds_bt(bt), 0) + lane * 16);
            }
            if constexpr (ILVB) {
                static_for<0, 3>([&](auto GC) {
                    constexpr int g = decltype(GC)::value;
                    CCSM_FENCE;
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt) acc[g][bt] = mfma16(wbh[1][g], xh[bt], acc[g][bt]);
                    CCSM_FENCE;
                    wbh[1][g] = w_at(NXT + ((3 + g) << 10));
                });
            } else
            CCSM_MAIN(wbh[1], xh, 3, 0);
            if constexpr (ILVB) {
            } else if constexpr (!LAST) {
#pragma unroll
                for (int g = 0; g < 3; ++g) wbh[1][g] = w_at(NXT + ((3 + g) << 10));
            } else {
                wcb1[0] = w8_at(OFF_C + 0 * PC + (3 << 10)); wcs[0] = ws_at(OFF_C + 0 * PC + (3 << 10) + 512); wch[1][0] = w_at(OFF_C + 1 * PC + (0 << 10));
            }
            if constexpr (DYN) {
                // the block scales of the three row tiles (vector ALU in the shadow of the main MFMAs above), then the correction products
                int sbd[NB];
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) sbd[bt] = 127 + dyn_block_exp(absmax8(xh[bt], pm[bt])) - (hh ? 11 : 0);
                CCSM_FENCE;
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) {
                    acc[0][bt] = mfma_corr_mx6<0>(wbb[0], wbb1[0], wbs, xc0[bt], xc1[bt], acc[0][bt], sbd[bt]);
                    acc[1][bt] = mfma_corr_mx6<1>(wbb[1], wbb1[1], wbs, xc0[bt], xc1[bt], acc[1][bt], sbd[bt]);
                    acc[2][bt] = mfma_corr_mx6<2>(wbb[2], wbb1[2], wbs, xc0[bt], xc1[bt], acc[2][bt], sbd[bt]);
                }
                CCSM_FENCE;
            } else if constexpr (ILVB) {
                static_for<0, 3>([&](auto GC) {
                    constexpr int g = decltype(GC)::value;
                    CCSM_FENCE;
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt) acc[g][bt] = mfma_corr_mx<g>(wbb[g], wbs, xc0[bt], xc1[bt], acc[g][bt], sbh);
                    CCSM_FENCE;
                    wbb[g] = w_at(NXT + ((6 + g) << 10));
                });
                wbs = ws_at(NXT + OFF_BS);
            } else {
                CCSM_CORR_G(3, wbb, wbs, sbh);
            }
            if constexpr (ILVB) {
            } else if constexpr (!LAST) {
#pragma unroll
                for (int g = 0; g < 3; ++g) wbb[g] = w_at(NXT + ((6 + g) << 10));
                if constexpr (DYN) {
#pragma unroll
                    for (int g = 0; g < 3; ++g) wbb1[g] = w8_at(NXT + (9 << 10) + g * 512);
                }
                wbs = ws_at(NXT + OFF_BS);
            } else {
                wch[1][1] = w_at(OFF_C + 1 * PC + (1 << 10));
            }
            CCSM_FENCE;
        });
        dma_ahead(slot_a15, s, NPAIR - 1);                          // the deferred ring refill: phase-C pair RS - 1
        // r = sigmoid(R) ; N = b_in + r * N
        {
            const f32x16 b2 = bias_set(2);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[2][bt][r] = b2[r] + diag_sigmoid(acc[0][bt][r]) * acc[2][bt][r];
        }
        // the blob of phase C's pair slot 1 and its slots 2 and 3: requested once R is dead (the accumulators drop from 144 to 96
        // registers; slot 0 and the hi fragments of slot 1 came with phase B's last pair)
        CCSM_FENCE;
        wcb[1] = w_at(OFF_C + 1 * PC + (2 << 10)); wcb1[1] = w8_at(OFF_C + 1 * PC + (3 << 10)); wcs[1] = ws_at(OFF_C + 1 * PC + (3 << 10) + 512);
#pragma unroll
        for (int q = 2; q < 4; ++q) {
            wch[q][0] = w_at(OFF_C + q * PC + (0 << 10)); wch[q][1] = w_at(OFF_C + q * PC + (1 << 10)); wcb[q] = w_at(OFF_C + q * PC + (2 << 10));
            wcb1[q] = w8_at(OFF_C + q * PC + (3 << 10)); wcs[q] = ws_at(OFF_C + q * PC + (3 << 10) + 512);
        }
        CCSM_FENCE;
        auto zwork = [&](int bt) {                                  // z = sigmoid(Z) in place, inside phase C (vector ALU otherwise idle)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = diag_sigmoid(acc[1][bt][r]);
#ifndef CCSM_ZWORK_FREE
                asm volatile("" : "+v"(v));                         // pins the evaluation HERE: the compiler otherwise sinks it to its first use, the tail
#endif
                acc[1][bt][r] = v;
            }
        };

        stamp(2);
        // ---------------- phase C: N += W_in x_t, pairs 0..15 (consumptions 16..31) ------------------------------------------
        // pair P lives in slot P % 4 of the n-gate weights, refilled with pair P + 4 (1 + 1 + 3 requests); the last four pairs take
        // the next step's phase-A slots 0 and 1 instead (5, 2, 5, 2 requests; slot 2 follows behind the tail)
        rdx(xh, slot_off(slot), 0, 0);
        static_for<0, NPAIR>([&](auto PC_) {
            constexpr int P = decltype(PC_)::value;
            constexpr int WS = P % 4;
            constexpr int AS = P >= 12 ? (P - 12) / 2 : 0, AF = P >= 12 ? (P - 12) % 2 : 0;   // pairs 12..15: phase-A slot AS, AF = 0 hi / 1 blobs
            const int xs = slot_off(slot);
            const int slot_n = slot == RS - 1 ? 0 : slot + 1;
            rdx(xh1, xs, 1, 0);
            CCSM_PIN_READS;
            CCSM_FENCE;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) acc[2][bt] = mfma16(wch[WS][0], xh[CCSM_PWR_LDS1 == 2 ? 0 : bt], acc[2][bt]);
            if constexpr (CCSM_PWR_LDS1 == 2) { asm volatile("" :: "v"(xh[NB - 1].x), "v"(xh[NB - 1].y), "v"(xh[NB - 1].z), "v"(xh[NB - 1].w), "v"(xh[NB > 1 ? 1 : 0].x), "v"(xh[NB > 1 ? 1 : 0].y), "v"(xh[NB > 1 ? 1 : 0].z), "v"(xh[NB > 1 ? 1 : 0].w)); }
            CCSM_FENCE;
            if constexpr (P + 4 < NPAIR) wch[WS][0] = w_at(OFF_C + (P + 4) * PC + (0 << 10));
            else if constexpr (AF == 0) wah[AS][0][0] = w_at(AS * PA + (0 << 10)); else was[AS] = ws_at(AS * PA + (6 << 10));
            rdx_blob(xs);
            CCSM_PIN_READS;
            CCSM_FENCE;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) acc[2][bt] = mfma16(wch[WS][1], xh1[CCSM_PWR_LDS1 == 2 ? 0 : bt], acc[2][bt]);
            if constexpr (CCSM_PWR_LDS1 == 2) { asm volatile("" :: "v"(xh1[NB - 1].x), "v"(xh1[NB - 1].y), "v"(xh1[NB - 1].z), "v"(xh1[NB - 1].w), "v"(xh1[NB > 1 ? 1 : 0].x), "v"(xh1[NB > 1 ? 1 : 0].y), "v"(xh1[NB > 1 ? 1 : 0].z), "v"(xh1[NB > 1 ? 1 : 0].w)); }
            CCSM_FENCE;
            if constexpr (P + 4 < NPAIR) wch[WS][1] = w_at(OFF_C + (P + 4) * PC + (1 << 10));
            else if constexpr (AF == 0) wah[AS][0][1] = w_at(AS * PA + (1 << 10)); else wab[AS][1] = w_at(AS * PA + (5 << 10));
            // operations since the awaited refill: the 3 requests behind it, two pairs of 5 + d (pairs 13 and 15: 2 + d), 2 of this pair
            if constexpr (P >= NPAIR - 2) CCSM_WAIT_XFER(14, 16); else CCSM_WAIT_XFER(17, 19);
            if constexpr (!(kMxDiag & 2)) __syncthreads();
            dma_ahead(slot, s, NPAIR + P);                              // the vacated slot is refilled at once
            if constexpr (P + 1 < NPAIR) rdx(xh, slot_off(slot_n), 0, 0);
            CCSM_PIN_READS;
            CCSM_FENCE;
            if constexpr (XD) {
                static_for<0, NB>([&](auto BC) {
                    constexpr int bt = decltype(BC)::value;
                    acc[2][bt] = mfma_corr_mx6<0, bt>(wcb[WS], wcb1[WS], wcs[WS], xc0[bt], xc1[bt], acc[2][bt], xsc);
                });
            } else {
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) acc[2][bt] = mfma_corr_mx6<0>(wcb[WS], wcb1[WS], wcs[WS], xc0[bt], xc1[bt], acc[2][bt], sb);
            }
            CCSM_FENCE;
            if constexpr (P + 4 < NPAIR) { wcb[WS] = w_at(OFF_C + (P + 4) * PC + (2 << 10)); wcb1[WS] = w8_at(OFF_C + (P + 4) * PC + (3 << 10));
                                           wcs[WS] = ws_at(OFF_C + (P + 4) * PC + (3 << 10) + 512); }
            else if constexpr (AF == 0) { wah[AS][1][0] = w_at(AS * PA + (2 << 10)); wah[AS][1][1] = w_at(AS * PA + (3 << 10)); wab[AS][0] = w_at(AS * PA + (4 << 10)); }
            CCSM_FENCE;
            slot = slot_n;
            if constexpr (P == 1) zwork(0);
            if constexpr (P == 5 && NB > 1) zwork(1);
            if constexpr (P == 9 && NB > 2) zwork(2);
        });
#undef CCSM_MAIN
        stamp(3);
        // (the hybrid's last layer hands the attention pool fp16 hi + lo fragments - attn_fc_kernel, three passes - instead of fp8 corr
        // fragments: its fp8 tail had no registers left beside the exact state's and spilled, profiles/r03_z_isa.md)
        if constexpr (HS3 && OUT_FP8) f3_tail<NB>(smem, acc[1], acc[2], out, tile0, t, dir, wave, lane16_here());
        else mx_tail<OUT_FP8, HS3, DYN, NB>(smem, kMx12LoOff, acc[1], acc[2], out, tile0, t, dir, wave, lane16_here());
        CCSM_FENCE;
        ldA_slot(2, 2);                                             // the third weight slot of the next step (needed two pairs in): not live across the tail
        if constexpr (NSA == 4) ldA_slot(3, 3);
        CCSM_FENCE;
        stamp(4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // no transfer may still be writing LDS when the workgroup retires
#undef CCSM_WAIT_XFER
}
#undef CCSM_CORR_G
#undef CCSM_FENCE

// Self-test of the split-mx product: C[unit][row] = sum_k W[unit][k] X[row][k] over one pair (32 k).  W fragments packed by the
// host (hi kb0, hi kb1, blob bytes 0-15, [scale dwords with the scale in byte 0 (256 B) | blob bytes 16-23 (512 B)]); with_corr =
// the blob's format (2 fp6, 4 fp4; 0 = main product only).  X given in fp32 as an MFMA-C-layout image and packed on the device with pack_pair_mx;
// blob_out receives the activation blobs (24 bytes per lane) so the host can check its own fp6 encoder against the instruction's.
__global__ void mx_selftest_kernel(const uint4* __restrict__ wfrag, const float* __restrict__ x /* [row 32][k 32] */,
                                   float* __restrict__ c /* [unit 32][row 32] */, uint32_t* __restrict__ blob_out, int with_corr, float scale, int sb_hi, int sb_lo) {
    const int lane = threadIdx.x & 63;
    const int n = lane & 31, hh = lane >> 5;
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] = x[n * 32 + 8 * q + 4 * hh + e];
    uint4 hi0, hi1, c0, lo8;
    uint2 c1;
    pack_pair_mx<true>(v, scale, hi0, hi1, c0, c1, lo8);
    blob_out[lane * 6 + 0] = c0.x; blob_out[lane * 6 + 1] = c0.y; blob_out[lane * 6 + 2] = c0.z; blob_out[lane * 6 + 3] = c0.w;
    blob_out[lane * 6 + 4] = c1.x; blob_out[lane * 6 + 5] = c1.y;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = mfma16(wfrag[0 * 64 + lane], hi0, acc);
    acc = mfma16(wfrag[1 * 64 + lane], hi1, acc);
    const uint32_t ws = reinterpret_cast<const uint32_t*>(wfrag + 3 * 64)[lane];
    if (with_corr == 4) acc = mfma_corr_mx<0>(wfrag[2 * 64 + lane], ws, c0, c1, acc, hh ? sb_lo : sb_hi);
    if (with_corr == 2)
        acc = mfma_corr_mx6<0>(wfrag[2 * 64 + lane], reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(wfrag + 3 * 64) + 256)[lane], ws, c0, c1, acc,
                               hh ? sb_lo : sb_hi);
#pragma unroll
    for (int r = 0; r < 16; ++r) c[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + n] = acc[r];
}

}  // namespace ccsm
