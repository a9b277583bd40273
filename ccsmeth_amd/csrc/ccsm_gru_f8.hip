// libccsm GRU layer, version 3 ("split-f8"): f16 main product + fp8 error-compensation products on the gfx950 block-scaled
// MFMA (v_mfma_scale_f32_32x32x64_f8f6f4).  Included by ccsm_api.hip after ccsm_kernels.hip.
//
// Arithmetic.  Every fp32 operand v is carried as hi = fp16(v) plus the residual lo = v - hi.  The product
//     W x  =  W_hi x_hi  +  W_lo x_hi  +  W_hi x_lo  (+ W_lo x_lo, dropped: 2^-22 relative)
// is issued as
//     main : v_mfma_f32_32x32x16_f16 on (W_hi, x_hi)                      32 cycles per 16 k
//     corr : v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3), K = 64 = [32 k of W_lo x_hi | 32 k of W_hi x_lo]
//                                                                         64 cycles per 32 k
// i.e. 128 MFMA cycles per 32 k instead of the 192 of three f16 passes (CCSM_PRECISION_SPLIT3).  The two correction terms
// are 2^-11 of the main term, so their fp8 operands (4 significant bits) leave a relative error of ~2^-15 per product,
// 16x below plain fp16 operands; measured on the parity suite max |dprob| is ~3e-6 (bar: 1e-4, SPLIT3: 2e-7).
// All sums are fp32 in the same accumulators; the 2^-11 and the operand pre-scales are folded into the instruction's
// E8M0 block scales, so there is no separate accumulator set and no post-scaling.
//
// Fragments.  Per k-block (16 k) there are still two 1 KiB fragments, [hi | corr]:
//   hi   : lane (n, g) holds fp16 M_hi[n][16kb + 8g + j], j < 8                                  (unchanged)
//   corr : lane (n, g) holds 16 fp8 bytes, byte j <-> k = 16kb + kCorrPerm[j];
//          weights     : g = 0 -> fp8(W_lo * 2^11 * sw),  g = 1 -> fp8(W_hi * sw)      (sw: per-matrix power of two)
//          activations : g = 0 -> fp8(x_hi * 64),         g = 1 -> fp8(x_lo * 2^11 * 64)
//   The K = 64 operand of one corr MFMA is the corr fragments of two consecutive k-blocks (8 dwords per lane): lanes
//   g = 0 multiply W_lo by x_hi, lanes g = 1 multiply W_hi by x_lo (A and B share the lane/byte -> k mapping).
//   kCorrPerm = [0 1 2 3 8 9 10 11 4 5 6 7 12 13 14 15] is what two v_permlane32_swap per k-block produce from the MFMA
//   C layout in the step epilogue; the host packs the weights with the same permutation.
// The recurrent state is carried as hi + fp8(lo) too (h_{t-1} is re-read from these fragments): 2^-16 relative per step.
//
// Layer 0 (KX == 1, K = 11 padded to 16, inputs up to hundreds in magnitude) keeps three f16 passes for its x-part.
// The attention pool (attn_fc_f8_kernel below) consumes the same [hi | corr] fragments.
#include <hip/hip_runtime.h>

namespace ccsm {

typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr float kCorrActHi = 64.0f;                // fp8 copy of x_hi carries x_hi * 2^6
constexpr float kCorrActLo = 2048.0f * 64.0f;      // fp8 copy of x_lo carries x_lo * 2^17
constexpr int kCorrScaleB = 127 - 6;               // E8M0 block scale of the activation operand: 2^-6
constexpr float kF8Clamp = 448.0f;                 // e4m3 finite maximum (v_cvt_pk_fp8_f32 does not saturate)

__device__ __forceinline__ f32x16 mfma_corr(uint4 w0, uint4 w1, uint4 x0, uint4 x1, f32x16 c, int scale_a) {
    const i32x8 a = {(int)w0.x, (int)w0.y, (int)w0.z, (int)w0.w, (int)w1.x, (int)w1.y, (int)w1.z, (int)w1.w};
    const i32x8 b = {(int)x0.x, (int)x0.y, (int)x0.z, (int)x0.w, (int)x1.x, (int)x1.y, (int)x1.z, (int)x1.w};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, kCorrScaleB);
}

__device__ __forceinline__ uint32_t cvt4_fp8(float a, float b, float c, float d) {
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (uint32_t)r;
}
__device__ __forceinline__ void swap32(uint32_t& x, uint32_t& y) {   // lanes 32-63 of x <-> lanes 0-31 of y
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}
typedef short short2v __attribute__((ext_vector_type(2)));
// v_cvt_scalef32_pk_fp8_{f16,f32}: dst = fp8(src / scale) (measured, tools/ubench: scale 2^-6 -> x 64), i.e. the operand
// pre-scale costs no multiply.  `seed` only supplies the register whose other half the first conversion leaves in place.
__device__ __forceinline__ uint32_t cvt4_fp8_h(uint32_t h01, uint32_t h23) {          // four fp16 (two packed pairs) * 2^6
    short2v r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(__builtin_bit_cast(short2v, h01), __builtin_bit_cast(half2v, h01),
                                                         1.0f / kCorrActHi, false);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(half2v, h23), 1.0f / kCorrActHi, true);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t cvt4_fp8_l(uint32_t seed, float a, float b, float c, float d) {   // four fp32 * 2^17
    short2v r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_bit_cast(short2v, seed), a, b, 1.0f / kCorrActLo, false);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, c, d, 1.0f / kCorrActLo, true);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float tanh_fold(float x) {    // 1 - 2 / (e^{2x} + 1) with the doubling folded into the exp2 constant
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.8853900817779268f) + 1.0f);
}

// Eight MFMA-C-layout values of one k-block held by lane (n, hh): v[0..3] = units e + 4hh, v[4..7] = units 8 + e + 4hh of
// batch row n.  Produces this lane's 16 bytes of the k-block's hi fragment and corr fragment.
__device__ __forceinline__ void pack_kb(const float (&v)[8], uint4& hi_out, uint4& co_out) {
    typedef _Float16 half2p __attribute__((ext_vector_type(2)));
    uint32_t hp[4];
    float lf[8];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const half2p h = {(_Float16)v[j], (_Float16)v[j + 1]};
        hp[j >> 1] = __builtin_bit_cast(uint32_t, h);
        lf[j] = v[j] - (float)h[0];
        lf[j + 1] = v[j + 1] - (float)h[1];
    }
    // The fp8 conversions do not saturate (an overflow is a NaN in the product), and the GRU state is only bounded by max(1, |h0|):
    // the values that go into the CORRECTION operands are clamped to the e4m3 range (|x_hi| <= 7, |x_lo| <= 448 / 2^17) first.
    uint32_t hc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const half2p lim = {(_Float16)(kF8Clamp / kCorrActHi), (_Float16)(kF8Clamp / kCorrActHi)};
        half2p t = __builtin_bit_cast(half2p, hp[j]);
        t = __builtin_elementwise_min(__builtin_elementwise_max(t, -lim), lim);
        hc[j] = __builtin_bit_cast(uint32_t, t);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) lf[j] = __builtin_amdgcn_fmed3f(lf[j], -kF8Clamp / kCorrActLo, kF8Clamp / kCorrActLo);
    uint32_t h0 = cvt4_fp8_h(hc[0], hc[1]);
    uint32_t h1 = cvt4_fp8_h(hc[2], hc[3]);
    uint32_t l0 = cvt4_fp8_l(hp[0], lf[0], lf[1], lf[2], lf[3]);
    uint32_t l1 = cvt4_fp8_l(hp[2], lf[4], lf[5], lf[6], lf[7]);
    uint32_t a0 = hp[0], a1 = hp[1], b0 = hp[2], b1 = hp[3];
    swap32(a0, b0);
    swap32(a1, b1);
    hi_out = make_uint4(a0, a1, b0, b1);
    swap32(h0, l0);   // lower lanes: (own hi, partner's hi) ; upper lanes: (partner's lo, own lo)
    swap32(h1, l1);
    co_out = make_uint4(h0, h1, l0, l1);
}

// ---------------------------------------------------------------------------------------------------------
// Attention pool + FC partials in split-f8 arithmetic: attn_fc_kernel (ccsm_kernels.hip) with the activation and the
// Wa / Ua fragments in [hi | corr] form.  One staged chunk (2 k-blocks) is exactly one pair: per timestep two main MFMAs and
// one K = 64 corr MFMA instead of six fp16 MFMAs.  The fc1 partial dot products need the activations themselves: hi from the
// main fragment in registers, the fp8 residual from the staged corr fragment (lane (n, 1), kCorrPerm order).
//   wa / ua : [wave][kb 32][hi|corr][64] uint4 ; sa_wa / sa_ua : E8M0 scales of their corr operands
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void attn_fc_f8_kernel(const uint4* __restrict__ out2, const uint4* __restrict__ wa,
                                                             const uint4* __restrict__ ua, const float* __restrict__ va,
                                                             const float* __restrict__ fcw, float* __restrict__ part,
                                                             SliceTable slices, int sa_wa, int sa_ua) {
    constexpr int TG = 7;                      // timesteps per Ua pass (21 = 3 * 7)
    constexpr int CK = 2;                      // k-blocks per staged chunk = one pair
    constexpr int NCHUNK = kKB12 / CK;
    constexpr int CHUNK_FRAGS = CK * TG * 2;   // 28 fragments of 1 KiB
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // 79.9 KiB in all (kAttF8Lds; round 2 kept per-wave score partials of all 21 timesteps and a per-wave fc1 array that was never
    // indexed by wave: 129 KiB).  Occupancy is set by the registers (seven accumulator tiles: 250 VGPRs, two waves per SIMD, one
    // workgroup per CU), not by this
    char* s_stage = smem;                                                     // [2][CK][TG][hi|corr] fragments
    float* s_epart = reinterpret_cast<float*>(smem + 2 * CHUNK_FRAGS * 1024);  // [wave][tt][32]: this timestep group's score partials
    float* s_e = s_epart + kWaves * TG * 32;                                   // [t][32]: scores, summed over the waves in a fixed order
    float* s_pfc = s_e + kSeqLen * 32;                                         // [t][32][2]
    float* s_fcw = s_pfc + kSeqLen * 32 * 2;                                   // [2][1024]
    float* s_va = s_fcw + kClasses * 4 * kHidden;                              // [256]: in LDS, the 16 registers go to the operand pipeline

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x;
    const int n = lane & 31, hh = lane >> 5;

    for (int i = threadIdx.x; i < kClasses * 4 * kHidden; i += blockDim.x) s_fcw[i] = fcw[i];
    if (threadIdx.x < kHidden) s_va[threadIdx.x] = va[threadIdx.x];

    const uint4* wap = wa + (size_t)wave * kKB12 * 2 * kFragU4 + lane;
    const uint4* uap = ua + (size_t)wave * kKB12 * 2 * kFragU4 + lane;
    const uint4* otile = out2 + (size_t)tile * kSeqLen * kKB12 * 2 * kFragU4;   // [t][kb][hi|corr][64]

    // ---- q = Wa h_n, h_n = [fwd final state = out[t=L-1][0:256] | bwd final state = out[t=0][256:512]] (models.py:135-137)
    f32x16 qacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) qacc[r] = 0.f;
    {   // operands of pairs p+1 .. p+3 are in flight while pair p multiplies (four register sets of 8 fragments: the activation
        // fragments come from HBM, one pair of look-ahead left the loop at one round trip per pair)
        uint4 qa[4][8];
        auto ldq = [&](uint4 (&d)[8], int kb) {
            const int tq = kb < kKBH ? kSeqLen - 1 : 0;
            const uint4* xp = otile + ((size_t)tq * kKB12 + kb) * 2 * kFragU4 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d[i] = wap[(kb * 2 + i) * kFragU4];           // hi0, corr0, hi1, corr1
                d[4 + i] = xp[i * kFragU4];
            }
        };
        ldq(qa[0], 0);
        ldq(qa[1], 2);
        ldq(qa[2], 4);
#pragma unroll
        for (int p = 0; p < kKB12 / 2; ++p) {
            if (p + 3 < kKB12 / 2) ldq(qa[(p + 3) & 3], 2 * (p + 3));
            asm volatile("" ::: "memory");
            const uint4(&d)[8] = qa[p & 3];
            qacc = mfma16(d[0], d[4], qacc);
            qacc = mfma16(d[2], d[6], qacc);
            qacc = mfma_corr(d[1], d[3], d[5], d[7], qacc, sa_wa);
            asm volatile("" ::: "memory");
        }
    }
    int strand = 0;   // which half of fc1.weight this lane's row multiplies: strand 2 rows are the upper half of a slice
    {
        const int row = tile * 32 + n;
        for (int i = 0; i < slices.count; ++i)
            if (row >= slices.row_base[i] && row < slices.row_base[i] + 2 * slices.n_sites[i])
                strand = (row - slices.row_base[i]) >= slices.n_sites[i];
    }

    // stage chunk `c` of timestep group t0 into buffer `buf`: fragment f = (kbl * TG + tt) * 2 + hl
    auto stage = [&](int t0, int c, int buf) {
#pragma unroll
        for (int i = 0; i < (CHUNK_FRAGS + kWaves - 1) / kWaves; ++i) {
            const int f = wave + kWaves * i;
            if (f < CHUNK_FRAGS) {
                const int hl = f & 1, tt = (f >> 1) % TG, kbl = (f >> 1) / TG;
                const uint4* src = otile + (((size_t)(t0 + tt) * kKB12 + (c * CK + kbl)) * 2 + hl) * kFragU4 + lane;
                dma16(src, __builtin_amdgcn_readfirstlane(
                               (unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_stage + (buf * CHUNK_FRAGS + f) * 1024)));
            }
        }
    };

    uint4 wu[CK][2];       // Ua fragments of the next chunk, requested one chunk ahead
#pragma unroll
    for (int kbl = 0; kbl < CK; ++kbl) { wu[kbl][0] = uap[(kbl * 2 + 0) * kFragU4]; wu[kbl][1] = uap[(kbl * 2 + 1) * kFragU4]; }

    for (int tg = 0; tg < kSeqLen / TG; ++tg) {
        const int t0 = tg * TG;
        f32x16 kacc[TG];
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) kacc[tt] = qacc;     // accumulate K_t on top of q
        // fc1 partials: wave w takes timestep t0 + w of the group for ALL k-blocks (w < 7): the work of every chunk is spread
        // over seven waves instead of falling on the two that own its k-blocks (they made the other six wait at the barrier),
        // and each (row, t) sum is complete in one wave's registers, in a fixed order
        float pf0 = 0.f, pf1 = 0.f;

        stage(t0, 0, 0);
#pragma unroll 1
        for (int c = 0; c < NCHUNK; ++c) {
            uint4 w[CK][2];
#pragma unroll
            for (int kbl = 0; kbl < CK; ++kbl) { w[kbl][0] = wu[kbl][0]; w[kbl][1] = wu[kbl][1]; }
            wait_dma();        // this wave's part of chunk c (and the Ua fragments above) has arrived ...
            __syncthreads();   // ... and so has everybody else's: chunk c is in LDS; buffer (c+1)&1 is free
            if (c + 1 < NCHUNK) stage(t0, c + 1, (c + 1) & 1);
            {
                const int cn = c + 1 < NCHUNK ? c + 1 : 0;       // Ua is re-streamed for every timestep group
#pragma unroll
                for (int kbl = 0; kbl < CK; ++kbl) {
                    wu[kbl][0] = uap[((cn * CK + kbl) * 2 + 0) * kFragU4];
                    wu[kbl][1] = uap[((cn * CK + kbl) * 2 + 1) * kFragU4];
                }
            }
            const char* sb0 = s_stage + (c & 1) * CHUNK_FRAGS * 1024;
            const char* sb = sb0 + lane * 16;
            // operands of timestep tt + 1 are read from LDS before the three MFMAs of timestep tt are issued (two register sets,
            // pinned with compiler fences: left alone the compiler reads each operand right in front of its MFMA and every MFMA
            // waits out an LDS round trip — 4.2 k cycles per chunk for 1.85 k of matrix work)
            uint4 xo[2][4];
            auto rdop = [&](uint4 (&d)[4], int tt) {
                d[0] = *reinterpret_cast<const uint4*>(sb + ((0 * TG + tt) * 2 + 0) * 1024);
                d[1] = *reinterpret_cast<const uint4*>(sb + ((0 * TG + tt) * 2 + 1) * 1024);
                d[2] = *reinterpret_cast<const uint4*>(sb + ((1 * TG + tt) * 2 + 0) * 1024);
                d[3] = *reinterpret_cast<const uint4*>(sb + ((1 * TG + tt) * 2 + 1) * 1024);
            };
            rdop(xo[0], 0);
#pragma unroll
            for (int tt = 0; tt < TG; ++tt) {
                asm volatile("" ::: "memory");
                if (tt + 1 < TG) rdop(xo[(tt + 1) & 1], tt + 1);
                asm volatile("" ::: "memory");
                const uint4 x0h = xo[tt & 1][0], x0c = xo[tt & 1][1], x1h = xo[tt & 1][2], x1c = xo[tt & 1][3];
                kacc[tt] = mfma16(w[0][0], x0h, kacc[tt]);
                kacc[tt] = mfma16(w[1][0], x1h, kacc[tt]);
                kacc[tt] = mfma_corr(w[0][1], w[1][1], x0c, x1c, kacc[tt], sa_ua);
                // fc1 partial of k-block (tt & 1) for timestep t0 + wave, issued behind this timestep's MFMAs (tt < 2): vector ALU
                // and LDS work in the shadow of the matrix pipe instead of a serial block at the end of the chunk
                if (tt < CK && wave < TG) {
                    const int kbl = tt;
                    const int kb = c * CK + kbl;
                    const char* fr = sb0 + ((kbl * TG + wave) * 2) * 1024;
                    const half8 xh = as_half8(*reinterpret_cast<const uint4*>(fr + lane * 16));
                    const int la = *reinterpret_cast<const int*>(fr + 1024 + (n + 32) * 16 + 4 * hh);
                    const int lb = *reinterpret_cast<const int*>(fr + 1024 + (n + 32) * 16 + 8 + 4 * hh);
                    const float xl[8] = {__builtin_amdgcn_cvt_f32_fp8(la, 0), __builtin_amdgcn_cvt_f32_fp8(la, 1),
                                         __builtin_amdgcn_cvt_f32_fp8(la, 2), __builtin_amdgcn_cvt_f32_fp8(la, 3),
                                         __builtin_amdgcn_cvt_f32_fp8(lb, 0), __builtin_amdgcn_cvt_f32_fp8(lb, 1),
                                         __builtin_amdgcn_cvt_f32_fp8(lb, 2), __builtin_amdgcn_cvt_f32_fp8(lb, 3)};
                    const float4* f0 = reinterpret_cast<const float4*>(s_fcw + 0 * 4 * kHidden + strand * 2 * kHidden + kb * 16 + hh * 8);
                    const float4* f1 = reinterpret_cast<const float4*>(s_fcw + 1 * 4 * kHidden + strand * 2 * kHidden + kb * 16 + hh * 8);
                    const float4 a0 = f0[0], a1 = f0[1], b0 = f1[0], b1 = f1[1];
                    const float fa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    const float fb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float xv = (float)xh[j] + xl[j] * (1.0f / kCorrActLo);
                        pf0 += fa[j] * xv;
                        pf1 += fb[j] * xv;
                    }
                }
            }
        }
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) {
            float e = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) e += s_va[(wave * 2 + hh) * 16 + r] * tanh_f(kacc[tt][r]);
            e += __shfl_xor(e, 32);
            if (hh == 0) s_epart[(wave * TG + tt) * 32 + n] = e;
        }
        if (wave < TG) {
            pf0 += __shfl_xor(pf0, 32);
            pf1 += __shfl_xor(pf1, 32);
            if (hh == 0) {
                s_pfc[((t0 + wave) * 32 + n) * 2 + 0] = pf0;
                s_pfc[((t0 + wave) * 32 + n) * 2 + 1] = pf1;
            }
        }
        __syncthreads();   // all waves are done with both staging buffers before the next group restages buffer 0
        if (threadIdx.x < TG * 32) {       // the group's scores: the eight waves' partials in a fixed order (deterministic)
            const int tt = threadIdx.x >> 5, rl = threadIdx.x & 31;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) v += s_epart[(w * TG + tt) * 32 + rl];
            s_e[(t0 + tt) * 32 + rl] = v;
        }                                  // (s_epart is next written at the end of the next group, a chunk loop of barriers away)
    }
    __syncthreads();

    // ---- softmax over t and the strand-half of the logits (fixed summation order: deterministic)
    if (threadIdx.x < 32) {
        const int rl = threadIdx.x;
        const int row = tile * 32 + rl;
        float e[kSeqLen];
        float m = -3.0e38f;
#pragma unroll
        for (int t = 0; t < kSeqLen; ++t) {
            const float v = s_e[t * 32 + rl];
            e[t] = v;
            m = fmaxf(m, v);
        }
        float den = 0.f;
#pragma unroll
        for (int t = 0; t < kSeqLen; ++t) { e[t] = __expf(e[t] - m); den += e[t]; }
        const float inv = 1.0f / den;
        float l0 = 0.f, l1 = 0.f;
#pragma unroll
        for (int t = 0; t < kSeqLen; ++t) {
            const float a = e[t] * inv;
            l0 += a * s_pfc[(t * 32 + rl) * 2 + 0];
            l1 += a * s_pfc[(t * 32 + rl) * 2 + 1];
        }
        part[(size_t)row * 2 + 0] = l0;
        part[(size_t)row * 2 + 1] = l1;
    }
}

// Self-test of the split-f8 product: C[unit][row] = sum_k W[unit][k] X[row][k] over 32 k, W fragments packed by the host
// (hi, corr of two k-blocks), X given in fp32 and packed on the device with pack_kb from an MFMA-C-layout register image.
__global__ void corr_selftest_kernel(const uint4* __restrict__ wfrag /* [kb 2][hi|corr][64] */, const float* __restrict__ x /* [row 32][k 32] */,
                                     float* __restrict__ c /* [unit 32][row 32] */, int scale_a, int with_corr) {
    const int lane = threadIdx.x & 63;
    const int n = lane & 31, hh = lane >> 5;
    uint4 xh[2], xc[2];
#pragma unroll
    for (int kbl = 0; kbl < 2; ++kbl) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = x[n * 32 + 16 * kbl + e + 4 * hh];
            v[4 + e] = x[n * 32 + 16 * kbl + 8 + e + 4 * hh];
        }
        pack_kb(v, xh[kbl], xc[kbl]);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = mfma16(wfrag[(0 * 2 + 0) * 64 + lane], xh[0], acc);
    acc = mfma16(wfrag[(1 * 2 + 0) * 64 + lane], xh[1], acc);
    if (with_corr) acc = mfma_corr(wfrag[(0 * 2 + 1) * 64 + lane], wfrag[(1 * 2 + 1) * 64 + lane], xc[0], xc[1], acc, scale_a);
#pragma unroll
    for (int r = 0; r < 16; ++r) c[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + n] = acc[r];
}

}  // namespace ccsm
