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
// Attention pool + FC partials in split-f8 arithmetic: attn_fc_kernel (ccsm_kernels.hip) with the Wa / Ua fragments in [hi | corr]
// form and the activations as the last GRU layer writes them for this kernel (mx_tail<OUT_FP8>): per pair of k-blocks the two fp16 hi
// fragments and ONE compact fp8 residual fragment (lane (n, g) = the 16 residual bytes of k-block g, byte j <-> k = kCorrPerm[j]) -
// 3 bytes per element instead of the 4 of two [fp8 x_hi | fp8 x_lo] corr fragments: the pool streams its input once at the rate the
// CU's outstanding-request capacity allows, so its time is its bytes.  The fp8 copy of x_hi that the K = 64 correction
// MFMA wants in its lower lanes is re-derived from the staged hi fragments, ONCE per staged chunk and workgroup (one 16-value item per
// thread), into a third LDS region; a lane reads its half of a corr operand from
//     base_g + tt * 1024 + kbl * 512 + n * 16,   base_0 = the derived region (CV), base_1 = the staged residual fragments (LO).
// One staged chunk (2 k-blocks) is one pair: per timestep two main MFMAs and one K = 64 corr MFMA.  Three staging buffers: while chunk c
// multiplies, chunk c + 1 (landed) is converted and chunk c + 2 is in flight; one barrier per chunk as before.
//   out2 : [tile][t][32 kb][hi | corr][64] uint4 - of each pair only hi (kb0), lo (in kb0's corr slot) and hi (kb1) are written / read
//   wa / ua : [wave][kb 32][hi|corr][64] uint4 ; sa_wa / sa_ua : E8M0 scales of their corr operands
// ---------------------------------------------------------------------------------------------------------
constexpr int kAttF8TG = 7;                                          // timesteps per Ua pass (21 = 3 * 7)
constexpr int kAttF8Frags = kAttF8TG * 3;                            // staged fragments per chunk: [kbl 2][tt] hi, then [tt] lo
constexpr int kAttF8Buf = (kAttF8Frags + kAttF8TG) * 1024;           // + the derived region: [tt][kbl][32 rows] x 16 B = 7 KiB
// packed pair of fp16 clamped to the e4m3 range of the x_hi copy (|x_hi| * 64 <= 448): the state is only bounded by max(1, |h0|) and the
// conversion does not saturate (an overflow would be a NaN in the product)
__device__ __forceinline__ uint32_t clamp_hi2(uint32_t h) {
    typedef _Float16 half2p __attribute__((ext_vector_type(2)));
    const half2p lim = {(_Float16)(kF8Clamp / kCorrActHi), (_Float16)(kF8Clamp / kCorrActHi)};
    half2p t = __builtin_bit_cast(half2p, h);
    t = __builtin_elementwise_min(__builtin_elementwise_max(t, -lim), lim);
    return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ uint32_t cvt4_fp8_hc(uint32_t a, uint32_t b) { return cvt4_fp8_h(clamp_hi2(a), clamp_hi2(b)); }
__device__ __forceinline__ uint4 cvt16_fp8_h(uint4 own, uint4 prt) {   // 16 fp16 of one k-block (own: k 0-7, prt: k 8-15) -> fp8 x 64, kCorrPerm order
    return make_uint4(cvt4_fp8_hc(own.x, own.y), cvt4_fp8_hc(prt.x, prt.y), cvt4_fp8_hc(own.z, own.w), cvt4_fp8_hc(prt.z, prt.w));
}
// lane id recomputed where it is needed (two instructions, no input): what the loops' epilogues index with, instead of values the
// compiler would have to keep through the accumulator-bound main loop (they were spilled)
__device__ __forceinline__ int lane_now() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__global__ __launch_bounds__(512, 2) void attn_fc_f8_kernel(const uint4* __restrict__ out2, const uint4* __restrict__ wa,
                                                             const uint4* __restrict__ ua, const float* __restrict__ va,
                                                             const uint4* __restrict__ fc3, float* __restrict__ part,
                                                             SliceTable slices, int sa_wa, int sa_ua, int sa_fc) {
    constexpr int TG = kAttF8TG;
    constexpr int CK = 2;                      // k-blocks per staged chunk = one pair
    constexpr int NCHUNK = kKB12 / CK;
    constexpr int NFR = kAttF8Frags;           // 21 fragments of 1 KiB per chunk
    constexpr int BUF = kAttF8Buf;             // 28 KiB: H [kbl][tt] 14 KiB | LO [tt] 7 KiB | CV [tt][kbl][n] 7 KiB
    constexpr int LO_OFF = 2 * TG * 1024, CV_OFF = 3 * TG * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_stage = smem;                                                     // [3] buffers
    float* s_epart = reinterpret_cast<float*>(smem + 3 * BUF);                 // [wave][tt][32]: this timestep group's score partials
    float* s_e = s_epart + kWaves * TG * 32;                                   // [t][32]: scores, summed over the waves in a fixed order
    float* s_pfc = s_e + kSeqLen * 32;                                         // [t][32][2]
    uint4* s_fc3 = reinterpret_cast<uint4*>(s_pfc + kSeqLen * 32 * 2);         // fc1.weight as compact A fragments [kb 32][hi|corr][q 2][i 4] + one zero line (pack_fc_v3)
    float* s_va = reinterpret_cast<float*>(s_fc3 + kAttFc3);                   // [256]
    float4* s_q = reinterpret_cast<float4*>(s_va + kHidden);                   // [wave][4][64] float4: q of this wave's units (re-read per timestep group:
                                                                               // the accumulators of the group leave it no 16 registers)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x;
    const int n = lane & 31, hh = lane >> 5;

    for (int i = threadIdx.x; i < kAttFc3; i += blockDim.x) s_fc3[i] = fc3[i];
    if (threadIdx.x < kHidden) s_va[threadIdx.x] = va[threadIdx.x];

    const uint4* wap = wa + (size_t)wave * kKB12 * 2 * kFragU4 + lane;
    const uint4* otile = out2 + (size_t)tile * kSeqLen * kKB12 * 2 * kFragU4;   // [t][kb][hi|corr][64]

    // ---- q = Wa h_n, h_n = [fwd final state = out[t=L-1][0:256] | bwd final state = out[t=0][256:512]] (models.py:135-137)
    f32x16 qacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) qacc[r] = 0.f;
    {   // operands of pairs p+1 .. p+2 are in flight while pair p multiplies.  Per pair and lane: the Wa fragments, the own hi fragments and
        // two more 16-byte pieces - lower lanes: the partner's hi (k 8-15) of both k-blocks, to derive the fp8 copy; upper lanes: the residual
        // bytes of both k-blocks (lanes n and n + 32 of the compact fragment)
        uint4 qa[3][8];
        auto ldq = [&](uint4 (&d)[8], int kb) {
            const int tq = kb < kKBH ? kSeqLen - 1 : 0;
            const uint4* xb = otile + ((size_t)tq * kKB12 + kb) * 2 * kFragU4;      // hi(kb) | lo(pair) | hi(kb + 1) | -
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = wap[(kb * 2 + i) * kFragU4];           // hi0, corr0, hi1, corr1
            d[4] = xb[lane];
            d[6] = xb[2 * kFragU4 + lane];
            d[5] = hh ? xb[kFragU4 + n] : xb[n + 32];
            d[7] = hh ? xb[kFragU4 + n + 32] : xb[2 * kFragU4 + n + 32];
        };
        ldq(qa[0], 0);
        ldq(qa[1], 2);
#pragma unroll
        for (int p = 0; p < kKB12 / 2; ++p) {
            if (p + 2 < kKB12 / 2) ldq(qa[(p + 2) % 3], 2 * (p + 2));
            asm volatile("" ::: "memory");
            const uint4(&d)[8] = qa[p % 3];
            const uint4 c0 = cvt16_fp8_h(d[4], d[5]), c1 = cvt16_fp8_h(d[6], d[7]);
            const uint4 x0c = hh ? d[5] : c0, x1c = hh ? d[7] : c1;
            qacc = mfma16(d[0], d[4], qacc);
            qacc = mfma16(d[2], d[6], qacc);
            qacc = mfma_corr(d[1], d[3], x0c, x1c, qacc, sa_wa);
            asm volatile("" ::: "memory");
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s_q[(wave * 4 + i) * 64 + lane] = make_float4(qacc[4 * i], qacc[4 * i + 1], qacc[4 * i + 2], qacc[4 * i + 3]);
    int strand = 0;   // which half of fc1.weight this lane's row multiplies: strand 2 rows are the upper half of a slice
    {
        const int row = tile * 32 + n;
        for (int i = 0; i < slices.count; ++i)
            if (row >= slices.row_base[i] && row < slices.row_base[i] + 2 * slices.n_sites[i])
                strand = (row - slices.row_base[i]) >= slices.n_sites[i];
    }

    // stage chunk `c` of timestep group t0 into buffer `buf`: fragments 0..13 = hi [kbl][tt], 14..20 = lo [tt]
    const u32x4_t ors = dma_rsrc(otile);
    const __amdgpu_buffer_rsrc_t urs = make_rsrc(ua + (size_t)wave * kKB12 * 2 * kFragU4);
    auto stage = [&](int t0, int c, int buf) {
#pragma unroll
        for (int i = 0; i < (NFR + kWaves - 1) / kWaves; ++i) {
            const int f = wave + kWaves * i;
            if (f < NFR) {
                const int lo = f >= 2 * TG;
                const int tt = lo ? f - 2 * TG : f % TG, kbl = lo ? 0 : f / TG;
                // (descriptor + wave-uniform byte offset + lane * 16: no 64-bit per-lane address kept through the chunk loop - they were spilled)
                const int soff = ((((t0 + tt) * kKB12 + (c * CK + kbl)) * 2 + lo) * kFragU4) * 16;
                dma16_buf(ors, lane_now() * 16, __builtin_amdgcn_readfirstlane(soff),
                          __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_stage + buf * BUF + f * 1024)));
            }
        }
    };
    // derive the fp8 copy of x_hi of a landed chunk: item = (tt, kbl, row), one per thread (448 of 512)
    auto convert = [&](int buf) {
        const int it = threadIdx.x;
        if (it < TG * 2 * 32) {
            const int tt = it >> 6, kbl = (it >> 5) & 1, r = it & 31;
            const char* hf = s_stage + buf * BUF + (kbl * TG + tt) * 1024 + r * 16;
            char* cv = s_stage + buf * BUF + CV_OFF + tt * 1024 + kbl * 512 + r * 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {                  // in two halves (8 bytes of the result each): few registers live at a time
                const uint2 own = *reinterpret_cast<const uint2*>(hf + 8 * h);
                const uint2 prt = *reinterpret_cast<const uint2*>(hf + 512 + 8 * h);
                *reinterpret_cast<uint2*>(cv + 8 * h) = make_uint2(cvt4_fp8_hc(own.x, own.y), cvt4_fp8_hc(prt.x, prt.y));
                asm volatile("" ::: "memory");
            }
        }
    };

    uint4 wu[CK][2];       // Ua fragments of the next chunk, requested one chunk ahead
#pragma unroll
    for (int kbl = 0; kbl < CK; ++kbl) { wu[kbl][0] = buf_load(urs, lane * 16, (kbl * 2 + 0) * 1024); wu[kbl][1] = buf_load(urs, lane * 16, (kbl * 2 + 1) * 1024); }

    for (int tg = 0; tg < kSeqLen / TG; ++tg) {
        const int t0 = tg * TG;
        f32x16 kacc[TG];
        {
            f32x16 q;                                        // (written by this lane itself: no barrier needed)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 v = s_q[(wave * 4 + i) * 64 + lane];
                q[4 * i] = v.x; q[4 * i + 1] = v.y; q[4 * i + 2] = v.z; q[4 * i + 3] = v.w;
            }
#pragma unroll
            for (int tt = 0; tt < TG; ++tt) kacc[tt] = q;    // accumulate K_t on top of q
        }
        // fc1 partials: wave w takes timestep t0 + w of the group for ALL k-blocks (w < 7), as three more MFMAs per chunk on the operands it
        // holds for that timestep anyway: fc1.weight is the A operand of a 33rd unit tile whose rows 0-3 are (class, strand half) and whose
        // other rows are zero (lanes n >= 4 read the zero line; pack_fc_v3).  Rounds 1-4 did this sum on the vector ALU: ~120 instructions
        // per chunk that made waves 0-6 take 2160 cycles per chunk against wave 7's 1160 (profiles/r04_j_attn_stamps_before.log).  Each
        // (row, t) sum is complete in one wave's registers, in a fixed order.
        f32x16 facc;
#pragma unroll
        for (int r = 0; r < 16; ++r) facc[r] = 0.f;

        stage(t0, 0, 0);
        stage(t0, 1, 1);
        wait_dma();
        __syncthreads();           // chunks 0 and 1 are in LDS (and every wave is done with the previous group's buffers: barrier at its end)
        convert(0);
#pragma unroll 1
        for (int c = 0; c < NCHUNK; ++c) {
            uint4 w[CK][2];
#pragma unroll
            for (int kbl = 0; kbl < CK; ++kbl) { w[kbl][0] = wu[kbl][0]; w[kbl][1] = wu[kbl][1]; }
            wait_dma();        // this wave's part of chunk c + 1 (and the Ua fragments above) has arrived ...
            __syncthreads();   // ... and everybody else's; chunk c's derived region is complete; buffer (c + 2) % 3 (chunk c - 1) is free
            if (c + 2 < NCHUNK) stage(t0, c + 2, (c + 2) % 3);
            if (c + 1 < NCHUNK) convert((c + 1) % 3);
            {
                const int cn = c + 1 < NCHUNK ? c + 1 : 0;       // Ua is re-streamed for every timestep group
#pragma unroll
                for (int kbl = 0; kbl < CK; ++kbl) {
                    wu[kbl][0] = buf_load(urs, lane_now() * 16, ((cn * CK + kbl) * 2 + 0) * 1024);
                    wu[kbl][1] = buf_load(urs, lane_now() * 16, ((cn * CK + kbl) * 2 + 1) * 1024);
                }
            }
            const char* sb0 = s_stage + (c % 3) * BUF;
            const char* sb = sb0 + lane * 16;
            const char* sc = sb0 + (hh ? LO_OFF : CV_OFF) + n * 16;     // this lane's half of the corr operands
            // the hi operands of timestep tt + 1 are read from LDS before the MFMAs of timestep tt are issued (two register sets); the corr
            // operands of timestep tt are read in front of its two main MFMAs and used behind them (one set: the kernel has no register to spare)
            uint4 xo[2][2], xc[2];
            auto rdhi = [&](uint4 (&d)[2], int tt) {
                d[0] = *reinterpret_cast<const uint4*>(sb + (0 * TG + tt) * 1024);
                d[1] = *reinterpret_cast<const uint4*>(sb + (1 * TG + tt) * 1024);
            };
            rdhi(xo[0], 0);
#pragma unroll
            for (int tt = 0; tt < TG; ++tt) {
                asm volatile("" ::: "memory");
                xc[0] = *reinterpret_cast<const uint4*>(sc + tt * 1024);
                xc[1] = *reinterpret_cast<const uint4*>(sc + tt * 1024 + 512);
                if (tt + 1 < TG) rdhi(xo[(tt + 1) & 1], tt + 1);
                asm volatile("" ::: "memory");
                const uint4 x0h = xo[tt & 1][0], x1h = xo[tt & 1][1];
                kacc[tt] = mfma16(w[0][0], x0h, kacc[tt]);
                kacc[tt] = mfma16(w[1][0], x1h, kacc[tt]);
                kacc[tt] = mfma_corr(w[0][1], w[1][1], xc[0], xc[1], kacc[tt], sa_ua);
                if (wave == tt) {         // this wave's timestep of the fc1 partials: the same operands against the fc1 tile
                    const int lf = lane_now(), fl = (lf & 31) < 4 ? (lf >> 5) * 4 + (lf & 31) : -1;
                    auto fcw = [&](int kbl, int hl) -> uint4 { return s_fc3[fl < 0 ? kAttFc3 - 1 : ((c * CK + kbl) * 2 + hl) * 8 + fl]; };
                    {   // (one or two fragments at a time: the kernel has no sixteen registers to hold the four)
                        const uint4 f = fcw(0, 0);
                        facc = mfma16(f, x0h, facc);
                        asm volatile("" ::: "memory");
                    }
                    {
                        const uint4 f = fcw(1, 0);
                        facc = mfma16(f, x1h, facc);
                        asm volatile("" ::: "memory");
                    }
                    facc = mfma_corr(fcw(0, 1), fcw(1, 1), xc[0], xc[1], facc, sa_fc);
                }
            }
        }
        const int ln = lane_now(), n2 = ln & 31, hh2 = ln >> 5;
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) {
            // sum_r va_r tanh(k_r) = sum_r va_r - 2 sum_r va_r / (e^{2 k_r} + 1): one multiply, exp2, add, rcp and multiply-add per value
            float e = 0.f, sv = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = s_va[(wave * 2 + hh2) * 16 + r];
                sv += v;
                e += v * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(kacc[tt][r] * 2.8853900817779268f) + 1.0f);
            }
            e = sv - 2.0f * e;
            e += __shfl_xor(e, 32);
            if (hh2 == 0) s_epart[(wave * TG + tt) * 32 + n2] = e;
        }
        if (wave < TG && hh2 == 0) {      // C rows 0-3 of the fc1 tile = registers 0-3 of the lower lanes: (class 0, strand), (class 1, strand)
            s_pfc[((t0 + wave) * 32 + n2) * 2 + 0] = strand ? facc[1] : facc[0];
            s_pfc[((t0 + wave) * 32 + n2) * 2 + 1] = strand ? facc[3] : facc[2];
        }
        __syncthreads();   // all waves are done with the staging buffers before the next group restages them
        const int tid2 = wave * 64 + ln;
        if (tid2 < TG * 32) {       // the group's scores: the eight waves' partials in a fixed order (deterministic)
            const int tt = tid2 >> 5, rl = tid2 & 31;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) v += s_epart[(w * TG + tt) * 32 + rl];
            s_e[(t0 + tt) * 32 + rl] = v;
        }                                  // (s_epart is next written at the end of the next group, a chunk loop of barriers away)
    }
    __syncthreads();

    // ---- softmax over t and the strand-half of the logits (fixed summation order: deterministic)
    if (wave == 0 && lane_now() < 32) {
        const int rl = lane_now();
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
