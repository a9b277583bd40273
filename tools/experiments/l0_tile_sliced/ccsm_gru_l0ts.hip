// libccsm GRU layer 0, "tile-sliced": the same arithmetic, operand formats, weight stream and LDS state layout as gru_layer0_mx_kernel
// (ccsm_gru_mx.hip), on a different schedule.  Included by ccsm_api.hip after ccsm_gru_mx.hip.
//
// Why.  Layer 0's input part is one k-block (K = 11), so a timestep of the k-outer kernel is 8 pairs of recurrent MFMAs followed by the
// step tail - gate activations, blend, fragment / blob packing for 96 rows: VALU only - and the cycle stamps (tools/gpu_phases.py) show
// the tail at 10-13 k of the step's 31 k cycles: both waves of a SIMD sit in it at the same time while the MFMA pipe idles (in layers
// 1-2 the same tail is 5-9 k of 73 k).  Rows are independent, so here a step is cut into three SLOTS, one per 32-row batch tile: slot
// i runs ALL of tile (i mod 3)'s MFMAs of step i / 3 (input part: 9, recurrent part: 8 pairs) into one accumulator set while the gate
// math and the packing of slot i - 1's tile run in the same instruction stream from the other set, i.e. in the shadow of the MFMAs.
// One barrier per slot publishes the h fragments the previous slot's tile wrote (they are read again three slots later).  The price:
// the recurrent weights (74 KiB per wave) stream three times per step instead of once - from L2, where a direction's 592 KiB stay.
//
// Every load is an ordinary buffer / global load with compiler-counted waits (no LDS-DMA: the input of a slot is two 16-byte
// fragments per lane, read straight into the B operand), so the only hand-placed synchronisation is the slot barrier, which waits for
// LDS traffic only (s_waitcnt lgkmcnt(0)): the weight prefetch stays in flight across it.
//   xin : [tile][t][hi|lo][64] uint4          out : [tile][t][32 kb][hi | corr][64] uint4 (activation blobs in the corr fragments)
// LDS : h fragments 96 KiB | (unused 12 KiB) | residuals 12 KiB | biases 4 KiB  (the offsets of gru_layer0_mx_kernel)
#include <hip/hip_runtime.h>

namespace ccsm {

// h_{t-1} of this wave's own 32 units of batch tile bt, MFMA C layout (what mx_tail reads inline)
template <bool HS3>
__device__ __forceinline__ void ts_prev_state(const char* smem, int lo_off, int bt, int wave, int t16, float (&hp)[16]) {
    const int own_off = wave * (2 * kMxNB * 2 * 1024);
    const char* t_wr = smem + (own_off + t16);
    const char* t_rd = smem + (own_off + (t16 & 0x1f0) + ((t16 >> 9) << 3));
    const char* t_lo = smem + (lo_off + wave * (kMxNB * 64 * 8) + (t16 >> 1));
    auto own_frag = [&](int kbl, int f) -> int { return ((kbl * kMxNB + bt) * 2 + f) << 10; };
    uint2 la = make_uint2(0, 0), lb = make_uint2(0, 0);
    if constexpr (!HS3) {
        la = *reinterpret_cast<const uint2*>(t_wr + own_frag(1, 1) + 8);
        lb = *reinterpret_cast<const uint2*>(t_lo + bt * (64 * 8));
    }
    const uint32_t l8[4] = {la.x, la.y, lb.x, lb.y};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const half4 hi = as_half4(*reinterpret_cast<const uint2*>(t_rd + own_frag(q >> 1, 0) + 512 * (q & 1)));
        if constexpr (HS3) {
            const half4 lo = as_half4(*reinterpret_cast<const uint2*>(t_rd + own_frag(q >> 1, 1) + 512 * (q & 1)));
#pragma unroll
            for (int e = 0; e < 4; ++e) hp[4 * q + e] = (float)hi[e] + (float)lo[e];
        } else {
            const int lo4 = (int)l8[q];
            hp[4 * q + 0] = (float)hi[0] + __builtin_amdgcn_cvt_f32_fp8(lo4, 0) * (1.0f / kMxLoScale);
            hp[4 * q + 1] = (float)hi[1] + __builtin_amdgcn_cvt_f32_fp8(lo4, 1) * (1.0f / kMxLoScale);
            hp[4 * q + 2] = (float)hi[2] + __builtin_amdgcn_cvt_f32_fp8(lo4, 2) * (1.0f / kMxLoScale);
            hp[4 * q + 3] = (float)hi[3] + __builtin_amdgcn_cvt_f32_fp8(lo4, 3) * (1.0f / kMxLoScale);
        }
    }
}

// h_t of the same units: fragments / blobs for the next step (LDS) and the next layer (HBM) - mx_tail's stores for one tile
template <bool HS3>
__device__ __forceinline__ void ts_store(char* smem, int lo_off, int bt, const float (&hn)[16], uint4* __restrict__ out, int tile0, int t, int dir,
                                         int wave, int t16) {
    const int own_off = wave * (2 * kMxNB * 2 * 1024);
    char* t_wr = smem + (own_off + t16);
    char* t_lo = smem + (lo_off + wave * (kMxNB * 64 * 8) + (t16 >> 1));
    auto own_frag = [&](int kbl, int f) -> int { return ((kbl * kMxNB + bt) * 2 + f) << 10; };
    uint4 hi0, hi1, c0, lo8;
    uint2 c1;
    pack_pair_mx<false>(hn, 0.25f, hi0, hi1, c0, c1, lo8);
    const uint4 c1w = make_uint4(c1.x, c1.y, lo8.x, lo8.y);
    *reinterpret_cast<uint4*>(t_wr + own_frag(0, 0)) = hi0;
    *reinterpret_cast<uint4*>(t_wr + own_frag(1, 0)) = hi1;
    if constexpr (HS3) {
        uint4 h0_, h1_, lo0, lo1;
        pack_pair_hl(hn, h0_, h1_, lo0, lo1);
        *reinterpret_cast<uint4*>(t_wr + own_frag(0, 1)) = lo0;
        *reinterpret_cast<uint4*>(t_wr + own_frag(1, 1)) = lo1;
    } else {
        *reinterpret_cast<uint4*>(t_wr + own_frag(0, 1)) = c0;
        *reinterpret_cast<uint4*>(t_wr + own_frag(1, 1)) = c1w;
        *reinterpret_cast<uint2*>(t_lo + bt * (64 * 8)) = make_uint2(lo8.z, lo8.w);
    }
    char* o = reinterpret_cast<char*>(out + (((size_t)(tile0 + bt) * kSeqLen + t) * kKB12 + (dir * kKBH + 2 * wave)) * 2 * kFragU4) + t16;
    nt_store(hi0, reinterpret_cast<uint4*>(o));
    nt_store(hi1, reinterpret_cast<uint4*>(o + 2048));
    nt_store(c0, reinterpret_cast<uint4*>(o + 1024));
    nt_store(c1w, reinterpret_cast<uint4*>(o + 3072));
}

#define CCSM_TS_FENCE asm volatile("" ::: "memory")
struct TsAcc { f32x16 r, z, nx, nh; };

template <bool HS3>
__global__ __launch_bounds__(512, 2) void gru_layer0_ts_kernel(const uint4* __restrict__ xin, uint4* __restrict__ out,
                                                                const uint4* __restrict__ wst, const float* __restrict__ bias,
                                                                const float* __restrict__ h0, int rows_p) {
    constexpr int NB = kMxNB;
    constexpr int PB = mx_pair_b(HS3);
    constexpr int OFF_B = 4 * 1024, OFF_C = OFF_B + (kKBH / 2) * PB;
    constexpr int NSLOT = kSeqLen * NB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dir = blockIdx.x & 1;
    const int tile0 = (blockIdx.x >> 1) * NB;
    const int hh = lane >> 5;
    const int lane16 = lane * 16;
    const int sb = hh ? kMxScaleLo : kMxScaleHi;

    if (threadIdx.x < kWaves * 4 * 32 / 4)
        reinterpret_cast<float4*>(smem + kMx0BiasOff)[threadIdx.x] = reinterpret_cast<const float4*>(bias + (size_t)dir * kWaves * 4 * 32)[threadIdx.x];
    mx_h0_to_lds<HS3>(smem, kMx0LoOff, h0 + (size_t)dir * rows_p * kHidden, tile0, wave, lane);

    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const char*>(wst) + (size_t)(dir * kWaves + wave) * mx0_wbytes(HS3));
    const int bias_off = kMx0BiasOff + wave * 4 * 32 * 4;
    auto w_at = [&](int off) -> uint4 { return buf_load(wrs, lane16, off); };
    auto ws_at = [&](int off) -> uint32_t { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, lane * 4, off, 0); };

    // recurrent weights: ONE pair resident, each position refilled with the next pair's (the next slot's first pair after the last) right
    // behind its use
    uint4 wbh[2][3], wbb[3], wbl[2][3];
    uint32_t wbs = 0;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        wbh[0][g] = w_at(OFF_B + (g << 10)); wbh[1][g] = w_at(OFF_B + ((3 + g) << 10));
        if constexpr (HS3) { wbl[0][g] = w_at(OFF_B + ((6 + g) << 10)); wbl[1][g] = w_at(OFF_B + ((9 + g) << 10)); }
        else wbb[g] = w_at(OFF_B + ((6 + g) << 10));
    }
    if constexpr (!HS3) wbs = ws_at(OFF_B + (9 << 10));
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // biases and h0 fragments in LDS

    auto bias_set = [&](int set) {
        f32x16 b;
        const char* bp = smem + (bias_off + (hh << 6));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(bp + set * 128 + q * 16);
            b[4 * q + 0] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
        }
        return b;
    };
    auto slot_t = [&](int i) { const int s = i / NB; return dir ? (kSeqLen - 1 - s) : s; };

    // One slot: (MF) all MFMAs of slot i's tile into `cur`, (PO) gate math, blend and stores of slot i - 1's tile from `prv`, interleaved
    // by hand at pair granularity (two or three elements' activations behind each pair's MFMAs, the packing behind the input part) and by
    // the compiler below that: there is no fence inside a slot.
    auto slot = [&](auto MFC, auto POC, int i, TsAcc& cur, TsAcc& prv) {
        constexpr bool MF = decltype(MFC)::value, PO = decltype(POC)::value;
        const int bt = i % NB, s = i / NB, t = slot_t(i);
        const int ip = i - 1;
        const int btp = ip % NB, tp = PO ? slot_t(ip) : 0;
        float hp[16], hn[16];
        uint4 x0[2];
        if constexpr (MF) {
            const uint4* xp = xin + ((size_t)((tile0 + bt) * kSeqLen + t) * 2) * kFragU4 + lane;
            x0[0] = xp[0];
            x0[1] = xp[kFragU4];
            cur.r = bias_set(0); cur.z = bias_set(1); cur.nx = bias_set(2); cur.nh = bias_set(3);
        }
        if constexpr (PO) ts_prev_state<HS3>(smem, kMx0LoOff, btp, wave, lane16, hp);
        auto gate_math = [&](int e) {
            const float r = sigmoid_f(prv.r[e]);
            const float nn = tanh_fold(prv.nx[e] + r * prv.nh[e]);
            const float z = sigmoid_f(prv.z[e]);
            hn[e] = (hp[e] - nn) * z + nn;
        };
        const int sbh = sb + (s == 0 ? kMxScaleHi0 - kMxScaleHi : 0);
        uint4 wx[3][2];
        CCSM_TS_FENCE;
        static_for<0, kKBH / 2>([&](auto QC) {
            constexpr int Q = decltype(QC)::value;
            constexpr int NXT = OFF_B + ((Q + 1) % (kKBH / 2)) * PB;
            if constexpr (MF) {
                const char* hb = smem + lane16;
                if constexpr (HS3) {
#pragma unroll
                    for (int kbl = 0; kbl < 2; ++kbl) {
                        const uint4 xh = *reinterpret_cast<const uint4*>(hb + mx_hfrag(2 * Q + kbl, bt, 0));
                        const uint4 xl = *reinterpret_cast<const uint4*>(hb + mx_hfrag(2 * Q + kbl, bt, 1));
                        cur.r = mfma16(wbh[kbl][0], xh, cur.r); cur.z = mfma16(wbh[kbl][1], xh, cur.z); cur.nh = mfma16(wbh[kbl][2], xh, cur.nh);
                        cur.r = mfma16(wbl[kbl][0], xh, cur.r); cur.z = mfma16(wbl[kbl][1], xh, cur.z); cur.nh = mfma16(wbl[kbl][2], xh, cur.nh);
                        cur.r = mfma16(wbh[kbl][0], xl, cur.r); cur.z = mfma16(wbh[kbl][1], xl, cur.z); cur.nh = mfma16(wbh[kbl][2], xl, cur.nh);
#pragma unroll
                        for (int g = 0; g < 3; ++g) { wbh[kbl][g] = w_at(NXT + ((3 * kbl + g) << 10)); wbl[kbl][g] = w_at(NXT + ((6 + 3 * kbl + g) << 10)); }
                    }
                } else {
                    const uint4 xh0 = *reinterpret_cast<const uint4*>(hb + mx_hfrag(2 * Q, bt, 0));
                    const uint4 xc0 = *reinterpret_cast<const uint4*>(hb + mx_hfrag(2 * Q, bt, 1));
                    const uint4 xh1 = *reinterpret_cast<const uint4*>(hb + mx_hfrag(2 * Q + 1, bt, 0));
                    const uint2 xc1 = *reinterpret_cast<const uint2*>(hb + mx_hfrag(2 * Q + 1, bt, 1));
                    cur.r = mfma16(wbh[0][0], xh0, cur.r); cur.z = mfma16(wbh[0][1], xh0, cur.z); cur.nh = mfma16(wbh[0][2], xh0, cur.nh);
                    cur.r = mfma16(wbh[1][0], xh1, cur.r); cur.z = mfma16(wbh[1][1], xh1, cur.z); cur.nh = mfma16(wbh[1][2], xh1, cur.nh);
                    cur.r = mfma_corr_mx<0>(wbb[0], wbs, xc0, xc1, cur.r, sbh);
                    cur.z = mfma_corr_mx<1>(wbb[1], wbs, xc0, xc1, cur.z, sbh);
                    cur.nh = mfma_corr_mx<2>(wbb[2], wbs, xc0, xc1, cur.nh, sbh);
#pragma unroll
                    for (int g = 0; g < 3; ++g) { wbh[0][g] = w_at(NXT + (g << 10)); wbh[1][g] = w_at(NXT + ((3 + g) << 10)); wbb[g] = w_at(NXT + ((6 + g) << 10)); }
                    wbs = ws_at(NXT + (9 << 10));
                }
                if constexpr (Q == 4) {         // the input part's weights (constant over the slots; not worth 24 registers all the time)
#pragma unroll
                    for (int g = 0; g < 2; ++g) { wx[g][0] = w_at((2 * g) << 10); wx[g][1] = w_at((2 * g + 1) << 10); }
                    wx[2][0] = w_at(OFF_C); wx[2][1] = w_at(OFF_C + 1024);
                }
            }
            if constexpr (PO) { gate_math(2 * Q); gate_math(2 * Q + 1); }
            CCSM_TS_FENCE;                         // a pair is the scheduling unit: its MFMAs, its refills and two elements' gate math
        });
        if constexpr (MF) {                     // input part: three fp16 passes on [hi | lo] fragments (raw z-scores reach hundreds)
            cur.r = mfma16(wx[0][0], x0[0], cur.r); cur.z = mfma16(wx[1][0], x0[0], cur.z); cur.nx = mfma16(wx[2][0], x0[0], cur.nx);
            cur.r = mfma16(wx[0][0], x0[1], cur.r); cur.z = mfma16(wx[1][0], x0[1], cur.z); cur.nx = mfma16(wx[2][0], x0[1], cur.nx);
            cur.r = mfma16(wx[0][1], x0[0], cur.r); cur.z = mfma16(wx[1][1], x0[0], cur.z); cur.nx = mfma16(wx[2][1], x0[0], cur.nx);
        }
        CCSM_TS_FENCE;
        if constexpr (PO) ts_store<HS3>(smem, kMx0LoOff, btp, hn, out, tile0, tp, dir, wave, lane16);
        // the tile written above is read again three slots later; the tile read above is overwritten in the next slot
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    TsAcc a, b;
    const std::true_type yes{};
    const std::false_type no{};
    slot(yes, no, 0, a, b);
    for (int i = 1; i + 1 < NSLOT; i += 2) {
        slot(yes, yes, i, b, a);
        slot(yes, yes, i + 1, a, b);
    }
    static_assert((NSLOT & 1) == 1, "the slot loop ends with the accumulators in `a`");
    slot(no, yes, NSLOT, b, a);
}

}  // namespace ccsm
