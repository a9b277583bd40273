// libccsm GRU layers 1-2, ONE pass over x_t per step (split-mx arithmetic of ccsm_gru_mx.hip, same fragments, blobs and tail).
// Included by ccsm_api.hip after ccsm_gru_mx.hip.
//
// Why.  The GRU kernels run at the package power cap, so their time is their energy (profiles/r02_c_power_attribution.md); moving
// x_t into LDS costs 14 % of it and the two-pass kernel moved every x_t twice (phase A for the r and z gates, phase C for the n
// gate), and every wave read every x fragment from LDS twice.  Reading x once needs FOUR accumulator sets — R, Z, N_i (input
// part of the n gate) and N_h (its recurrent part: n = tanh(N_i + r * N_h)) — which do not fit beside the operands for 96 batch
// rows per workgroup (192 of 256 registers), but do for 64: 2 batch tiles x 4 sets x 16 = 128 accumulator registers.
//
// Workgroup = 64 batch rows of one direction, 8 waves, wave w owns hidden units [32w, 32w + 32) of all gates.  Per step:
//   X : R, Z, N_i += W_i{r,z,n} x_t      16 pairs of k-blocks, operands through the x ring (one barrier per pair)
//   H : R, Z, N_h += W_h{r,z,n} h_{t-1}   8 pairs, operands are the h fragments in LDS
//   tail (mx_tail): r, z = sigmoid; n = tanh(N_i + r N_h); h' = n + z (h - n); fragments + blobs to LDS and HBM
// x ring: SIX pair slots of 8 KiB (2 k-blocks x 2 batch tiles x [hi | corr]); wave w moves fragment w of a slot, so every wave
// issues exactly ONE transfer per pair.  The slot pair P vacates is refilled with pair P + 6 at the END of pair P, as the youngest
// vector-memory operation of the pair.  A wave's vector-memory operations retire in order: the weight requests behind a transfer
// cannot return before it has landed (a few thousand cycles from HBM), so
//   * weights are requested two pairs ahead in phase X (two register slots of 13 requests: 6 hi fragments, 3 fp6 blobs in two
//     pieces, 1 scale dword), one pair ahead in phase H (10 requests, fp4 blobs);
//   * the refills of pairs 13-15 are deferred to the end of phase H: issued in place they would sit in front of phase H's
//     one-pair-ahead weight requests.
// The barrier of pair P (mid-pair: main MFMAs done, correction MFMAs still to come) publishes pair P + 1: each wave first waits
// for its own transfer of that pair with a COUNTED s_waitcnt — kMx1Wait[P] = the number of vector-memory operations the wave has
// issued since that transfer (every load below is unconditional; derivation at the table).
//   xin : [tile][t][32 kb][hi | corr][64] uint4      out : the same (OUT_FP8: fp8 corr fragments for the attention kernel)
//   wst : per (direction, wave): 16 X pairs of kMxPairX bytes, then 8 H pairs of kMxPairB bytes
//     X pair : hi (kbl, g) at (3 kbl + g) KiB | fp6 blob (g): bytes 0-15 at (6 + g) KiB, bytes 16-23 (lane * 8) at 9 KiB + 512 g |
//              scale dwords (lane * 4, byte g = gate g) at 10.5 KiB
//     H pair : as in ccsm_gru_mx.hip (hi (kbl, g) at (3 kbl + g) KiB | fp4 blob (g) at (6 + g) KiB | scales at 9 KiB)
// LDS : h fragments 64 KiB | x ring 6 x 8 KiB | residuals 8 KiB | biases 4 KiB = 124 KiB
#include <hip/hip_runtime.h>

namespace ccsm {

constexpr int kMx1NB = 2;                                   // batch tiles (32 rows) per workgroup
constexpr int kMx1RS = 6;                                   // ring slots
constexpr int kMx1SlotBytes = 2 * kMx1NB * 2 * 1024;
constexpr int kMx1HBytes = kKBH * kMx1NB * 2 * 1024;
constexpr int kMx1XOff = kMx1HBytes, kMx1LoOff = kMx1XOff + kMx1RS * kMx1SlotBytes, kMx1BiasOff = kMx1LoOff + kWaves * kMx1NB * 64 * 8;
constexpr int kMx1Lds = kMx1BiasOff + kWaves * 4 * 32 * 4;
constexpr int kMxPairX = 10 * 1024 + 512 + 256;
constexpr int kMx1OffH = (kKB12 / 2) * kMxPairX;
constexpr int kMx1WBytes = kMx1OffH + (kKBH / 2) * kMxPairB;

// Vector-memory operations of one wave, in program order, per step:
//   X pair P: [a] 3 (hi, first k-block of pair P + 2) [b] 3 (second k-block) | wait, barrier | [c] 7 (blobs, scales) [d] 1 transfer
//             P = 14 requests phase H's first pair instead (3, 3, 4), P = 15 nothing; P = 13, 14, 15 defer [d]
//   H pair Q: 10 requests (pair Q + 1; Q = 7: 13 requests = X slot 0 of the next step)
//   then the 3 deferred transfers (pairs 3, 4, 5 of the next step), 13 requests (X slot 1 of the next step), 8 stores (tail)
// The transfer of pair P + 1 is [d] of pair P - 5 (P >= 5), or of pairs 11, 12 of the step before (P = 0, 1), or one of the deferred
// ones (P = 2, 3, 4).  Operations issued since, up to the wait of pair P (the counter's ceiling is 63):
//   P = 5..13 : 4 pairs x 14 + 6                                  = 62
//   P = 14    : pairs 10-13: 4 x 13 + 3 transfers, + 6            = 61
//   P = 15    : pairs 11-14: 13 + 13 + 13 + 10 + 2 transfers, + 0 = 51
//   P = 0, 1  : the whole phase H lies in between                > 63
//   P = 2     : 2 transfers + 13 + 8 + pairs 0, 1 (28) + 6        = 57
//   P = 3, 4  : 1 | 0 transfers + 13 + 8 + 42 | 56 + 6           > 63
constexpr int kMx1Wait[16] = {63, 63, 57, 63, 63, 62, 62, 62, 62, 62, 62, 62, 62, 62, 61, 51};

template <bool OUT_FP8, bool DBG>
__global__ __launch_bounds__(512, 2) void gru_layer12_mx1_kernel(const uint4* __restrict__ xin, uint4* __restrict__ out,
                                                                  const uint4* __restrict__ wst, const float* __restrict__ bias,
                                                                  const float* __restrict__ h0, int rows_p,
                                                                  unsigned long long* __restrict__ dbg) {
    constexpr int NB = kMx1NB, KX = kKB12, NPAIR = KX / 2, RS = kMx1RS, SLOT_BYTES = kMx1SlotBytes;
    constexpr int PX = kMxPairX, PB = kMxPairB, OFF_H = kMx1OffH, X_OFF = kMx1XOff;
    static_assert(2 * NB * 2 == kWaves, "one ring fragment per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dir = blockIdx.x & 1;
    const int tile0 = (blockIdx.x >> 1) * NB;
    const int hh = lane >> 5;
    const int lane16 = lane * 16;
    const int sb = hh ? kMxScaleLo : kMxScaleHi;

    if (threadIdx.x < kWaves * 4 * 32 / 4)
        reinterpret_cast<float4*>(smem + kMx1BiasOff)[threadIdx.x] = reinterpret_cast<const float4*>(bias + (size_t)dir * kWaves * 4 * 32)[threadIdx.x];
    mx_h0_to_lds<NB>(smem, kMx1LoOff, h0 + (size_t)dir * rows_p * kHidden, tile0, wave, lane);

    // ---- x transfers: fragment f = (kbl * NB + bt) * 2 + hl of a ring slot; wave w moves fragment w
    const u32x4_t xrs = dma_rsrc(xin);
    const unsigned sx_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + X_OFF);
    auto dma_pair = [&](int slot, int sd, int jd) {                 // wave-uniform: ring slot, step (clamped), pair of x_t(sd)
        const int sc_ = sd < kSeqLen ? sd : kSeqLen - 1;
        const int td = dir ? kSeqLen - 1 - sc_ : sc_;
        const int f = wave;
        const int hl = f & 1, bt = (f >> 1) % NB, kbl = (f >> 1) / NB;
        const int soff = (((((tile0 + bt) * kSeqLen + td) * KX + (2 * jd + kbl)) * 2 + hl) << 10);
        dma16_buf(xrs, lane16, __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane((int)(sx_base + slot * SLOT_BYTES + (f << 10))));
    };
    auto dma_ahead = [&](int slot, int s, int p) {                  // pair p + RS (of the next step where that is past this one's last)
        const int g = p + RS;
        dma_pair(slot, s + (g >> 4), g & (NPAIR - 1));
    };

    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const char*>(wst) + (size_t)(dir * kWaves + wave) * kMx1WBytes);
    const int bias_off = kMx1BiasOff + wave * 4 * 32 * 4;
    auto w_at = [&](int off) -> uint4 { return buf_load(wrs, lane16, off); };
    auto ws_at = [&](int off) -> uint32_t { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, lane * 4, off, 0); };
    auto w8_at = [&](int off) -> uint2 {            // bytes 16-23 of an fp6 blob: lane * 8
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(wrs, lane * 8, off, 0);
        return make_uint2(v[0], v[1]);
    };

    // weight registers: phase X two pair slots [slot][kb in pair][gate] hi, [slot][gate] fp6 blobs (16 + 8 bytes), [slot] scale bytes;
    // phase H one resident pair (fp4 blobs)
    uint4 wxh[2][2][3], wxb[2][3];
    uint2 wxb1[2][3];
    uint32_t wxs[2];
    uint4 wbh[2][3], wbb[3];
    uint32_t wbs;
    auto ldXh = [&](uint4 (&d)[3], int p, int kbl) {
#pragma unroll
        for (int g = 0; g < 3; ++g) d[g] = w_at(p * PX + ((3 * kbl + g) << 10));
    };
    auto ldXb = [&](int ws, int p) {                // 7 requests
#pragma unroll
        for (int g = 0; g < 3; ++g) { wxb[ws][g] = w_at(p * PX + ((6 + g) << 10)); wxb1[ws][g] = w8_at(p * PX + (9 << 10) + 512 * g); }
        wxs[ws] = ws_at(p * PX + (10 << 10) + 512);
    };
    auto ldX_slot = [&](int ws, int p) { ldXh(wxh[ws][0], p, 0); ldXh(wxh[ws][1], p, 1); ldXb(ws, p); };     // 13 requests

    // ---- prologue: the ring's first pairs, the two weight slots
#pragma unroll
    for (int g = 0; g < RS; ++g) dma_pair(g, 0, g);
    ldX_slot(0, 0);
    ldX_slot(1, 1);
    asm volatile("s_waitcnt vmcnt(26)" ::: "memory");               // all ring transfers (older than the 26 weight requests)
    __syncthreads();                                                // ring, h0 fragments and biases are in LDS

    int slot = 0;                                                   // ring slot of the next pair (wave-uniform)
    for (int s = 0; s < kSeqLen; ++s) {
        const int t = dir ? (kSeqLen - 1 - s) : s;
        auto stamp = [&](int k) {
            if constexpr (DBG) {
                if (dbg != nullptr && blockIdx.x == 0 && lane == 0) dbg[(s * kWaves + wave) * 5 + k] = __builtin_readcyclecounter();
            }
        };
        stamp(0);
        f32x16 acc[4][NB];                                          // R, Z, N_i, N_h
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
            const f32x16 b0 = bias_set(0), b1 = bias_set(1), b2 = bias_set(2);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) { acc[0][bt] = b0; acc[1][bt] = b1; acc[2][bt] = b2; }
        }

        uint4 xh[NB], xh1[NB], xc0[NB];
        uint2 xc1[NB];
        auto rdx = [&](uint4 (&x)[NB], int xs, int kbl, int f) {    // xs = byte offset of the slot + lane * 16
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) x[bt] = *reinterpret_cast<const uint4*>(smem + xs + (((kbl * NB + bt) * 2 + f) << 10));
        };
        auto rdx_blob = [&](int xs) {
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                xc0[bt] = *reinterpret_cast<const uint4*>(smem + xs + (((0 * NB + bt) * 2 + 1) << 10));
                xc1[bt] = *reinterpret_cast<const uint2*>(smem + xs + (((1 * NB + bt) * 2 + 1) << 10));
            }
        };
        auto slot_off = [&](int sl) -> int { return X_OFF + sl * SLOT_BYTES + lane16; };
#define CCSM_FENCE asm volatile("" ::: "memory")
        // main product of one k-block: three gates into the accumulator sets S0, S1, S2
#define CCSM_MAIN3(W, X, S0, S1, S2)                                                                          \
    do {                                                                                                      \
        CCSM_FENCE;                                                                                           \
        _Pragma("unroll") for (int bt = 0; bt < NB; ++bt) {                                                   \
            acc[S0][bt] = mfma16(W[0], X[bt], acc[S0][bt]);                                                   \
            acc[S1][bt] = mfma16(W[1], X[bt], acc[S1][bt]);                                                   \
            acc[S2][bt] = mfma16(W[2], X[bt], acc[S2][bt]);                                                   \
        }                                                                                                     \
        CCSM_FENCE;                                                                                           \
    } while (0)

        // ---------------- phase X: R, Z, N_i += W_i{r,z,n} x_t, pairs 0..15 ---------------------------------------------------
        // pair P lives in weight slot P % 2; behind its MFMA groups the slot is refilled with pair P + 2 (3 + 3 + 7 requests); pair 14
        // takes phase H's first pair instead (3 + 3 + 4), pair 15 has nothing left to request
        rdx(xh, slot_off(slot), 0, 0);
        int slot_def[3] = {0, 0, 0};
        static_for<0, NPAIR>([&](auto PC_) {
            constexpr int P = decltype(PC_)::value;
            constexpr int WS = P % 2;
            const int xs = slot_off(slot);
            const int slot_n = slot == RS - 1 ? 0 : slot + 1;
            rdx(xh1, xs, 1, 0);
            CCSM_MAIN3(wxh[WS][0], xh, 0, 1, 2);
            if constexpr (P + 2 < NPAIR) ldXh(wxh[WS][0], P + 2, 0);
            else if constexpr (P == 14) {
#pragma unroll
                for (int g = 0; g < 3; ++g) wbh[0][g] = w_at(OFF_H + (g << 10));
            }
            rdx_blob(xs);
            CCSM_MAIN3(wxh[WS][1], xh1, 0, 1, 2);
            if constexpr (P + 2 < NPAIR) ldXh(wxh[WS][1], P + 2, 1);
            else if constexpr (P == 14) {
#pragma unroll
                for (int g = 0; g < 3; ++g) wbh[1][g] = w_at(OFF_H + ((3 + g) << 10));
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kMx1Wait[P]) : "memory");   // this wave's fragment of the next pair has landed
            __syncthreads();             // the next pair is in LDS; every wave has read this pair's operands
            if constexpr (P + 1 < NPAIR) rdx(xh, slot_off(slot_n), 0, 0);
            CCSM_FENCE;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                acc[0][bt] = mfma_corr_mx6<0>(wxb[WS][0], wxb1[WS][0], wxs[WS], xc0[bt], xc1[bt], acc[0][bt], sb);
                acc[1][bt] = mfma_corr_mx6<1>(wxb[WS][1], wxb1[WS][1], wxs[WS], xc0[bt], xc1[bt], acc[1][bt], sb);
                acc[2][bt] = mfma_corr_mx6<2>(wxb[WS][2], wxb1[WS][2], wxs[WS], xc0[bt], xc1[bt], acc[2][bt], sb);
            }
            CCSM_FENCE;
            if constexpr (P + 2 < NPAIR) ldXb(WS, P + 2);
            else if constexpr (P == 14) {
#pragma unroll
                for (int g = 0; g < 3; ++g) wbb[g] = w_at(OFF_H + ((6 + g) << 10));
                wbs = ws_at(OFF_H + (9 << 10));
            }
            CCSM_FENCE;
            if constexpr (P < NPAIR - 3) dma_ahead(slot, s, P); else slot_def[P - (NPAIR - 3)] = slot;
            slot = slot_n;
        });

        stamp(1);
        // ---------------- phase H: R, Z, N_h += W_h{r,z,n} h_{t-1}  (N_h starts at b_hn) -------------------------------------
        {
            const f32x16 b3 = bias_set(3);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) acc[3][bt] = b3;
        }
        const int sbh = sb + (s == 0 ? kMxScaleHi0 - kMxScaleHi : 0);
        // pair Q = k-blocks 2Q, 2Q + 1: one pair resident, refilled with the next pair behind each MFMA group (3 + 3 + 4 requests);
        // the last pair's positions take the next step's X slot 0 (3 + 3 + 7)
        static_for<0, kKBH / 2>([&](auto QC) {
            constexpr int Q = decltype(QC)::value;
            constexpr bool LAST = Q == kKBH / 2 - 1;
            constexpr int NXT = OFF_H + (Q + 1) * PB;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                xh[bt] = *reinterpret_cast<const uint4*>(smem + mx_hfrag<NB>(2 * Q, bt, 0) + lane * 16);
                xc0[bt] = *reinterpret_cast<const uint4*>(smem + mx_hfrag<NB>(2 * Q, bt, 1) + lane * 16);
                xc1[bt] = *reinterpret_cast<const uint2*>(smem + mx_hfrag<NB>(2 * Q + 1, bt, 1) + lane * 16);
            }
            CCSM_MAIN3(wbh[0], xh, 0, 1, 3);
            if constexpr (!LAST) {
#pragma unroll
                for (int g = 0; g < 3; ++g) wbh[0][g] = w_at(NXT + (g << 10));
            } else {
                ldXh(wxh[0][0], 0, 0);
            }
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) xh[bt] = *reinterpret_cast<const uint4*>(smem + mx_hfrag<NB>(2 * Q + 1, bt, 0) + lane * 16);
            CCSM_MAIN3(wbh[1], xh, 0, 1, 3);
            if constexpr (!LAST) {
#pragma unroll
                for (int g = 0; g < 3; ++g) wbh[1][g] = w_at(NXT + ((3 + g) << 10));
            } else {
                ldXh(wxh[0][1], 0, 1);
            }
            CCSM_FENCE;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                acc[0][bt] = mfma_corr_mx<0>(wbb[0], wbs, xc0[bt], xc1[bt], acc[0][bt], sbh);
                acc[1][bt] = mfma_corr_mx<1>(wbb[1], wbs, xc0[bt], xc1[bt], acc[1][bt], sbh);
                acc[3][bt] = mfma_corr_mx<2>(wbb[2], wbs, xc0[bt], xc1[bt], acc[3][bt], sbh);
            }
            CCSM_FENCE;
            if constexpr (!LAST) {
#pragma unroll
                for (int g = 0; g < 3; ++g) wbb[g] = w_at(NXT + ((6 + g) << 10));
                wbs = ws_at(NXT + (9 << 10));
            } else {
                ldXb(0, 0);
            }
            CCSM_FENCE;
        });
        // the deferred ring refills: pairs 3, 4, 5 of the next step
        dma_ahead(slot_def[0], s, NPAIR - 3);
        dma_ahead(slot_def[1], s, NPAIR - 2);
        dma_ahead(slot_def[2], s, NPAIR - 1);
        stamp(2);
        // n's argument: N_i + sigmoid(R) * N_h; z = sigmoid(Z)
#pragma unroll
        for (int bt = 0; bt < NB; ++bt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[2][bt][r] = acc[2][bt][r] + sigmoid_f(acc[0][bt][r]) * acc[3][bt][r];
        CCSM_FENCE;
        ldX_slot(1, 1);                                             // the next step's second weight slot, once R and N_h are dead
        CCSM_FENCE;
#pragma unroll
        for (int bt = 0; bt < NB; ++bt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[1][bt][r] = sigmoid_f(acc[1][bt][r]);
        stamp(3);
        __syncthreads();                                            // every wave has read h_{t-1} (phase H) before anybody overwrites its fragments
        mx_tail<OUT_FP8, NB>(smem, kMx1LoOff, acc[1], acc[2], out, tile0, t, dir, wave, lane16_here());
        CCSM_FENCE;
        stamp(4);
#undef CCSM_MAIN3
#undef CCSM_FENCE
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // no transfer may still be writing LDS when the workgroup retires
}

}  // namespace ccsm
