// libccsm GRU layers 1-2 in the THREE-PASS split-fp16 arithmetic (CCSM_PRECISION_SPLIT3) on the schedule of the split-mx kernels.
// Included by ccsm_api.hip after ccsm_gru_mx.hip (whose helpers, LDS layout of the state and layer-0 kernel it shares).
//
// Why it exists (round 4): `precision 0` serves every TRAINED checkpoint in this arithmetic (the block-scaled ones are heavy-tailed
// there: profiles/r04_a_tail_study.log), and until now it ran on the round-1 kernel (gru_layer_v2_kernel: x staged by plain global loads
// in chunks of four k-blocks, 2.90 ms per 12288-site launch against 1.83 ms for split-mx).  The phase stamps of the split-mx kernel
// (profiles/r02_i_phases_product_kernels.log) say where that kernel loses its time: its phase B (recurrent part, weights only) runs at
// the MFMA rate, its phases A and C (input part: weights from L2 AND x_t from HBM through the CU's one vector-memory path, 55-74 B/clk
// wanted of 64) at 59 % and 37 % of it.  With three fp16 passes per product the same bytes feed twice the MFMAs, so the phases that are
// memory-path-bound in split-mx are MFMA-bound here: this kernel is that schedule - LDS-DMA ring for x_t (five slots here), one barrier per
// pair of k-blocks, counted s_waitcnt, weights straight from L2 into registers one to two pairs ahead - with
//     W x = W_hi x_hi + W_lo x_hi + W_hi x_lo        (all on v_mfma_f32_32x32x16_f16, fp32 accumulation; the state as fp16 hi + lo)
// per k-block, the third product of a pair issued behind the pair's barrier (where split-mx has its correction product).
//
//   xin / out : [tile][t][32 kb][hi | lo][64] uint4 (what gru_layer0_mx_kernel<.., F3> writes and attn_fc_kernel reads)
//   wst       : per (direction, wave) one stream in consumption order, 1 KiB fragments (lane * 16):
//                 phase-A pair (r, z) : hi (kbl, g) at (2 kbl + g) KiB | lo (kbl, g) at (4 + 2 kbl + g) KiB                = 8 KiB  x 16
//                 phase-B pair (r,z,n): hi (kbl, g) at (3 kbl + g) KiB | lo (kbl, g) at (6 + 3 kbl + g) KiB  (= the hybrid) = 12 KiB x 8
//                 phase-C pair (n)    : hi (kbl) at kbl KiB | lo (kbl) at (2 + kbl) KiB, pairs in zig-zag order            = 4 KiB  x 16
//   LDS       : h fragments [kb 16][bt NB][hi | lo] 32 NB KiB | x ring RS x [kbl 2][bt NB][hi | lo] 4 NB KiB | biases 4 KiB  = 160 KiB at NB = 3, RS = 5
// Vector-memory operations of a wave per pair (they set the counted waits; a wave's operations retire in order):
//   phase A: 4 requests before the pair's barrier (the lo fragments of the slot's next pair), d transfer instructions and 4 requests
//            (its hi fragments) behind it; phase C: 2 + d + 2; d = 2 for the waves that move two fragments of a ring slot, 1 for the others.
#include <hip/hip_runtime.h>

namespace ccsm {

constexpr int kF3PairA = 8 * 1024, kF3PairB = 12 * 1024, kF3PairC = 4 * 1024;
constexpr int kF3OffB = (kKB12 / 2) * kF3PairA;
constexpr int kF3OffC = kF3OffB + (kKBH / 2) * kF3PairB;
constexpr int kF3WBytes = kF3OffC + (kKB12 / 2) * kF3PairC;          // 288 KiB per (direction, wave)
#ifndef CCSM_F3_RING
#define CCSM_F3_RING 5               // ring slots of x_t: the kernel's LDS has room for five at 96 rows (96 + 5 x 12 + 4 = 160 KiB); -DCCSM_F3_RING=4: A/B
#endif
constexpr int kF3RS = CCSM_F3_RING;
constexpr int f3_xoff(int nb) { return mx_hbytes(nb); }
constexpr int f3_biasoff(int nb) { return f3_xoff(nb) + kF3RS * mx_slot_bytes(nb); }
constexpr int f3_lds(int nb) { return f3_biasoff(nb) + kWaves * 4 * 32 * 4; }

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#define CCSM_FENCE asm volatile("" ::: "memory")

// DBG (only instantiated in a -DCCSM_PHASE_STAMPS build): workgroup 0 records the cycle counter at step start / behind phase A / B / C / the tail
template <int NB_ = kMxNB, bool DBG = false>
__global__ __launch_bounds__(512, 2) void gru_layer12_f3_kernel(const uint4* __restrict__ xin, uint4* __restrict__ out, const uint4* __restrict__ wst,
                                                                 const float* __restrict__ bias, const float* __restrict__ h0, int rows_p,
                                                                 unsigned long long* __restrict__ dbg = nullptr) {
    constexpr int NB = NB_, KX = kKB12, NPAIR = KX / 2, RS = kF3RS, SLOT_BYTES = mx_slot_bytes(NB);
    constexpr int X_OFF = f3_xoff(NB), BIAS_OFF = f3_biasoff(NB);
    constexpr int PA = kF3PairA, PB = kF3PairB, PC = kF3PairC, OFF_B = kF3OffB, OFF_C = kF3OffC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dir = blockIdx.x & 1;
    const int tile0 = (blockIdx.x >> 1) * NB;
    const int lane16 = lane * 16;
    start_stagger((blockIdx.x >> 1) & 3);

    if (threadIdx.x < kWaves * 4 * 32 / 4)
        reinterpret_cast<float4*>(smem + BIAS_OFF)[threadIdx.x] = reinterpret_cast<const float4*>(bias + (size_t)dir * kWaves * 4 * 32)[threadIdx.x];
    mx_h0_to_lds<true, false, NB>(smem, 0, h0 + (size_t)dir * rows_p * kHidden, tile0, wave, lane);

    // ---- x transfers: fragment f = (kbl * NB + bt) * 2 + hl of a ring slot; wave w moves fragment w, waves 0-3 also w + 8 (NB = 3)
    const u32x4_t xrs = dma_rsrc(xin + (size_t)tile0 * kSeqLen * KX * 2 * kFragU4);      // per-workgroup base: see gru_layer12_mx_kernel
    const unsigned sx_base = NB == 1 ? (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)X_OFF
                                     : (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + X_OFF);
    auto dma_pair = [&](int slot, int sd, int jd) {                 // wave-uniform: ring slot, step (clamped), pair of x_t(sd)
        const int sc_ = sd < kSeqLen ? sd : kSeqLen - 1;
        const int td = dir ? kSeqLen - 1 - sc_ : sc_;
        auto one = [&](int f) {
            const int hl = f & 1, bt = (f >> 1) % NB, kbl = (f >> 1) / NB;
            const int soff = ((((bt * kSeqLen + td) * KX + (2 * jd + kbl)) * 2 + hl) << 10);
            dma16_buf(xrs, lane16, __builtin_amdgcn_readfirstlane(soff),
                      __builtin_amdgcn_readfirstlane((int)(sx_base + slot * SLOT_BYTES + (f << 10))));
        };
        if (4 * NB >= kWaves || wave < 4 * NB) one(wave);
        if (4 * NB > kWaves && wave < 4 * NB - kWaves) one(wave + 8);
    };
    // consumption jj of a step: 0-15 phase-A pairs, 16-31 phase-C pairs (zig-zag); the transfer of consumption jj + RS goes into the slot
    // consumption jj just vacated
    auto dma_ahead = [&](int slot, int s, int jj) {
        const int g = jj + RS;
        const int c = g & (2 * NPAIR - 1);
        dma_pair(slot, s + (g >> 5), kMxZigZag && c >= NPAIR ? 2 * NPAIR - 1 - c : c & (NPAIR - 1));
    };
    // wait until this wave's part of a transfer has landed: at most BASE + K d younger operations, d = its transfer instructions per pair
    auto xfer_wait = [&](auto base_c, auto k_c) {
        constexpr int BASE = decltype(base_c)::value, K = decltype(k_c)::value;
        if (4 * NB > kWaves && wave < 4 * NB - kWaves) wait_vm<BASE + 2 * K>();
        else if (4 * NB >= kWaves || wave < 4 * NB) wait_vm<BASE + K>();
    };
#define CCSM_XW(BASE, K) xfer_wait(std::integral_constant<int, (BASE)>{}, std::integral_constant<int, (K)>{})

    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const char*>(wst) + (size_t)(dir * kWaves + wave) * kF3WBytes);
    const int bias_off = BIAS_OFF + wave * 4 * 32 * 4;
    auto w_at = [&](int off) -> uint4 { return buf_load(wrs, lane16, off); };

    // weight registers: phase A two pair slots [slot][kb in pair][gate r, z] hi / lo; phase B one resident pair [kb in pair][gate] hi / lo;
    // phase C four pair slots of the n gate [slot][kb in pair] hi / lo
    uint4 wah[2][2][2], wal[2][2][2];
    uint4 wbh[2][3], wbl[2][3];
    uint4 wch[4][2], wcl[4][2];
    auto a_hi = [&](int p, int kbl, int g) -> uint4 { return w_at(p * PA + ((2 * kbl + g) << 10)); };
    auto a_lo = [&](int p, int kbl, int g) -> uint4 { return w_at(p * PA + ((4 + 2 * kbl + g) << 10)); };
    auto c_hi = [&](int pp, int kbl) -> uint4 { return w_at(OFF_C + pp * PC + (kbl << 10)); };
    auto c_lo = [&](int pp, int kbl) -> uint4 { return w_at(OFF_C + pp * PC + ((2 + kbl) << 10)); };

    // ---- prologue: the ring's first pairs, the two phase-A weight slots
#pragma unroll
    for (int g = 0; g < RS; ++g) dma_pair(g, 0, g);
#pragma unroll
    for (int ws = 0; ws < 2; ++ws)
#pragma unroll
        for (int kbl = 0; kbl < 2; ++kbl)
#pragma unroll
            for (int g = 0; g < 2; ++g) { wal[ws][kbl][g] = a_lo(ws, kbl, g); wah[ws][kbl][g] = a_hi(ws, kbl, g); }
    wait_vm<16>();                                                  // all ring transfers (older than the 16 weight requests)
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
        uint4 xh[NB], xh1[NB], xl[NB], xl1[NB];
        auto rdx = [&](uint4 (&x)[NB], int xs, int kbl, int f) {    // xs = byte offset of the slot + lane * 16
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) x[bt] = *reinterpret_cast<const uint4*>(smem + xs + (((kbl * NB + bt) * 2 + f) << 10));
        };
        auto slot_off = [&](int sl) -> int { return X_OFF + sl * SLOT_BYTES + lane16; };

        // ---------------- phase A: R, Z += W_i{r,z} x_t, pairs 0..15; pair P lives in weight slot P & 1 ----------------------------
        // before the barrier: (W_hi + W_lo) x_hi of both k-blocks, the slot's lo fragments refilled with pair P + 2 behind each group;
        // behind it: W_hi x_lo of both k-blocks, then the hi fragments.  Pairs 14 and 15 request phase B's first pair instead.
        rdx(xh, slot_off(slot), 0, 0);
        int slot_a15 = 0;
        static_for<0, NPAIR>([&](auto PC_) {
            constexpr int P = decltype(PC_)::value;
            constexpr int WS = P & 1;
            const int xs = slot_off(slot);
            const int slot_n = slot == RS - 1 ? 0 : slot + 1;
            rdx(xh1, xs, 1, 0);
            CCSM_FENCE;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[g][bt] = mfma16(wah[WS][0][g], xh[bt], acc[g][bt]);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[g][bt] = mfma16(wal[WS][0][g], xh[bt], acc[g][bt]);
            CCSM_FENCE;
            if constexpr (P + 2 < NPAIR) { wal[WS][0][0] = a_lo(P + 2, 0, 0); wal[WS][0][1] = a_lo(P + 2, 0, 1); }
            else if constexpr (P == NPAIR - 2) { wbh[0][0] = w_at(OFF_B + (0 << 10)); wbh[0][1] = w_at(OFF_B + (1 << 10)); }
            else { wbh[1][2] = w_at(OFF_B + (5 << 10)); wbl[1][0] = w_at(OFF_B + (9 << 10)); }
            rdx(xl, xs, 0, 1);
            rdx(xl1, xs, 1, 1);
            CCSM_FENCE;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[g][bt] = mfma16(wah[WS][1][g], xh1[bt], acc[g][bt]);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[g][bt] = mfma16(wal[WS][1][g], xh1[bt], acc[g][bt]);
            CCSM_FENCE;
            if constexpr (P + 2 < NPAIR) { wal[WS][1][0] = a_lo(P + 2, 1, 0); wal[WS][1][1] = a_lo(P + 2, 1, 1); }
            else if constexpr (P == NPAIR - 2) { wbh[0][2] = w_at(OFF_B + (2 << 10)); wbl[0][0] = w_at(OFF_B + (6 << 10)); }
            else { wbl[1][1] = w_at(OFF_B + (10 << 10)); wbl[1][2] = w_at(OFF_B + (11 << 10)); }
            // this wave's part of the next pair's transfer (issued behind the barrier of pair P - (RS - 1)) has landed.  Operations since:
            // 4 behind that barrier, RS - 2 pairs of 4 + d + 4, 4 of this pair = 8 + (RS - 2)(8 + d).  The first RS - 1 pairs of a step look
            // back over the tail's 4 NB stores and phase C's lighter pairs (2 + d + 2): 6 + 4 (RS - 2) + 4 P + 4 NB + (RS - 2) d
            {
                constexpr int FULL = 8 + 8 * (RS - 2), EARLY = 6 + 4 * (RS - 2) + 4 * P + 4 * NB;
                CCSM_XW((P < RS - 1 && EARLY < FULL ? EARLY : FULL), RS - 2);
            }
            __syncthreads();             // the next pair is in LDS; every wave has read this pair's operands
            if constexpr (P + 1 < NPAIR) dma_ahead(slot, s, P); else slot_a15 = slot;   // the vacated slot is refilled at once (pair 15: behind phase B)
            if constexpr (P + 1 < NPAIR) rdx(xh, slot_off(slot_n), 0, 0);
            CCSM_FENCE;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[g][bt] = mfma16(wah[WS][0][g], xl[bt], acc[g][bt]);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[g][bt] = mfma16(wah[WS][1][g], xl1[bt], acc[g][bt]);
            CCSM_FENCE;
            if constexpr (P + 2 < NPAIR) {
                wah[WS][0][0] = a_hi(P + 2, 0, 0); wah[WS][0][1] = a_hi(P + 2, 0, 1); wah[WS][1][0] = a_hi(P + 2, 1, 0); wah[WS][1][1] = a_hi(P + 2, 1, 1);
            } else if constexpr (P == NPAIR - 2) {
                wbl[0][1] = w_at(OFF_B + (7 << 10)); wbl[0][2] = w_at(OFF_B + (8 << 10)); wbh[1][0] = w_at(OFF_B + (3 << 10)); wbh[1][1] = w_at(OFF_B + (4 << 10));
            }
            CCSM_FENCE;
            slot = slot_n;
        });

        stamp(1);
        // ---------------- phase B: R, Z, N += W_h{r,z,n} h_{t-1}  (N starts at b_hn): three passes per k-block on the fp16 hi + lo state;
        // one pair resident, each k-block's six fragments refilled with the next pair's right behind its MFMAs; the last pair's positions
        // take phase C's first two pair slots ----------------------------------------------------------------------------------------
        {
            const f32x16 b3 = bias_set(3);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) acc[2][bt] = b3;
        }
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
                for (int g = 0; g < 3; ++g) acc[g][bt] = mfma16(wbh[KBL][g], xh[bt], acc[g][bt]);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g][bt] = mfma16(wbl[KBL][g], xh[bt], acc[g][bt]);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g][bt] = mfma16(wbh[KBL][g], xl[bt], acc[g][bt]);
            CCSM_FENCE;
            if constexpr (Q + 1 < kKBH / 2) {
#pragma unroll
                for (int g = 0; g < 3; ++g) { wbh[KBL][g] = w_at(NXT + ((3 * KBL + g) << 10)); wbl[KBL][g] = w_at(NXT + ((6 + 3 * KBL + g) << 10)); }
            } else {
                wch[KBL][0] = c_hi(KBL, 0); wch[KBL][1] = c_hi(KBL, 1); wcl[KBL][0] = c_lo(KBL, 0); wcl[KBL][1] = c_lo(KBL, 1);
            }
            CCSM_FENCE;
        });
        // r = sigmoid(R) ; N = b_in + r * N
        {
            const f32x16 b2 = bias_set(2);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[2][bt][r] = b2[r] + sigmoid_f(acc[0][bt][r]) * acc[2][bt][r];
        }
        // phase C's pair slots 2 and 3 once R is dead, then the deferred ring refill (phase-C pair RS - 1): the waits of phase C's first
        // pairs count from it
        CCSM_FENCE;
#pragma unroll
        for (int q = 2; q < 4; ++q) { wch[q][0] = c_hi(q, 0); wch[q][1] = c_hi(q, 1); wcl[q][0] = c_lo(q, 0); wcl[q][1] = c_lo(q, 1); }
        CCSM_FENCE;
        dma_ahead(slot_a15, s, NPAIR - 1);
        CCSM_FENCE;
        auto zwork = [&](int bt) {                                  // z = sigmoid(Z) in place, inside phase C
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = sigmoid_f(acc[1][bt][r]);
                asm volatile("" : "+v"(v));                         // pins the evaluation HERE (the compiler otherwise sinks it to the tail)
                acc[1][bt][r] = v;
            }
        };

        stamp(2);
        // ---------------- phase C: N += W_in x_t, pairs 0..15 (consumptions 16..31, zig-zag); pair P lives in slot P & 3, refilled with
        // pair P + 4 (1 + 1 requests before the barrier, 2 behind it); pairs 12..15 request the next step's phase-A slots instead ----------
        rdx(xh, slot_off(slot), 0, 0);
        static_for<0, NPAIR>([&](auto PC_) {
            constexpr int P = decltype(PC_)::value;
            constexpr int WS = P & 3;
            constexpr int AS = P >= NPAIR - 4 ? (P - (NPAIR - 4)) / 2 : 0, AH = P >= NPAIR - 4 ? (P - (NPAIR - 4)) & 1 : 0;   // pairs 12..15: phase-A slot AS, AH = 0 lo / 1 hi
            const int xs = slot_off(slot);
            const int slot_n = slot == RS - 1 ? 0 : slot + 1;
            rdx(xh1, xs, 1, 0);
            CCSM_FENCE;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) acc[2][bt] = mfma16(wch[WS][0], xh[bt], acc[2][bt]);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) acc[2][bt] = mfma16(wcl[WS][0], xh[bt], acc[2][bt]);
            CCSM_FENCE;
            if constexpr (P + 4 < NPAIR) wcl[WS][0] = c_lo(P + 4, 0);
            else if constexpr (AH == 0) wal[AS][0][0] = a_lo(AS, 0, 0); else wah[AS][0][0] = a_hi(AS, 0, 0);
            rdx(xl, xs, 0, 1);
            rdx(xl1, xs, 1, 1);
            CCSM_FENCE;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) acc[2][bt] = mfma16(wch[WS][1], xh1[bt], acc[2][bt]);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) acc[2][bt] = mfma16(wcl[WS][1], xh1[bt], acc[2][bt]);
            CCSM_FENCE;
            if constexpr (P + 4 < NPAIR) wcl[WS][1] = c_lo(P + 4, 1);
            else if constexpr (AH == 0) wal[AS][0][1] = a_lo(AS, 0, 1); else wah[AS][0][1] = a_hi(AS, 0, 1);
            // operations since the awaited refill (behind the barrier of pair P - (RS - 1)): 2 behind it, RS - 2 pairs of 2 + d + 2, 2 of this
            // pair = 4 + (RS - 2)(4 + d); the first pairs count from the deferred refill behind phase B (d instructions, nothing else since):
            // pairs 0 .. RS - 3 must not wait for it (d + P (4 + d) + 2), pair RS - 2 waits for it ((RS - 2)(4 + d) + 2)
            if constexpr (P < RS - 2) CCSM_XW(2 + 4 * P, P + 1);
            else if constexpr (P == RS - 2) CCSM_XW(2 + 4 * (RS - 2), RS - 2);
            else CCSM_XW(4 + 4 * (RS - 2), RS - 2);
            __syncthreads();
            dma_ahead(slot, s, NPAIR + P);                              // the vacated slot is refilled at once
            if constexpr (P + 1 < NPAIR) rdx(xh, slot_off(slot_n), 0, 0);
            CCSM_FENCE;
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) acc[2][bt] = mfma16(wch[WS][0], xl[bt], acc[2][bt]);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) acc[2][bt] = mfma16(wch[WS][1], xl1[bt], acc[2][bt]);
            CCSM_FENCE;
            if constexpr (P + 4 < NPAIR) { wch[WS][0] = c_hi(P + 4, 0); wch[WS][1] = c_hi(P + 4, 1); }
            else if constexpr (AH == 0) { wal[AS][1][0] = a_lo(AS, 1, 0); wal[AS][1][1] = a_lo(AS, 1, 1); }
            else { wah[AS][1][0] = a_hi(AS, 1, 0); wah[AS][1][1] = a_hi(AS, 1, 1); }
            CCSM_FENCE;
            slot = slot_n;
            if constexpr (P == 1) zwork(0);
            if constexpr (P == 5 && NB > 1) zwork(1);
            if constexpr (P == 9 && NB > 2) zwork(2);
        });
        stamp(3);
        f3_tail<NB>(smem, acc[1], acc[2], out, tile0, t, dir, wave, lane16_here());
        CCSM_FENCE;
        stamp(4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // no transfer may still be writing LDS when the workgroup retires
#undef CCSM_XW
}
#undef CCSM_FENCE

}  // namespace ccsm
