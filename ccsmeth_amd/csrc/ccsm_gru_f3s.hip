// libccsm GRU layers 1-2 in the THREE-PASS split-fp16 arithmetic (CCSM_PRECISION_SPLIT3) on v_mfma_f32_16x16x32_f16.
// Included by ccsm_api.hip after ccsm_gru_f3.hip, whose schedule (x_t ring by LDS-DMA, one barrier per pair of k-blocks, counted waits,
// weights straight from L2 into registers, three phases over three accumulator sets), LDS layout, weight-stream SIZES and input / output
// formats it keeps byte for byte; what changes is the matrix instruction and therefore which values a weight fragment holds.
//
// Why (round 5, profiles/r05_m_mfma_power_shapes.log): every GRU kernel of this library runs at the package power cap, where a kernel's
// time is its energy.  On random register-resident fp16 operands the cap lets v_mfma_f32_32x32x16_f16 sustain 1715 TFLOP/s (sclk 1.68 GHz)
// and v_mfma_f32_16x16x32_f16 2015 TFLOP/s (2.03 GHz): the 16x16 shape reduces 32 k inside the array before it touches an accumulator -
// half the fp32 accumulate traffic per MAC for twice the operand reads - and that trade is worth +17.5 % at the cap.  split3 (three fp16
// passes per product: what every trained checkpoint is served with) is the arithmetic that sits on that ceiling (ccsm_gru_f3.hip's header).
//
// Operands.  x_t, the state and the layer output stay 32-row fragments of the 32x32x16 layout (lane n + 32 g <- row n, k = 16 kb + 8 g + j):
// a B operand of the 16x16x32 instruction (lane n' + 16 q <- row n', k = 8 q + j of 32) for rows [16 s, 16 s + 16) of a 32-row tile and the
// pair of k-blocks (2 P, 2 P + 1) is those fragments read with per-lane addresses -
//     lane (n', q) reads 16 B at  fragment(2 P + (q >> 1)) + (16 s + n' + 32 (q & 1)) * 16
// (every 16 lanes one contiguous 256 B: conflict-free) - so nothing about transfers, ring or producers changes.  The C tile of the
// instruction is 16 units x 16 rows (lane (n', q) <- row n', units 4 q + i): a wave's 32 hidden units are two unit tiles T per gate with
//     unit tile T, row 4 q + i  <->  hidden unit 32 wave + 8 q + 4 T + i
// (a permutation the host applies when it packs the weight fragments: pack_wstream_f3s), so that after the gate math lane (n', q) holds
// units 32 wave + 8 q + 0..7 of row n' - exactly its own 16 bytes of the 32-row output fragment: the step tail needs NO lane exchange
// (the 32x32 form: eight v_permlane32_swap per row tile).
//   wst : per (direction, wave) one stream in consumption order, 1 KiB fragments (lane (m, q) <- unit-tile row m, k = 32 pair + 8 q + j):
//           phase-A pair (r, z) : hi (T, g) at (2 T + g) KiB | lo (T, g) at (4 + 2 T + g) KiB          = 8 KiB  x 16
//           phase-B pair (r,z,n): hi (T, g) at (3 T + g) KiB | lo (T, g) at (6 + 3 T + g) KiB          = 12 KiB x 8
//           phase-C pair (n)    : hi (T) at T KiB | lo (T) at (2 + T) KiB, pairs in zig-zag order      = 4 KiB  x 16
//   bias: natural unit order [direction][wave][set r, z, n_x, n_h][32] (pack_bias_natural)
//   LDS, vector-memory operations per pair, counted waits: ccsm_gru_f3.hip.
#include <hip/hip_runtime.h>

namespace ccsm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// The accumulate chain as an asm statement with the accumulator TIED (D = C): through the builtin the register allocator treats the
// 4-register accumulators as three-address values and moves them around under this kernel's pressure (1401 of 2592 instructions with
// D != C, 700 copies and 580 s_nop per step: profiles/r05_q).  Hazards the compiler no longer pads (cdna_hip_programming.md 5.7): a chain
// D -> C needs no state; the vector ALU reads accumulators only behind mfma_drain() (gate math, step tail); A / B operands come from
// ds_read / buffer_load results the compiler waits for by register.  -DCCSM_F3S_BUILTIN_MFMA builds the builtin form (A/B).
#ifdef CCSM_F3S_BUILTIN_MFMA
__device__ __forceinline__ f32x4 mfma32k(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(as_half8(a), as_half8(b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma32k_first(uint4 a, uint4 b, f32x4 c) { return mfma32k(a, b, c); }
__device__ __forceinline__ void mfma_drain() {}
template <int NB> __device__ __forceinline__ void mfma_drained(f32x4 (&)[2][NB][2]) {}
#else
__device__ __forceinline__ f32x4 mfma32k(uint4 a, uint4 b, f32x4 c) {
    const u32x4_t av = __builtin_bit_cast(u32x4_t, a), bv = __builtin_bit_cast(u32x4_t, b);
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
    return c;
}
// the FIRST product into an accumulator that the vector ALU has just written (the bias copies, which the compiler places right in front of
// their use; N behind the gate math): the two wait states a VALU write -> MFMA read wants, inside the statement (48 of a step's 2592 instructions)
// (-DCCSM_F3S_NO_FIRST_NOP, -DCCSM_F3S_NO_DRAINED: deliberately broken builds that tools/isa_gate.py must reject - tests/test_isa_gate.py)
__device__ __forceinline__ f32x4 mfma32k_first(uint4 a, uint4 b, f32x4 c) {
    const u32x4_t av = __builtin_bit_cast(u32x4_t, a), bv = __builtin_bit_cast(u32x4_t, b);
#ifdef CCSM_F3S_NO_FIRST_NOP
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
#else
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
#endif
    return c;
}
// every MFMA issued so far has written its result (4-pass instruction: 8 states would do; 16 here, once per phase)
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15" ::: "memory"); }
// ... and the vector ALU's readers of these accumulators take them from HERE: the drain's memory clobber orders loads and stores only - a
// register-only reader (the gate math's first multiply) may otherwise be scheduled between the last product and the drain, inside the wait
// states an XDL write -> VALU read needs (round 6: tools/isa_gate.py found exactly that in the 32-row instantiations).  asm volatile
// statements keep their order among themselves; these emit no instruction.
template <int NB>
__device__ __forceinline__ void mfma_drained(f32x4 (&a)[2][NB][2]) {
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int bt = 0; bt < NB; ++bt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#ifndef CCSM_F3S_NO_DRAINED
                asm volatile("" : "+v"(a[T][bt][h]));
#endif
            }
}
#endif

#define CCSM_FENCE asm volatile("" ::: "memory")
// Round 6 (profiles/r06_h, r06_j): the CU's vector-memory path takes one 1-KiB request per 16 cycles from all eight waves together and a wave
// that waits for its slot issues no MFMA.  gru_layer12_mx_kernel gained 1 % from issuing its weight requests ONE at a time, each behind the
// last use of the fragment it replaces (ccsm_gru_mx.hip: kMxIlv).  The same order for these kernels is written below (-DCCSM_F3S_ILV: same
// requests in the same order among themselves - the counted waits stand -, same products into every accumulator in the same order: same
// bits by construction, tools/isa_gate.py green) but was NOT measured: the GPU pool closed before its A/B run (tools/ab_bits.py +
// tools/ab_variants.sh f3ilv f3noilv), so the product keeps the order that round 5 validated.  With it the transfer skew is off (it would put
// waves 4-7's refill behind the interleaved requests; it measured nothing in round 5).
#ifdef CCSM_F3S_ILV
constexpr bool kF3sIlv = true;
#else
constexpr bool kF3sIlv = false;
#endif
#if defined(CCSM_F3S_NO_SKEW) || defined(CCSM_F3S_ILV)
constexpr bool kF3sSkew = false;
#else
constexpr bool kF3sSkew = true;
#endif

// Step tail: n = tanh(N); h' = n + z (h_{t-1} - n) for this lane's eight units of row n' of every 16-row sub-tile; fp16 hi + lo into the
// lane's own 16 bytes of the state fragments (LDS) and of the layer output (HBM).  accz = sigmoid(Z) already, accn = N.
// lx = this lane's byte offset inside a [kb][bt][hl] fragment array of NB row tiles (see the kernel), lo = the same for the output (NB = 1 strides)
template <int NB>
__device__ __forceinline__ void f3s_tail(char* smem, const f32x4 (&accz)[2][NB][2], const f32x4 (&accn)[2][NB][2], uint4* __restrict__ out, int tile0,
                                         int t, int dir, int wave, int lx, int lo_) {
    char* own = smem + (wave * (2 * NB * 2 * 1024) + lx);
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
        char* o = reinterpret_cast<char*>(out + (((size_t)(tile0 + bt) * kSeqLen + t) * kKB12 + (dir * kKBH + 2 * wave)) * 2 * kFragU4) + (uint32_t)lo_;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            char* p = own + (bt * 2048 + s * 256);
            const half8 hi = as_half8(*reinterpret_cast<const uint4*>(p));
            const half8 lo = as_half8(*reinterpret_cast<const uint4*>(p + 1024));
            _Float16 nh[8], nl[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float hp = (float)hi[j] + (float)lo[j];
                const float nn = tanh_fold(accn[j >> 2][bt][s][j & 3]);
                const float hn = (hp - nn) * accz[j >> 2][bt][s][j & 3] + nn;
                split16(hn, nh[j], nl[j]);
            }
            const uint4 h4 = make_uint4(pack2(nh[0], nh[1]), pack2(nh[2], nh[3]), pack2(nh[4], nh[5]), pack2(nh[6], nh[7]));
            const uint4 l4 = make_uint4(pack2(nl[0], nl[1]), pack2(nl[2], nl[3]), pack2(nl[4], nl[5]), pack2(nl[6], nl[7]));
            *reinterpret_cast<uint4*>(p) = h4;
            *reinterpret_cast<uint4*>(p + 1024) = l4;
            nt_store(h4, reinterpret_cast<uint4*>(o + s * 256));
            nt_store(l4, reinterpret_cast<uint4*>(o + s * 256 + 1024));
        }
    }
}

// DBG (only instantiated in a -DCCSM_PHASE_STAMPS build): workgroup 0 records the cycle counter at step start / behind phase A / B / C / the tail
template <int NB_ = kMxNB, bool DBG = false>
__global__ __launch_bounds__(512, 2) void gru_layer12_f3s_kernel(const uint4* __restrict__ xin, uint4* __restrict__ out, const uint4* __restrict__ wst,
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
    const u32x4_t xrs = dma_rsrc(xin + (size_t)tile0 * kSeqLen * KX * 2 * kFragU4);
    const unsigned sx_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)X_OFF;    // (cast first: see gru_layer0_mx_kernel)
    auto dma_pair = [&](int slot, int sd, int jd) {
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
    auto dma_ahead = [&](int slot, int s, int jj) {
        const int g = jj + RS;
        const int c = g & (2 * NPAIR - 1);
        dma_pair(slot, s + (g >> 5), kMxZigZag && c >= NPAIR ? 2 * NPAIR - 1 - c : c & (NPAIR - 1));
    };
    auto xfer_wait = [&](auto base_c, auto k_c) {
        constexpr int BASE = decltype(base_c)::value, K = decltype(k_c)::value;
        if (4 * NB > kWaves && wave < 4 * NB - kWaves) wait_vm<BASE + 2 * K>();
        else if (4 * NB >= kWaves || wave < 4 * NB) wait_vm<BASE + K>();
    };
#define CCSM_XW(BASE, K) xfer_wait(std::integral_constant<int, (BASE)>{}, std::integral_constant<int, (K)>{})

    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const char*>(wst) + (size_t)(dir * kWaves + wave) * kF3WBytes);
    const int bias_off = BIAS_OFF + wave * 4 * 32 * 4;
    auto w_at = [&](int off) -> uint4 { return buf_load(wrs, lane16, off); };

    // weight registers: phase A two pair slots [slot][unit tile][gate r, z] hi / lo; phase B one resident pair [unit tile][gate] hi / lo;
    // phase C four pair slots of the n gate [slot][unit tile] hi / lo
    uint4 wah[2][2][2], wal[2][2][2];
    uint4 wbh[2][3], wbl[2][3];
    uint4 wch[4][2], wcl[4][2];
    auto a_hi = [&](int p, int T, int g) -> uint4 { return w_at(p * PA + ((2 * T + g) << 10)); };
    auto a_lo = [&](int p, int T, int g) -> uint4 { return w_at(p * PA + ((4 + 2 * T + g) << 10)); };
    auto c_hi = [&](int pp, int T) -> uint4 { return w_at(OFF_C + pp * PC + (T << 10)); };
    auto c_lo = [&](int pp, int T) -> uint4 { return w_at(OFF_C + pp * PC + ((2 + T) << 10)); };

    // ---- prologue: the ring's first pairs, the two phase-A weight slots
#pragma unroll
    for (int g = 0; g < RS; ++g) dma_pair(g, 0, g);
#pragma unroll
    for (int ws = 0; ws < 2; ++ws)
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int g = 0; g < 2; ++g) { wal[ws][T][g] = a_lo(ws, T, g); wah[ws][T][g] = a_hi(ws, T, g); }
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
        // DBG: step 10 also records, per pair of phases A (0-15) and C (16-31): start / before the transfer wait / behind it / behind the barrier / end
        auto pstamp = [&](int idx, int k) {
            if constexpr (DBG) {
                if (dbg != nullptr && s == 10 && blockIdx.x == 0 && lane == 0) dbg[kSeqLen * kWaves * 5 + (wave * 32 + idx) * 5 + k] = __builtin_readcyclecounter();
            }
        };
        f32x4 acc[3][2][NB][2];                                     // [gate R, Z, N][unit tile][row tile][16-row half]
        // this lane's byte offset inside a [kb][bt][hl] fragment array with NB row tiles per k-block: k-block (q >> 1) of the pair, lane
        // position n' + 32 (q & 1) (+ 16 for the upper half: + 256 B); opaque, so that the address arithmetic is redone where a phase needs it
        int lxv;                                                    // (computed once per step; the copies handed out are opaque)
        {
            int v = lane;
            asm volatile("" : "+v"(v));
            const int kbo = (v & 32) << 6;                          // 2048 for the pair's second k-block
            lxv = (NB == 1 ? kbo : NB == 2 ? (kbo << 1) : kbo + (kbo << 1)) + ((((v >> 4) & 1) * 32 + (v & 15)) << 4);
        }
        auto lane_x = [&]() -> int {
            int v = lxv;
            asm volatile("" : "+v"(v));
            return v;
        };
        auto bias_set = [&](int set, f32x4 (&b)[2]) {               // b[T][i] = bias of unit 8 q + 4 T + i
            int v = lane;
            asm volatile("" : "+v"(v));
            const char* bp = smem + (bias_off + set * 128 + ((v >> 4) << 5));
            const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 16);
            b[0] = f32x4{b0.x, b0.y, b0.z, b0.w};
            b[1] = f32x4{b1.x, b1.y, b1.z, b1.w};
        };
        {
            f32x4 b0[2], b1[2];
            bias_set(0, b0);
            bias_set(1, b1);
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int h = 0; h < 2; ++h) { acc[0][T][bt][h] = b0[T]; acc[1][T][bt][h] = b1[T]; }
        }
        uint4 xh[2][NB], xl[2][NB];                                 // [16-row half][row tile]
        auto rdx = [&](uint4 (&x)[NB], int xs, int h, int f) {      // xs = byte offset of the slot + lane_x
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) x[bt] = *reinterpret_cast<const uint4*>(smem + xs + (((bt * 2 + f) << 10) + h * 256));
        };
        auto slot_off = [&](int sl) -> int { return X_OFF + sl * SLOT_BYTES + lane_x(); };

        // ---------------- phase A: R, Z += W_i{r,z} x_t, pairs 0..15; pair P lives in weight slot P & 1 ----------------------------
        // before the barrier: (W_hi + W_lo) x_hi of unit tile 0, then of unit tile 1, the tile's lo fragments refilled with pair P + 2 behind
        // each group; behind it: W_hi x_lo of both tiles, then the hi fragments.  Pairs 14 and 15 request phase B's first pair instead.
        rdx(xh[0], slot_off(slot), 0, 0);
        int slot_a15 = 0;
        static_for<0, NPAIR>([&](auto PC_) {
            constexpr int P = decltype(PC_)::value;
            constexpr int WS = P & 1;
            const int xs = slot_off(slot);
            const int slot_n = slot == RS - 1 ? 0 : slot + 1;
            pstamp(P, 0);
            rdx(xh[1], xs, 1, 0);
            CCSM_FENCE;
if constexpr (kF3sIlv && P + 2 < NPAIR) {
                static_for<0, 2>([&](auto GC) {
                    constexpr int g = decltype(GC)::value;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int bt = 0; bt < NB; ++bt) acc[g][0][bt][h] = P == 0 ? mfma32k_first(wah[WS][0][g], xh[h][bt], acc[g][0][bt][h]) : mfma32k(wah[WS][0][g], xh[h][bt], acc[g][0][bt][h]);
#pragma unroll
                        for (int bt = 0; bt < NB; ++bt) acc[g][0][bt][h] = mfma32k(wal[WS][0][g], xh[h][bt], acc[g][0][bt][h]);
                    }
                    CCSM_FENCE;
                    wal[WS][0][g] = a_lo(P + 2, 0, g);
                    CCSM_FENCE;
                });
            } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[g][0][bt][h] = P == 0 ? mfma32k_first(wah[WS][0][g], xh[h][bt], acc[g][0][bt][h]) : mfma32k(wah[WS][0][g], xh[h][bt], acc[g][0][bt][h]);
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[g][0][bt][h] = mfma32k(wal[WS][0][g], xh[h][bt], acc[g][0][bt][h]);
            }
            }
            CCSM_FENCE;
            if constexpr (kF3sIlv && P + 2 < NPAIR) {}
            else if constexpr (P + 2 < NPAIR) { wal[WS][0][0] = a_lo(P + 2, 0, 0); wal[WS][0][1] = a_lo(P + 2, 0, 1); }
            else if constexpr (P == NPAIR - 2) { wbh[0][0] = w_at(OFF_B + (0 << 10)); wbh[0][1] = w_at(OFF_B + (1 << 10)); }
            else { wbh[1][2] = w_at(OFF_B + (5 << 10)); wbl[1][0] = w_at(OFF_B + (9 << 10)); }
            rdx(xl[0], xs, 0, 1);
            rdx(xl[1], xs, 1, 1);
            CCSM_FENCE;
if constexpr (kF3sIlv && P + 2 < NPAIR) {
                static_for<0, 2>([&](auto GC) {
                    constexpr int g = decltype(GC)::value;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int bt = 0; bt < NB; ++bt) acc[g][1][bt][h] = P == 0 ? mfma32k_first(wah[WS][1][g], xh[h][bt], acc[g][1][bt][h]) : mfma32k(wah[WS][1][g], xh[h][bt], acc[g][1][bt][h]);
#pragma unroll
                        for (int bt = 0; bt < NB; ++bt) acc[g][1][bt][h] = mfma32k(wal[WS][1][g], xh[h][bt], acc[g][1][bt][h]);
                    }
                    CCSM_FENCE;
                    wal[WS][1][g] = a_lo(P + 2, 1, g);
                    CCSM_FENCE;
                });
            } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[g][1][bt][h] = P == 0 ? mfma32k_first(wah[WS][1][g], xh[h][bt], acc[g][1][bt][h]) : mfma32k(wah[WS][1][g], xh[h][bt], acc[g][1][bt][h]);
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[g][1][bt][h] = mfma32k(wal[WS][1][g], xh[h][bt], acc[g][1][bt][h]);
            }
            }
            CCSM_FENCE;
            if constexpr (kF3sIlv && P + 2 < NPAIR) {}
            else if constexpr (P + 2 < NPAIR) { wal[WS][1][0] = a_lo(P + 2, 1, 0); wal[WS][1][1] = a_lo(P + 2, 1, 1); }
            else if constexpr (P == NPAIR - 2) { wbh[0][2] = w_at(OFF_B + (2 << 10)); wbl[0][0] = w_at(OFF_B + (6 << 10)); }
            else { wbl[1][1] = w_at(OFF_B + (10 << 10)); wbl[1][2] = w_at(OFF_B + (11 << 10)); }
            // counted wait for this wave's part of the next pair's transfer: ccsm_gru_f3.hip (the operation counts are the same)
            pstamp(P, 1);
            {
                constexpr int FULL = 8 + 8 * (RS - 2), EARLY = 6 + 4 * (RS - 2) + 4 * P + 4 * NB;
                CCSM_XW((P < RS - 1 && EARLY < FULL ? EARLY : FULL), RS - 2);
            }
            pstamp(P, 2);
            __syncthreads();             // the next pair is in LDS; every wave has read this pair's operands
            pstamp(P, 3);
            // the vacated slot is refilled at once (pair 15: behind phase B) - by waves 0-3 in front of the W_hi x_lo group, by waves 4-7 behind
            // it (kF3sSkew): the same order among a wave's vector-memory operations, but the eight waves' transfer instructions no longer
            // reach the CU's vector-memory path in one burst while no MFMA runs
            if constexpr (P + 1 < NPAIR) { if (!kF3sSkew || wave < 4) dma_ahead(slot, s, P); } else slot_a15 = slot;
            if constexpr (P + 1 < NPAIR) rdx(xh[0], slot_off(slot_n), 0, 0);
            CCSM_FENCE;
            if constexpr (kF3sIlv && P + 2 < NPAIR) {
                static_for<0, 4>([&](auto FC) {
                    constexpr int T = decltype(FC)::value >> 1, g = decltype(FC)::value & 1;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int bt = 0; bt < NB; ++bt) acc[g][T][bt][h] = mfma32k(wah[WS][T][g], xl[h][bt], acc[g][T][bt][h]);
                    CCSM_FENCE;
                    wah[WS][T][g] = a_hi(P + 2, T, g);
                    CCSM_FENCE;
                });
            } else {
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                        for (int g = 0; g < 2; ++g) acc[g][T][bt][h] = mfma32k(wah[WS][T][g], xl[h][bt], acc[g][T][bt][h]);
            }
            CCSM_FENCE;
            if constexpr (P + 1 < NPAIR) { if (kF3sSkew && wave >= 4) dma_ahead(slot, s, P); }
            CCSM_FENCE;
            if constexpr (kF3sIlv && P + 2 < NPAIR) {
            } else if constexpr (P + 2 < NPAIR) {
                wah[WS][0][0] = a_hi(P + 2, 0, 0); wah[WS][0][1] = a_hi(P + 2, 0, 1); wah[WS][1][0] = a_hi(P + 2, 1, 0); wah[WS][1][1] = a_hi(P + 2, 1, 1);
            } else if constexpr (P == NPAIR - 2) {
                wbl[0][1] = w_at(OFF_B + (7 << 10)); wbl[0][2] = w_at(OFF_B + (8 << 10)); wbh[1][0] = w_at(OFF_B + (3 << 10)); wbh[1][1] = w_at(OFF_B + (4 << 10));
            }
            CCSM_FENCE;
            pstamp(P, 4);
            slot = slot_n;
        });

        stamp(1);
        // ---------------- phase B: R, Z, N += W_h{r,z,n} h_{t-1}  (N starts at b_hn): three passes per pair of k-blocks on the fp16 hi + lo
        // state; one pair of weights resident.  Row halves in turn: half 0 against both unit tiles, half 1 against unit tile 0 - its six
        // fragments refilled with the next pair's - then against unit tile 1, refilled likewise; the last pair's positions take phase C's
        // first two pair slots.  The other order (unit tiles outermost: 54 instead of 27 MFMAs between a refill and its use, every B operand
        // read from LDS twice) needs 1 % fewer CYCLES and 1 % more TIME (profiles/r05_v_ab_b_order.log): at the power cap the extra LDS
        // reads cost more than the idle cycles they remove -------------------------------------------------------------------------------
        {
            f32x4 b3[2];
            bias_set(3, b3);
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int h = 0; h < 2; ++h) acc[2][T][bt][h] = b3[T];
        }
        // B operands of group (pair Q, row half H): the hi fragments -> buffer H, requested one group ahead (a group is 27 MFMAs = 432
        // cycles: an LDS read issued in front of it would be waited for on the spot)
        auto rdh = [&](auto QC, auto HC) {                          // the hi fragments: one group ahead
            constexpr int Q = decltype(QC)::value, H = decltype(HC)::value;
            const int hs = lane_x() + ((2 * Q * NB * 2) << 10);     // mx_hfrag(2 Q, 0, 0) + lane_x
            rdx(xh[H], hs, H, 0);
        };
        auto rdl = [&](auto QC, auto HC) {                          // the lo fragments: at the group's start (first used behind 18 MFMAs)
            constexpr int Q = decltype(QC)::value, H = decltype(HC)::value;
            const int hs = lane_x() + ((2 * Q * NB * 2) << 10);
            rdx(xl[0], hs, H, 1);
        };
        rdh(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        static_for<0, kKBH / 2>([&](auto QC) {
            constexpr int Q = decltype(QC)::value;
            constexpr int NXT = OFF_B + (Q + 1) * PB;
            auto group = [&](auto TC, auto HC) {
                constexpr int T = decltype(TC)::value, h = decltype(HC)::value;
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g][T][bt][h] = Q == 0 ? mfma32k_first(wbh[T][g], xh[h][bt], acc[g][T][bt][h]) : mfma32k(wbh[T][g], xh[h][bt], acc[g][T][bt][h]);
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g][T][bt][h] = mfma32k(wbl[T][g], xh[h][bt], acc[g][T][bt][h]);
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g][T][bt][h] = mfma32k(wbh[T][g], xl[0][bt], acc[g][T][bt][h]);
            };
#ifndef CCSM_F3S_B_TMAJOR        // row halves outermost: each B operand is read from LDS once per pair (-DCCSM_F3S_B_TMAJOR: unit tiles outermost, below)
            static_for<0, 2>([&](auto HC) {
                constexpr int H = decltype(HC)::value;
                if constexpr (H == 0) rdh(QC, std::integral_constant<int, 1>{});
                else if constexpr (Q + 1 < kKBH / 2) rdh(std::integral_constant<int, Q + 1>{}, std::integral_constant<int, 0>{});
                rdl(QC, HC);
                CCSM_FENCE;
                static_for<0, 2>([&](auto TC) {
                    constexpr int T = decltype(TC)::value;
                    if constexpr (kF3sIlv && H == 1 && Q + 1 < kKBH / 2) {
                        // gate by gate: nine products, then the gate's hi and lo fragments of the next pair
                        static_for<0, 3>([&](auto GC) {
                            constexpr int g = decltype(GC)::value;
#pragma unroll
                            for (int bt = 0; bt < NB; ++bt) acc[g][T][bt][H] = Q == 0 ? mfma32k_first(wbh[T][g], xh[H][bt], acc[g][T][bt][H]) : mfma32k(wbh[T][g], xh[H][bt], acc[g][T][bt][H]);
#pragma unroll
                            for (int bt = 0; bt < NB; ++bt) acc[g][T][bt][H] = mfma32k(wbl[T][g], xh[H][bt], acc[g][T][bt][H]);
#pragma unroll
                            for (int bt = 0; bt < NB; ++bt) acc[g][T][bt][H] = mfma32k(wbh[T][g], xl[0][bt], acc[g][T][bt][H]);
                            CCSM_FENCE;
                            wbh[T][g] = w_at(NXT + ((3 * T + g) << 10));
                            wbl[T][g] = w_at(NXT + ((6 + 3 * T + g) << 10));
                            CCSM_FENCE;
                        });
                        return;
                    }
                    group(TC, HC);
                    CCSM_FENCE;
                    if constexpr (H == 1) {
                        if constexpr (Q + 1 < kKBH / 2) {
#pragma unroll
                            for (int g = 0; g < 3; ++g) { wbh[T][g] = w_at(NXT + ((3 * T + g) << 10)); wbl[T][g] = w_at(NXT + ((6 + 3 * T + g) << 10)); }
                        } else {
                            wch[T][0] = c_hi(T, 0); wch[T][1] = c_hi(T, 1); wcl[T][0] = c_lo(T, 0); wcl[T][1] = c_lo(T, 1);
                        }
                        CCSM_FENCE;
                    }
                });
            });
#else
            static_for<0, 2>([&](auto TC) {
                constexpr int T = decltype(TC)::value;
                static_for<0, 2>([&](auto HC) {
                    constexpr int H = decltype(HC)::value;
                    // the next group's operands into the other buffer
                    if constexpr (H == 0) rdh(QC, std::integral_constant<int, 1>{});
                    else if constexpr (T == 0) rdh(QC, std::integral_constant<int, 0>{});
                    else if constexpr (Q + 1 < kKBH / 2) rdh(std::integral_constant<int, Q + 1>{}, std::integral_constant<int, 0>{});
                    rdl(QC, HC);
                    CCSM_FENCE;
                    group(TC, HC);
                    CCSM_FENCE;
                });
                if constexpr (Q + 1 < kKBH / 2) {
#pragma unroll
                    for (int g = 0; g < 3; ++g) { wbh[T][g] = w_at(NXT + ((3 * T + g) << 10)); wbl[T][g] = w_at(NXT + ((6 + 3 * T + g) << 10)); }
                } else {
                    wch[T][0] = c_hi(T, 0); wch[T][1] = c_hi(T, 1); wcl[T][0] = c_lo(T, 0); wcl[T][1] = c_lo(T, 1);
                }
                CCSM_FENCE;
            });
#endif
        });
        // r = sigmoid(R) ; N = b_in + r * N
        mfma_drain();
        mfma_drained<NB>(acc[0]);
        mfma_drained<NB>(acc[1]);
        mfma_drained<NB>(acc[2]);
        {
            f32x4 b2[2];
            bias_set(2, b2);
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[2][T][bt][h][i] = b2[T][i] + sigmoid_f(acc[0][T][bt][h][i]) * acc[2][T][bt][h][i];
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
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = sigmoid_f(acc[1][T][bt][h][i]);
                        asm volatile("" : "+v"(v));                 // pins the evaluation HERE (the compiler otherwise sinks it to the tail)
                        acc[1][T][bt][h][i] = v;
                    }
        };

        stamp(2);
        // ---------------- phase C: N += W_in x_t, pairs 0..15 (consumptions 16..31, zig-zag); pair P lives in slot P & 3, refilled with
        // pair P + 4 (1 + 1 requests before the barrier, 2 behind it); pairs 12..15 request the next step's phase-A slots instead ----------
        rdx(xh[0], slot_off(slot), 0, 0);
        static_for<0, NPAIR>([&](auto PC_) {
            constexpr int P = decltype(PC_)::value;
            constexpr int WS = P & 3;
            constexpr int AS = P >= NPAIR - 4 ? (P - (NPAIR - 4)) / 2 : 0, AH = P >= NPAIR - 4 ? (P - (NPAIR - 4)) & 1 : 0;   // pairs 12..15: phase-A slot AS, AH = 0 lo / 1 hi
            const int xs = slot_off(slot);
            const int slot_n = slot == RS - 1 ? 0 : slot + 1;
            pstamp(NPAIR + P, 0);
            rdx(xh[1], xs, 1, 0);
            CCSM_FENCE;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) acc[2][0][bt][h] = P == 0 ? mfma32k_first(wch[WS][0], xh[h][bt], acc[2][0][bt][h]) : mfma32k(wch[WS][0], xh[h][bt], acc[2][0][bt][h]);
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) acc[2][0][bt][h] = mfma32k(wcl[WS][0], xh[h][bt], acc[2][0][bt][h]);
            }
            CCSM_FENCE;
            if constexpr (P + 4 < NPAIR) wcl[WS][0] = c_lo(P + 4, 0);
            else if constexpr (AH == 0) wal[AS][0][0] = a_lo(AS, 0, 0); else wah[AS][0][0] = a_hi(AS, 0, 0);
            rdx(xl[0], xs, 0, 1);
            rdx(xl[1], xs, 1, 1);
            CCSM_FENCE;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) acc[2][1][bt][h] = P == 0 ? mfma32k_first(wch[WS][1], xh[h][bt], acc[2][1][bt][h]) : mfma32k(wch[WS][1], xh[h][bt], acc[2][1][bt][h]);
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) acc[2][1][bt][h] = mfma32k(wcl[WS][1], xh[h][bt], acc[2][1][bt][h]);
            }
            CCSM_FENCE;
            if constexpr (P + 4 < NPAIR) wcl[WS][1] = c_lo(P + 4, 1);
            else if constexpr (AH == 0) wal[AS][0][1] = a_lo(AS, 0, 1); else wah[AS][0][1] = a_hi(AS, 0, 1);
            pstamp(NPAIR + P, 1);
            if constexpr (P < RS - 2) CCSM_XW(2 + 4 * P, P + 1);
            else if constexpr (P == RS - 2) CCSM_XW(2 + 4 * (RS - 2), RS - 2);
            else CCSM_XW(4 + 4 * (RS - 2), RS - 2);
            pstamp(NPAIR + P, 2);
            __syncthreads();
            pstamp(NPAIR + P, 3);
            if (!kF3sSkew || wave < 4) dma_ahead(slot, s, NPAIR + P);   // the vacated slot is refilled at once (waves 4-7: behind the group, as in phase A)
            if constexpr (P + 1 < NPAIR) rdx(xh[0], slot_off(slot_n), 0, 0);
            CCSM_FENCE;
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt) acc[2][T][bt][h] = mfma32k(wch[WS][T], xl[h][bt], acc[2][T][bt][h]);
            CCSM_FENCE;
            if (kF3sSkew && wave >= 4) dma_ahead(slot, s, NPAIR + P);
            CCSM_FENCE;
            if constexpr (P + 4 < NPAIR) { wch[WS][0] = c_hi(P + 4, 0); wch[WS][1] = c_hi(P + 4, 1); }
            else if constexpr (AH == 0) { wal[AS][1][0] = a_lo(AS, 1, 0); wal[AS][1][1] = a_lo(AS, 1, 1); }
            else { wah[AS][1][0] = a_hi(AS, 1, 0); wah[AS][1][1] = a_hi(AS, 1, 1); }
            CCSM_FENCE;
            pstamp(NPAIR + P, 4);
            slot = slot_n;
            if constexpr (P == 1) zwork(0);
            if constexpr (P == 5 && NB > 1) zwork(1);
            if constexpr (P == 9 && NB > 2) zwork(2);
        });
        stamp(3);
        mfma_drain();
        mfma_drained<NB>(acc[2]);
        {
            int v = lane;
            asm volatile("" : "+v"(v));
            const int lo_ = ((v & 32) << 6) + ((((v >> 4) & 1) * 32 + (v & 15)) << 4);
            f3s_tail<NB>(smem, acc[1], acc[2], out, tile0, t, dir, wave, lane_x(), lo_);
        }
        CCSM_FENCE;
        stamp(4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // no transfer may still be writing LDS when the workgroup retires
#undef CCSM_XW
}

// ---------------------------------------------------------------------------------------------------------
// Layer 0 in the same arithmetic and instruction shape (lock-step form: two barriers per step).  K = 11 ... 16 input columns = ONE k-block
// of fp16 hi | lo fragments; the 16x16x32 instruction takes 32 k, so its B operand is [x_hi | x_lo] (lane (n', q): fragment hl = q >> 1)
// against A1 = [W_hi | W_hi] and A2 = [W_lo | 0]: all three passes of the split product in two instructions.
//   xin : [tile][t][hi|lo][64] uint4                  out : [tile][t][32 kb][hi | lo][64] uint4
//   wst : per (direction, wave): phase A  (T, g in r, z): A1 at (4 T + 2 g) KiB, A2 at (4 T + 2 g + 1) KiB               = 8 KiB
//                                phase B  8 pairs as in layers 1-2 (hi (T, g) at (3 T + g) KiB | lo at (6 + 3 T + g) KiB)   = 12 KiB x 8
//                                phase C  (T): A1 at 2 T KiB, A2 at (2 T + 1) KiB                                         = 4 KiB
//   LDS : the layer-0 layout of ccsm_gru_mx.hip (h fragments | x double buffer | (unused) | biases)
// ---------------------------------------------------------------------------------------------------------
constexpr int kF3s0OffB = 8 * 1024, kF3s0OffC = kF3s0OffB + (kKBH / 2) * kF3PairB, kF3s0WBytes = kF3s0OffC + 4 * 1024;

template <int NB_ = kMxNB>
__global__ __launch_bounds__(512, 2) void gru_layer0_f3s_kernel(const uint4* __restrict__ xin, uint4* __restrict__ out, const uint4* __restrict__ wst,
                                                                 const float* __restrict__ bias, const float* __restrict__ h0, int rows_p) {
    constexpr int NB = NB_;
    constexpr int X_OFF = mx0_xoff(NB), BIAS_OFF = mx0_biasoff(NB);
    constexpr int PB = kF3PairB, OFF_B = kF3s0OffB, OFF_C = kF3s0OffC;
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

    const u32x4_t xrs = dma_rsrc(xin + (size_t)tile0 * kSeqLen * 2 * kFragU4);
    const unsigned sx_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)X_OFF;    // (cast first: see gru_layer0_mx_kernel)
    auto stage_load = [&](int t, int buf) {                         // 2 NB fragments per step (bt x hi|lo); every wave issues ONE transfer (the
        const int f = wave < 2 * NB ? wave : 2 * NB - 1;            // spare waves re-stage the last fragment: same bytes, same place)
        const int hl = f & 1, bt = f >> 1;
        const int soff = (((bt * kSeqLen + t) * 2 + hl) << 10);
        dma16_buf(xrs, lane16, __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane((int)(sx_base + ((buf * 2 * NB + f) << 10))));
    };
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const char*>(wst) + (size_t)(dir * kWaves + wave) * kF3s0WBytes);
    const int bias_off = BIAS_OFF + wave * 4 * 32 * 4;
    auto w_at = [&](int off) -> uint4 { return buf_load(wrs, lane16, off); };

    uint4 wxa[2][2][2];                                             // phase A: [unit tile][gate r, z][A1, A2]
    uint4 wxc[2][2];                                                // phase C: [unit tile][A1, A2]
    uint4 wbh[2][3], wbl[2][3];                                     // phase B resident pair: [unit tile][gate] hi / lo
    auto ld_first = [&]() {                                         // everything a step needs before its second phase-B pair: 20 requests
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int g = 0; g < 2; ++g) { wxa[T][g][0] = w_at((4 * T + 2 * g) << 10); wxa[T][g][1] = w_at((4 * T + 2 * g + 1) << 10); }
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int g = 0; g < 3; ++g) { wbh[T][g] = w_at(OFF_B + ((3 * T + g) << 10)); wbl[T][g] = w_at(OFF_B + ((6 + 3 * T + g) << 10)); }
    };
    stage_load(dir ? kSeqLen - 1 : 0, 0);
    ld_first();
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");               // the first transfer (older than the 20 weight requests)

    for (int s = 0; s < kSeqLen; ++s) {
        const int t = dir ? (kSeqLen - 1 - s) : s;
        const int tn = s + 1 < kSeqLen ? (dir ? t - 1 : t + 1) : t;
        f32x4 acc[3][2][NB][2];                                     // [gate R, Z, N][unit tile][row tile][16-row half]
        auto lane_q = [&]() -> int {
            int v = lane;
            asm volatile("" : "+v"(v));
            return v;
        };
        int lpv, lxv;                                               // lane position n' + 32 (q & 1) in bytes; + the k-block (q >> 1) of a state pair
        {
            const int v = lane_q();
            const int kbo = (v & 32) << 6;
            lpv = (((v >> 4) & 1) * 32 + (v & 15)) << 4;
            lxv = lpv + (NB == 1 ? kbo : NB == 2 ? (kbo << 1) : kbo + (kbo << 1));
        }
        auto lane_pos = [&]() -> int { int v = lpv; asm volatile("" : "+v"(v)); return v; };
        auto lane_xs = [&]() -> int { int v = lxv; asm volatile("" : "+v"(v)); return v; };
        auto bias_set = [&](int set, f32x4 (&b)[2]) {               // b[T][i] = bias of unit 8 q + 4 T + i
            const char* bp = smem + (bias_off + set * 128 + ((lane_q() >> 4) << 5));
            const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 16);
            b[0] = f32x4{b0.x, b0.y, b0.z, b0.w};
            b[1] = f32x4{b1.x, b1.y, b1.z, b1.w};
        };
        {
            f32x4 b0[2], b1[2];
            bias_set(0, b0);
            bias_set(1, b1);
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int h = 0; h < 2; ++h) { acc[0][T][bt][h] = b0[T]; acc[1][T][bt][h] = b1[T]; }
        }
        // the transfer of this step's x (issued one step ago) is older than the 20 weight requests and 4 NB output stores of the tail
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(20 + 4 * NB) : "memory");
        __syncthreads();                                            // x_t in LDS; everybody's h_{t-1} fragments written
        stage_load(tn, (s + 1) & 1);
        // B operand [x_hi | x_lo] of rows [16 h, 16 h + 16) of row tile bt: fragment hl = q >> 1, lane position n' + 32 (q & 1) + 16 h
        uint4 x0[2][NB];
        auto rd_x0 = [&]() {
            const int lx0 = lane_pos() + ((lane_q() & 32) << 5);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
                    x0[h][bt] = *reinterpret_cast<const uint4*>(smem + X_OFF + ((((s & 1) * NB + bt) * 2) << 10) + h * 256 + lx0);
        };
        // ---------------- phase A: R, Z += W_i{r,z} x_t ---------------------------------------------------------------------------
        rd_x0();
        CCSM_FENCE;
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[g][T][bt][h] = mfma32k_first(wxa[T][g][0], x0[h][bt], acc[g][T][bt][h]);
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[g][T][bt][h] = mfma32k(wxa[T][g][1], x0[h][bt], acc[g][T][bt][h]);
        CCSM_FENCE;
        // ---------------- phase B: R, Z, N += W_h{r,z,n} h_{t-1}  (N starts at b_hn): gru_layer12_f3s_kernel's ---------------------
        {
            f32x4 b3[2];
            bias_set(3, b3);
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int h = 0; h < 2; ++h) acc[2][T][bt][h] = b3[T];
        }
        uint4 xh[2][NB], xl[1][NB];
        auto rdh = [&](auto QC, auto HC) {                          // B operands of group (pair Q, row half H): the hi fragments -> buffer H, one group ahead
            constexpr int Q = decltype(QC)::value, H = decltype(HC)::value;
            const int hs = lane_xs() + ((2 * Q * NB * 2) << 10);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) xh[H][bt] = *reinterpret_cast<const uint4*>(smem + hs + ((bt * 2) << 10) + H * 256);
        };
        auto rdl = [&](auto QC, auto HC) {                          // the lo fragments: at the group's start (first used behind 18 MFMAs)
            constexpr int Q = decltype(QC)::value, H = decltype(HC)::value;
            const int hs = lane_xs() + ((2 * Q * NB * 2) << 10);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) xl[0][bt] = *reinterpret_cast<const uint4*>(smem + hs + ((bt * 2 + 1) << 10) + H * 256);
        };
        rdh(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        static_for<0, kKBH / 2>([&](auto QC) {
            constexpr int Q = decltype(QC)::value;
            constexpr int NXT = OFF_B + (Q + 1) * PB;
            auto group = [&](auto TC, auto HC) {
                constexpr int T = decltype(TC)::value, h = decltype(HC)::value;
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g][T][bt][h] = Q == 0 ? mfma32k_first(wbh[T][g], xh[h][bt], acc[g][T][bt][h]) : mfma32k(wbh[T][g], xh[h][bt], acc[g][T][bt][h]);
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g][T][bt][h] = mfma32k(wbl[T][g], xh[h][bt], acc[g][T][bt][h]);
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g][T][bt][h] = mfma32k(wbh[T][g], xl[0][bt], acc[g][T][bt][h]);
            };
#ifndef CCSM_F3S_B_TMAJOR        // row halves outermost: each B operand is read from LDS once per pair (-DCCSM_F3S_B_TMAJOR: unit tiles outermost, below)
            static_for<0, 2>([&](auto HC) {
                constexpr int H = decltype(HC)::value;
                if constexpr (H == 0) rdh(QC, std::integral_constant<int, 1>{});
                else if constexpr (Q + 1 < kKBH / 2) rdh(std::integral_constant<int, Q + 1>{}, std::integral_constant<int, 0>{});
                rdl(QC, HC);
                CCSM_FENCE;
                static_for<0, 2>([&](auto TC) {
                    constexpr int T = decltype(TC)::value;
                    if constexpr (kF3sIlv && H == 1 && Q + 1 < kKBH / 2) {
                        // gate by gate: nine products, then the gate's hi and lo fragments of the next pair
                        static_for<0, 3>([&](auto GC) {
                            constexpr int g = decltype(GC)::value;
#pragma unroll
                            for (int bt = 0; bt < NB; ++bt) acc[g][T][bt][H] = Q == 0 ? mfma32k_first(wbh[T][g], xh[H][bt], acc[g][T][bt][H]) : mfma32k(wbh[T][g], xh[H][bt], acc[g][T][bt][H]);
#pragma unroll
                            for (int bt = 0; bt < NB; ++bt) acc[g][T][bt][H] = mfma32k(wbl[T][g], xh[H][bt], acc[g][T][bt][H]);
#pragma unroll
                            for (int bt = 0; bt < NB; ++bt) acc[g][T][bt][H] = mfma32k(wbh[T][g], xl[0][bt], acc[g][T][bt][H]);
                            CCSM_FENCE;
                            wbh[T][g] = w_at(NXT + ((3 * T + g) << 10));
                            wbl[T][g] = w_at(NXT + ((6 + 3 * T + g) << 10));
                            CCSM_FENCE;
                        });
                        return;
                    }
                    group(TC, HC);
                    CCSM_FENCE;
                    if constexpr (H == 1) {
                        if constexpr (Q + 1 < kKBH / 2) {
#pragma unroll
                            for (int g = 0; g < 3; ++g) { wbh[T][g] = w_at(NXT + ((3 * T + g) << 10)); wbl[T][g] = w_at(NXT + ((6 + 3 * T + g) << 10)); }
                        } else {
                            wxc[T][0] = w_at(OFF_C + ((2 * T) << 10)); wxc[T][1] = w_at(OFF_C + ((2 * T + 1) << 10));
                        }
                        CCSM_FENCE;
                    }
                });
            });
#else
            static_for<0, 2>([&](auto TC) {
                constexpr int T = decltype(TC)::value;
                static_for<0, 2>([&](auto HC) {
                    constexpr int H = decltype(HC)::value;
                    if constexpr (H == 0) rdh(QC, std::integral_constant<int, 1>{});
                    else if constexpr (T == 0) rdh(QC, std::integral_constant<int, 0>{});
                    else if constexpr (Q + 1 < kKBH / 2) rdh(std::integral_constant<int, Q + 1>{}, std::integral_constant<int, 0>{});
                    rdl(QC, HC);
                    CCSM_FENCE;
                    group(TC, HC);
                    CCSM_FENCE;
                });
                if constexpr (Q + 1 < kKBH / 2) {
#pragma unroll
                    for (int g = 0; g < 3; ++g) { wbh[T][g] = w_at(NXT + ((3 * T + g) << 10)); wbl[T][g] = w_at(NXT + ((6 + 3 * T + g) << 10)); }
                } else {
                    wxc[T][0] = w_at(OFF_C + ((2 * T) << 10)); wxc[T][1] = w_at(OFF_C + ((2 * T + 1) << 10));
                }
                CCSM_FENCE;
            });
#endif
        });
        // r = sigmoid(R) ; N = b_in + r * N
        mfma_drain();
        mfma_drained<NB>(acc[0]);
        mfma_drained<NB>(acc[1]);
        mfma_drained<NB>(acc[2]);
        {
            f32x4 b2[2];
            bias_set(2, b2);
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[2][T][bt][h][i] = b2[T][i] + sigmoid_f(acc[0][T][bt][h][i]) * acc[2][T][bt][h][i];
        }
        // ---------------- phase C: N += W_in x_t (x_t is still in its buffer) ------------------------------------------------------
        rd_x0();
        CCSM_FENCE;
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) acc[2][T][bt][h] = mfma32k_first(wxc[T][0], x0[h][bt], acc[2][T][bt][h]);       // (N was just written by the gate math)
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) acc[2][T][bt][h] = mfma32k(wxc[T][1], x0[h][bt], acc[2][T][bt][h]);
        CCSM_FENCE;
        mfma_drain();
        mfma_drained<NB>(acc[2]);
        ld_first();                                                 // the next step's first weight fragments: in flight during the tail
        CCSM_FENCE;
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[1][T][bt][h][i] = sigmoid_f(acc[1][T][bt][h][i]);
        __syncthreads();                                            // every wave has read h_{t-1} (phase B) before anybody overwrites its fragments
        {
            const int lo_ = lane_pos() + ((lane_q() & 32) << 6);
            f3s_tail<NB>(smem, acc[1], acc[2], out, tile0, t, dir, wave, lane_xs(), lo_);
        }
        CCSM_FENCE;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // no transfer may still be writing LDS when the workgroup retires
}
#undef CCSM_FENCE

}  // namespace ccsm
