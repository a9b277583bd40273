// libccsm, split-mx arithmetic: the input part of the n gate of GRU layers 1-2 as a product of its own (round 5).
//
//     G_n[row][t][dir][unit] = b_in[dir][unit] + sum_k W_in[dir][unit][k] x_t[row][k]           (reference: models.py:125-130, the n gate's
//                                                                                               gi term of torch.nn.GRU)
//
// Why.  Inside gru_layer12_mx_kernel this product is "phase C": one gate, so every B operand read from LDS feeds ONE MFMA, the weight
// fragments are reused over 96 rows only, x_t is fetched a second time, and a barrier closes every pair of k-blocks - the phase runs at
// 37 % of the MFMA rate and takes a third of the step (profiles/r02_i_phases_product_kernels.log).  Nothing in it depends on the
// recurrence.  Here it runs for all timesteps at once:
//   * a block = 2 row tiles x 2 timesteps = 128 rows of x, BOTH directions (the two directions of a layer read the same input): a B operand
//     read feeds two MFMAs, a weight fragment four; x_t is read once for both directions;
//   * persistent workgroups (8 waves; wave w owns hidden units [32 w, 32 w + 32) of both directions: 2 x 4 accumulator tiles), a six-slot ring
//     of x pairs (96 KiB) filled by LDS-DMA two iterations ahead, ONE barrier per TWO pairs, counted s_waitcnt, the ring running on across
//     the workgroup's blocks;
//   * the weights are phase C's own streams (ccsm_create packs nothing new): position P of a (direction, wave) stream holds pair
//     kMxZigZag ? 15 - P : P.
// The recurrent kernel (NXP instantiation of gru_layer12_mx_kernel) adds G_n behind its phase B:  N = G_n + sigmoid(R) (W_hn h + b_hn).
// The products and their order are phase C's (main k-block 0, main k-block 1, correction, pair by pair in stream order), but the sum starts
// from b_in instead of b_in + r N_h: the last bits of N differ from the fused kernel's (fp32 rounding order), so EVERY launch form (96-,
// 64-, 32-row workgroups) runs this way - a site's bits must not depend on the launch it travels in.
//   xin  : [tile][t][32 kb][hi | corr][64] uint4   (the layer input: ccsm_gru_mx.hip)
//   gout : [tile][t][dir][wave][q 4][64] uint4     = the 16 fp32 accumulator registers of lane (n, hh): units 8 (r >> 2) + 4 hh + (r & 3)
#include <hip/hip_runtime.h>

namespace ccsm {

constexpr int kNxBT = 4;                                   // B tiles of a block: [row tile 2][timestep 2]
constexpr int kNxRS = 6;                                   // ring slots
constexpr int kNxSlotBytes = 2 * kNxBT * 2 * 1024;         // [kbl 2][bt 4][hi | corr] x 1 KiB
constexpr int kNxLds = kNxRS * kNxSlotBytes;               // 96 KiB
constexpr int kNxTQ = (kSeqLen + 1) / 2;                   // timestep pairs per tile pair (the last one holds t = 20 twice)

__device__ __forceinline__ uint32_t f2u(float v) { return __builtin_bit_cast(uint32_t, v); }

__global__ __launch_bounds__(512, 2) void gru_nx_mx_kernel(const uint4* __restrict__ xin, uint4* __restrict__ gout, const uint4* __restrict__ wst,
                                                            const float* __restrict__ bias, int n_tile_pairs) {
    constexpr int PC = kMxPairC, OFF_C = mx12_off_c(false, false), WB = mx12_wbytes(false, false);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hh = lane >> 5;
    const int lane16 = lane * 16;
    const int sb = hh ? kMxScaleLo : kMxScaleHi;
    const int n_blocks = n_tile_pairs * kNxTQ;
    if ((int)blockIdx.x >= n_blocks) return;
    const int my_blocks = (n_blocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int n_cons = my_blocks * 16;                      // consumptions (pairs of k-blocks) of this workgroup
    const unsigned sx_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- x transfers: fragment f = (kbl * 4 + b) * 2 + hl of a slot, b = ti * 2 + tj; wave w moves fragments w and w + 8
    auto dma_cons = [&](int c0) {                           // wave-uniform; beyond the last consumption: that one again (same bytes, same place)
        const int c = c0 < n_cons ? c0 : n_cons - 1;
        const int id = (int)blockIdx.x + (c >> 4) * (int)gridDim.x;
        const int tp = id / kNxTQ, tq = id - tp * kNxTQ;
        const int P = c & 15;
        const int jd = kMxZigZag ? 15 - P : P;
        const int slot = c % kNxRS;
        const u32x4_t xrs = dma_rsrc(xin + (size_t)(2 * tp) * kSeqLen * kKB12 * 2 * kFragU4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = wave + 8 * i;
            const int hl = f & 1, b = (f >> 1) & 3, kbl = f >> 3;
            const int ti = b >> 1, t = min(2 * tq + (b & 1), kSeqLen - 1);
            const int soff = ((((ti * kSeqLen + t) * kKB12 + (2 * jd + kbl)) * 2 + hl) << 10);
            dma16_buf(xrs, lane16, __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane((int)(sx_base + slot * kNxSlotBytes + (f << 10))));
        }
    };

    // ---- weights: two register slots (consumption parity); 10 requests per consumption
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const char*>(wst) + (size_t)wave * WB);
    uint4 wh[2][2][2], wb0[2][2];                           // [slot][dir][kbl], [slot][dir]
    uint2 wb1[2][2];
    uint32_t wsc[2][2];
    auto ld_w = [&](int ws, int P) {
        int l16 = lane16;
        asm volatile("" : "+v"(l16));
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int base = d * kWaves * WB + OFF_C + P * PC;
            wh[ws][d][0] = buf_load(wrs, l16, base + (0 << 10));
            wh[ws][d][1] = buf_load(wrs, l16, base + (1 << 10));
            wb0[ws][d] = buf_load(wrs, l16, base + (2 << 10));
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(wrs, l16 >> 1, base + (3 << 10), 0);
            wb1[ws][d] = make_uint2(v[0], v[1]);
            wsc[ws][d] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, l16 >> 2, base + (3 << 10) + 512, 0);
        }
    };

    f32x16 acc[2][kNxBT];
    auto init_acc = [&]() {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const float* bp = bias + ((size_t)(d * kWaves + wave) * 4 + 2) * 32 + hh * 16;
            f32x16 b;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(bp + 4 * q);
                b[4 * q + 0] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int bt = 0; bt < kNxBT; ++bt) acc[d][bt] = b;
        }
    };
    auto store_block = [&](int k) {                         // the finished block k (local index): 2 directions x 4 B tiles x 4 KiB of this wave
        const int id = (int)blockIdx.x + k * (int)gridDim.x;
        const int tp = id / kNxTQ, tq = id - tp * kNxTQ;
#pragma unroll
        for (int bt = 0; bt < kNxBT; ++bt) {
            const int ti = bt >> 1, t = 2 * tq + (bt & 1);
            if (t >= kSeqLen) continue;                     // (the duplicate of t = 20)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                uint4* o = gout + ((((size_t)(2 * tp + ti) * kSeqLen + t) * 2 + d) * kWaves + wave) * 4 * kFragU4 + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    // (by value through f2u: __builtin_bit_cast applied to a vector-element lvalue reads element 0 whatever the subscript -
                    // clang 19 / ROCm 7.2, seen in the ISA as four stores of one broadcast register)
                    nt_store(make_uint4(f2u(acc[d][bt][4 * q + 0]), f2u(acc[d][bt][4 * q + 1]), f2u(acc[d][bt][4 * q + 2]), f2u(acc[d][bt][4 * q + 3])), o + q * kFragU4);
            }
        }
    };

    // ---- prologue: the ring's first pairs, the first two weight slots
#pragma unroll
    for (int c = 0; c < kNxRS; ++c) dma_cons(c);             // (n_cons >= 16)
    ld_w(0, 0);
    ld_w(1, 1);
    init_acc();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    uint4 xh[kNxBT], xh1[kNxBT], xc0[kNxBT];
    uint2 xc1[kNxBT];
    auto rd_hi = [&](uint4 (&x)[kNxBT], int so, int kbl) {
#pragma unroll
        for (int bt = 0; bt < kNxBT; ++bt) x[bt] = *reinterpret_cast<const uint4*>(smem + so + (((kbl * kNxBT + bt) * 2 + 0) << 10));
    };
    auto rd_blob = [&](int so) {
#pragma unroll
        for (int bt = 0; bt < kNxBT; ++bt) {
            xc0[bt] = *reinterpret_cast<const uint4*>(smem + so + (((0 * kNxBT + bt) * 2 + 1) << 10));
            xc1[bt] = *reinterpret_cast<const uint2*>(smem + so + (((1 * kNxBT + bt) * 2 + 1) << 10));
        }
    };
#define CCSM_NX_FENCE asm volatile("" ::: "memory")
    auto main_kb = [&](int ws, int kbl, const uint4 (&x)[kNxBT]) {
        CCSM_NX_FENCE;
#pragma unroll
        for (int bt = 0; bt < kNxBT; ++bt)
#pragma unroll
            for (int d = 0; d < 2; ++d) acc[d][bt] = mfma16(wh[ws][d][kbl], x[bt], acc[d][bt]);
        CCSM_NX_FENCE;
    };
    auto corr = [&](int ws) {
        CCSM_NX_FENCE;
#pragma unroll
        for (int bt = 0; bt < kNxBT; ++bt)
#pragma unroll
            for (int d = 0; d < 2; ++d) acc[d][bt] = mfma_corr_mx6<0>(wb0[ws][d], wb1[ws][d], wsc[ws][d], xc0[bt], xc1[bt], acc[d][bt], sb);
        CCSM_NX_FENCE;
    };

    // One iteration = two consumptions.  Vector-memory operations retire in order, so a weight request issued behind a transfer cannot be
    // used before that transfer has landed: the iteration's transfers are its LAST requests (behind both weight reloads), which gives every
    // younger weight request 1.5 - 2 iterations (~5 k cycles) before its first use.  Every request is unconditional (reloads wrap around
    // the stream, transfers repeat the last consumption), so the counted wait below holds in every iteration.
    int slot = 0;                                           // ring slot of consumption a
    for (int a = 0; a < n_cons; a += 2) {
        const int slot_b = slot == kNxRS - 1 ? 0 : slot + 1;
        const int so_a = slot * kNxSlotBytes + lane16, so_b = slot_b * kNxSlotBytes + lane16;
        // ---- consumption a (weight slot 0)
        rd_hi(xh, so_a, 0);
        rd_hi(xh1, so_a, 1);
        main_kb(0, 0, xh);
        rd_blob(so_a);
        main_kb(0, 1, xh1);
        rd_hi(xh, so_b, 0);
        corr(0);
        ld_w(0, (a + 2) & 15);
        CCSM_NX_FENCE;
        // ---- consumption a + 1 (weight slot 1): main products, the barrier, then its correction product (operands in registers)
        rd_hi(xh1, so_b, 1);
        main_kb(1, 0, xh);
        rd_blob(so_b);
        main_kb(1, 1, xh1);
        // this wave's parts of consumptions a + 2, a + 3 (issued at the end of the iteration before the previous one) have landed: since
        // then 24 requests of the previous iteration and 10 of this one (block ends add stores and bias loads: stricter, still right)
        asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
        __syncthreads();                                    // consumptions a + 2, a + 3 are in LDS; every wave has read a and a + 1
        corr(1);
        if ((a & 15) == 14) {                               // the block's last pair is done
            store_block(a >> 4);
            init_acc();
        }
        CCSM_NX_FENCE;
        ld_w(1, (a + 3) & 15);
        CCSM_NX_FENCE;
        dma_cons(a + kNxRS);                                // the two slots just vacated
        dma_cons(a + kNxRS + 1);
        CCSM_NX_FENCE;
        slot = slot_b == kNxRS - 1 ? 0 : slot_b + 1;
    }
#undef CCSM_NX_FENCE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace ccsm
