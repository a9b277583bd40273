// The recurrent phase of the split-mx GRU kernels (gru_layer12_mx_kernel's phase B, nearly all of gru_layer0_mx_kernel's step) in isolation, on
// the 32-wide matrix instructions it ships with and on the 16-wide ones: what the round-5 power microbenchmark's +15 % for the instruction mix
// (tools/ubench/mfma_power_mix_shapes.hip: registers only) is worth once the operands come from LDS and the weights stream from L2.
//   ph32 : tools/ubench/rows128_phase.hip's phase H at the shipping shape (96 rows, 8 waves x 32 units): per pair of k-blocks and wave
//          2 x 3 hi weight fragments + 3 fp4 blobs + a scale dword, 18 v_mfma_f32_32x32x16_f16 + 9 v_mfma_scale_f32_32x32x64_f8f6f4,
//          B operands = the state's [hi | blob] fragments in LDS (lane * 16)
//   ph16 : the same products on v_mfma_f32_16x16x32_f16 + v_mfma_scale_f32_16x16x128_f8f6f4.  The scaled instruction contracts K = 128 = four
//          blocks of 32: the two correction terms of TWO pairs, so the unit of work is a double pair - 2 x 36 main products, then 36 correction
//          products whose A operand is (W_lo, W_hi) of both pairs (lane (m, q): q >> 1 = pair, q & 1 = term) and whose B operand is the two
//          pairs' activation blobs read with per-lane addresses (lane (n', q) reads row 16 h + n' of the 32-row blob fragment of pair q >> 1,
//          lane position n' + 32 (q & 1) + 16 h): no new activation format.  Weights in two register slots of six fragments in use order
//          (hi of pair 2D | hi of pair 2D + 1 | blobs of D), each refilled right behind its use (36 MFMAs of lead); accumulators tied by asm
//          statements (the register allocator moves 4-register accumulators around otherwise: profiles/r05_q).
//   px32 / px16 : the INPUT-part phase (gru_layer12_mx_kernel's phase A: r and z gates, 16 pairs per step; x_t HBM -> LDS by LDS-DMA into a
//          four-slot ring refilled right behind the pair's barrier, counted s_waitcnt, weights three pairs ahead in register slots) in the same two
//          forms.  px16 answers the design question of the 16-wide family: the correction product of pairs (P - 1, P) needs both pairs' blobs, and
//          the ring releases a slot at its pair's barrier - so the lanes that take pair P - 1's blob (q < 2: lanes 0-31) read it WHILE PAIR P - 1 IS
//          IN ITS SLOT and keep it in registers across the barrier; at pair P lanes 32-63 read theirs into the same registers (exec-masked
//          ds_reads), then the 24 correction instructions issue.  The ring, its refills and its waits stay as they are.
// Both: 256 workgroups of 512 threads, state and weights RANDOM (rows128_phase.hip's state is a constant pattern: fine for cycles, not for
// watts), 2000 steps of 8 pairs; prints cycles per pair (wave 0) and ns per (row, pair) from HIP events, three repetitions each, alternating.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/phase_h_shapes.hip -o tools/ubench/_build/phase_h_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x6_t __attribute__((ext_vector_type(6)));

__device__ __forceinline__ f32x16 mfma16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
template <int G>
__device__ __forceinline__ f32x16 mfma_mx(uint4 w, uint32_t ws, uint4 x0, uint2 x1, f32x16 c, int sb) {
    const i32x8 a = {(int)w.x, (int)w.y, (int)w.z, (int)w.w, 0, 0, 0, 0};
    const i32x8 b = {(int)x0.x, (int)x0.y, (int)x0.z, (int)x0.w, (int)x1.x, (int)x1.y, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4 /* fp4 */, 2 /* fp6 */, G, (int)ws, 0, sb);
}
// 16-wide instructions with the accumulator tied
__device__ __forceinline__ f32x4 m16(uint4 a, uint4 b, f32x4 c) {
    const u32x4_t av = __builtin_bit_cast(u32x4_t, a), bv = __builtin_bit_cast(u32x4_t, b);
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
    return c;
}
template <int BYTE>     // scale byte of the A operand's scale dword
__device__ __forceinline__ f32x4 c16(uint4 a, uint32_t sa, uint4 b0, uint2 b1, uint32_t sb, f32x4 c) {
    const u32x4_t av = __builtin_bit_cast(u32x4_t, a);
    const u32x6_t bv = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y};
    if constexpr (BYTE == 0) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:2" : "+v"(c) : "v"(av), "v"(bv), "v"(sa), "v"(sb));
    else if constexpr (BYTE == 1) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[1,0,0] op_sel_hi:[0,0,0] cbsz:4 blgp:2" : "+v"(c) : "v"(av), "v"(bv), "v"(sa), "v"(sb));
    else asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[1,0,0] cbsz:4 blgp:2" : "+v"(c) : "v"(av), "v"(bv), "v"(sa), "v"(sb));
    return c;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ u32x4_t dma_rsrc(const void* base) {
    const unsigned long long b = (unsigned long long)base;
    return u32x4_t{(unsigned)__builtin_amdgcn_readfirstlane((int)b), (unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) & 0xffffu, 0x7fffffffu, 0x00020000u};
}
__device__ __forceinline__ void dma16_buf(u32x4_t rsrc, int voff, int soff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base) : "memory");
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
#define FENCE asm volatile("" ::: "memory")

constexpr int kPairsH = 8, NB = 3;
constexpr int kStateBytes = 16 * NB * 2 * 1024;

__device__ __forceinline__ void fill_state(char* smem, const uint4* __restrict__ rnd) {
    for (int i = threadIdx.x; i < kStateBytes / 16; i += blockDim.x) reinterpret_cast<uint4*>(smem)[i] = rnd[(blockIdx.x * 977 + i) & 0xffff];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void ph32(const uint4* __restrict__ wst, const uint4* __restrict__ rnd, float* __restrict__ out,
                                               unsigned long long* __restrict__ cyc, int steps) {
    constexpr int PW = (6 + 3) * 1024 + 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];          // state: [kb 16][bt NB][hi | blob] x 1 KiB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane16 = lane * 16;
    fill_state(smem, rnd);
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const char*>(wst) + (size_t)((blockIdx.x & 1) * 8 + wave) * kPairsH * PW);
    uint4 wh[2][3], wb[3];
    uint32_t wsc;
    auto ld_h = [&](int k, int p) {
#pragma unroll
        for (int g = 0; g < 3; ++g) wh[k][g] = buf_load(wrs, lane16, p * PW + ((k * 3 + g) << 10));
    };
    auto ld_b = [&](int p) {
#pragma unroll
        for (int g = 0; g < 3; ++g) wb[g] = buf_load(wrs, lane16, p * PW + ((6 + g) << 10));
        wsc = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, lane * 4, p * PW + (9 << 10), 0);
    };
    ld_h(0, 0); ld_h(1, 0); ld_b(0);
    f32x16 acc[3][NB];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][b][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        uint4 xh[NB], xc0[NB];
        uint2 xc1[NB];
        static_for<0, kPairsH>([&](auto PC_) {
            constexpr int P = decltype(PC_)::value;
            constexpr int QN = (P + 1) % kPairsH;
            static_for<0, 2>([&](auto KC) {
                constexpr int K = decltype(KC)::value;
#pragma unroll
                for (int b = 0; b < NB; ++b) xh[b] = *reinterpret_cast<const uint4*>(smem + lane16 + ((((2 * P + K) * NB + b) * 2 + 0) << 10));
                FENCE;
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g][b] = mfma16(wh[K][g], xh[b], acc[g][b]);
                FENCE;
                ld_h(K, QN);
                FENCE;
            });
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                xc0[b] = *reinterpret_cast<const uint4*>(smem + lane16 + ((((2 * P) * NB + b) * 2 + 1) << 10));
                xc1[b] = *reinterpret_cast<const uint2*>(smem + lane16 + ((((2 * P + 1) * NB + b) * 2 + 1) << 10));
            }
            FENCE;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                acc[0][b] = mfma_mx<0>(wb[0], wsc, xc0[b], xc1[b], acc[0][b], 125);
                acc[1][b] = mfma_mx<1>(wb[1], wsc, xc0[b], xc1[b], acc[1][b], 125);
                acc[2][b] = mfma_mx<2>(wb[2], wsc, xc0[b], xc1[b], acc[2][b], 125);
            }
            FENCE;
            ld_b(QN);
            FENCE;
        });
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[g][b][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// weight stream per wave and double pair D: hi of pair 2D (T, g) at (3 T + g) KiB | hi of pair 2D + 1 at 6 + ... | blobs (T, g) at 12 + ... |
// scale dwords (T = 0: bytes 0-2 = gates; T = 1 likewise) at 18 KiB and 18 KiB + 256
__global__ __launch_bounds__(512, 1) void ph16(const uint4* __restrict__ wst, const uint4* __restrict__ rnd, float* __restrict__ out,
                                               unsigned long long* __restrict__ cyc, int steps) {
    constexpr int DW = 18 * 1024 + 512, ND = kPairsH / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane16 = lane * 16;
    fill_state(smem, rnd);
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const char*>(wst) + (size_t)((blockIdx.x & 1) * 8 + wave) * ND * DW);
    // this lane's byte offset inside the state's [kb][bt][hl] fragments: k-block (q >> 1) of a pair, lane position n' + 32 (q & 1)
    const int kbo = (lane & 32) << 6;
    const int lx = kbo + (kbo << 1) + ((((lane >> 4) & 1) * 32 + (lane & 15)) << 4);          // NB = 3
    // ... and of a double pair's blobs: pair (q >> 1), lane position n' + 32 (q & 1)
    const int lb = (lane >> 5) * (2 * NB * 2048) + ((((lane >> 4) & 1) * 32 + (lane & 15)) << 4);
    const uint32_t sbv = (lane & 16) ? 113u : 125u;                          // x_lo * 2^14 | x_hi * 4
    uint4 w[2][6];                                                            // two slots of six fragments, in use order
    uint32_t ws[2] = {0, 0};
    // use u of a step (12 per step): set u at D = u / 3, kind = u % 3 (0, 1: hi of pair 2D + kind; 2: blobs)
    auto ld_set = [&](auto UC) {
        constexpr int U = decltype(UC)::value % 12;
        constexpr int D = U / 3, KIND = U % 3, SL = U & 1;
#pragma unroll
        for (int i = 0; i < 6; ++i) w[SL][i] = buf_load(wrs, lane16, D * DW + ((KIND * 6 + i) << 10));
        if constexpr (KIND == 2) {
            ws[0] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, lane * 4, D * DW + (18 << 10), 0);
            ws[1] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, lane * 4, D * DW + (18 << 10) + 256, 0);
        }
    };
    ld_set(std::integral_constant<int, 0>{});
    ld_set(std::integral_constant<int, 1>{});
    f32x4 acc[3][2][NB][2];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[g][T][b][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        static_for<0, 12>([&](auto UC) {
            constexpr int U = decltype(UC)::value;
            constexpr int D = U / 3, KIND = U % 3, SL = U & 1;
            if constexpr (KIND < 2) {                       // main products of pair 2 D + KIND: row halves in turn
                constexpr int P = 2 * D + KIND;
                uint4 xh[NB];
                static_for<0, 2>([&](auto HC) {
                    constexpr int H = decltype(HC)::value;
#pragma unroll
                    for (int b = 0; b < NB; ++b) xh[b] = *reinterpret_cast<const uint4*>(smem + lx + ((((2 * P) * NB + b) * 2) << 10) + H * 256);
                    FENCE;
#pragma unroll
                    for (int T = 0; T < 2; ++T)
#pragma unroll
                        for (int b = 0; b < NB; ++b)
#pragma unroll
                            for (int g = 0; g < 3; ++g) acc[g][T][b][H] = m16(w[SL][3 * T + g], xh[b], acc[g][T][b][H]);
                    FENCE;
                });
            } else {                                        // correction products of the double pair
                uint4 xc0[NB];
                uint2 xc1[NB];
                static_for<0, 2>([&](auto HC) {
                    constexpr int H = decltype(HC)::value;
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        xc0[b] = *reinterpret_cast<const uint4*>(smem + lb + ((((4 * D) * NB + b) * 2 + 1) << 10) + H * 256);
                        xc1[b] = *reinterpret_cast<const uint2*>(smem + lb + ((((4 * D + 1) * NB + b) * 2 + 1) << 10) + H * 256);
                    }
                    FENCE;
#pragma unroll
                    for (int T = 0; T < 2; ++T)
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            acc[0][T][b][H] = c16<0>(w[SL][3 * T + 0], ws[T], xc0[b], xc1[b], sbv, acc[0][T][b][H]);
                            acc[1][T][b][H] = c16<1>(w[SL][3 * T + 1], ws[T], xc0[b], xc1[b], sbv, acc[1][T][b][H]);
                            acc[2][T][b][H] = c16<2>(w[SL][3 * T + 2], ws[T], xc0[b], xc1[b], sbv, acc[2][T][b][H]);
                        }
                    FENCE;
                });
            }
            ld_set(std::integral_constant<int, U + 2>{});      // the slot just used takes the set of two uses ahead
            FENCE;
        });
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15" ::: "memory");
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int h = 0; h < 2; ++h) sum += acc[g][T][b][h][0] + acc[g][T][b][h][1] + acc[g][T][b][h][2] + acc[g][T][b][h][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// input-part phase: rows128_phase.hip's phase_x<8, 1, 3, 2> (px32) and its 16-wide form (px16)
constexpr int kPairsX = 16, kRS = 4, FR = 4 * NB, SLOT = FR * 1024, NSA = 3;

__global__ __launch_bounds__(512, 1) void px32(const uint4* __restrict__ xin, const uint4* __restrict__ wst, float* __restrict__ out,
                                               unsigned long long* __restrict__ cyc, int steps) {
    constexpr int G = 2, PW = (2 * G + G) * 1024 + 256, WOPS = 2 * G + G + 1;
    constexpr int DHI = (FR + 7) / 8, DLO = FR / 8, NHIW = FR - DLO * 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane16 = lane * 16;
    const size_t x_rows = (size_t)kPairsX * SLOT;
    const u32x4_t xrs = dma_rsrc(reinterpret_cast<const char*>(xin) + (size_t)blockIdx.x * 21 * x_rows);
    const unsigned sx_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const char*>(wst) + (size_t)((blockIdx.x & 1) * 8 + wave) * kPairsX * PW);
    auto dma_pair = [&](int slot, int s, int p) {
        const int base = ((s % 21) * kPairsX + p) * SLOT;
#pragma unroll
        for (int i = 0; i < DHI; ++i) {
            const int f = wave + i * 8;
            if (f < FR) dma16_buf(xrs, lane16, __builtin_amdgcn_readfirstlane(base + (f << 10)), __builtin_amdgcn_readfirstlane((int)(sx_base + slot * SLOT + (f << 10))));
        }
    };
    uint4 wh[NSA][2][G], wb[NSA][G];
    uint32_t wsc[NSA];
    auto ld_slot = [&](int ws, int p) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int g = 0; g < G; ++g) wh[ws][k][g] = buf_load(wrs, lane16, p * PW + ((k * G + g) << 10));
#pragma unroll
        for (int g = 0; g < G; ++g) wb[ws][g] = buf_load(wrs, lane16, p * PW + ((2 * G + g) << 10));
        wsc[ws] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, lane * 4, p * PW + ((3 * G) << 10), 0);
    };
#pragma unroll
    for (int g = 0; g < kRS; ++g) dma_pair(g, 0, g);
    ld_slot(0, 0); ld_slot(1, 1); ld_slot(2, 2);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSA * WOPS) : "memory");
    __syncthreads();
    f32x16 acc[G][NB];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][b][r] = 0.f;
    int slot = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        uint4 xh[NB], xh1[NB], xc0[NB];
        uint2 xc1[NB];
        static_for<0, kPairsX>([&](auto PC_) {
            constexpr int P = decltype(PC_)::value;
            constexpr int WS = P % NSA;
            const int xs = slot * SLOT + lane16;
            const int slot_n = slot == kRS - 1 ? 0 : slot + 1;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                xh[b] = *reinterpret_cast<const uint4*>(smem + xs + (((0 * NB + b) * 2 + 0) << 10));
                xh1[b] = *reinterpret_cast<const uint4*>(smem + xs + (((1 * NB + b) * 2 + 0) << 10));
            }
            FENCE;
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g][b] = mfma16(wh[WS][0][g], xh[b], acc[g][b]);
            FENCE;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                xc0[b] = *reinterpret_cast<const uint4*>(smem + xs + (((0 * NB + b) * 2 + 1) << 10));
                xc1[b] = *reinterpret_cast<const uint2*>(smem + xs + (((1 * NB + b) * 2 + 1) << 10));
            }
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g][b] = mfma16(wh[WS][1][g], xh1[b], acc[g][b]);
            FENCE;
            if (wave < NHIW) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((kRS - 1) * WOPS + (kRS - 2) * DHI) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((kRS - 1) * WOPS + (kRS - 2) * DLO) : "memory");
            __syncthreads();
            dma_pair(slot, s + (P + kRS) / kPairsX, (P + kRS) % kPairsX);
            FENCE;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                acc[0][b] = mfma_mx<0>(wb[WS][0], wsc[WS], xc0[b], xc1[b], acc[0][b], 125);
                acc[1][b] = mfma_mx<1>(wb[WS][1], wsc[WS], xc0[b], xc1[b], acc[1][b], 125);
            }
            FENCE;
            ld_slot(WS, (P + NSA) % kPairsX);
            FENCE;
            slot = slot_n;
        });
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[g][b][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

// weight stream per wave and pair (uniform slots of 10 KiB + 512 B): hi (T, g) at (2 T + g) KiB | [odd pairs: the double pair's blobs (T, g) at
// 4 + (2 T + g) KiB | scale dwords of T = 0, 1 at 8 KiB, 8 KiB + 256]
__global__ __launch_bounds__(512, 1) void px16(const uint4* __restrict__ xin, const uint4* __restrict__ wst, float* __restrict__ out,
                                               unsigned long long* __restrict__ cyc, int steps) {
    constexpr int PW = 8 * 1024 + 512, WE = 4, WO = 10;              // vector-memory requests of a weight slot: even / odd pair
    constexpr int DHI = (FR + 7) / 8, DLO = FR / 8, NHIW = FR - DLO * 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane16 = lane * 16;
    const size_t x_rows = (size_t)kPairsX * SLOT;
    const u32x4_t xrs = dma_rsrc(reinterpret_cast<const char*>(xin) + (size_t)blockIdx.x * 21 * x_rows);
    const unsigned sx_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const char*>(wst) + (size_t)((blockIdx.x & 1) * 8 + wave) * kPairsX * PW);
    auto dma_pair = [&](int slot, int s, int p) {
        const int base = ((s % 21) * kPairsX + p) * SLOT;
#pragma unroll
        for (int i = 0; i < DHI; ++i) {
            const int f = wave + i * 8;
            if (f < FR) dma16_buf(xrs, lane16, __builtin_amdgcn_readfirstlane(base + (f << 10)), __builtin_amdgcn_readfirstlane((int)(sx_base + slot * SLOT + (f << 10))));
        }
    };
    // per-lane offsets inside a ring slot ([kbl][bt][hl] fragments): k-block q >> 1, lane position n' + 32 (q & 1); and for a blob: lane position only
    const int lpos = (((lane >> 4) & 1) * 32 + (lane & 15)) << 4;
    const int kbo = (lane & 32) << 6;
    const int lxs = lpos + kbo + (kbo << 1);                         // NB = 3
    const uint32_t sbv = (lane & 16) ? 113u : 125u;
    uint4 wh[2][4], wb[2][4];                                        // hi (T, g) of TWO pairs (three pairs of lead as in px32 spill: 256 registers + 40 B of scratch); blobs of two double pairs
    uint32_t wsc[2][2];
    auto ld_slot = [&](auto PC_) {                                  // pair p (static): hi into slot p % 3; odd p: the double pair's blobs into slot (p >> 1) & 1
        constexpr int p = decltype(PC_)::value % kPairsX;
#pragma unroll
        for (int i = 0; i < 4; ++i) wh[p & 1][i] = buf_load(wrs, lane16, p * PW + (i << 10));
        if constexpr (p & 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wb[(p >> 1) & 1][i] = buf_load(wrs, lane16, p * PW + ((4 + i) << 10));
            wsc[(p >> 1) & 1][0] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, lane * 4, p * PW + (8 << 10), 0);
            wsc[(p >> 1) & 1][1] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, lane * 4, p * PW + (8 << 10) + 256, 0);
        }
    };
#pragma unroll
    for (int g = 0; g < kRS; ++g) dma_pair(g, 0, g);
    ld_slot(std::integral_constant<int, 0>{}); ld_slot(std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WE + WO) : "memory");
    __syncthreads();
    f32x4 acc[2][2][NB][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[g][T][b][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 xc0[2][NB];                                               // the double pair's blob operands: lanes 0-31 pair P - 1, lanes 32-63 pair P
    uint2 xc1[2][NB];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int b = 0; b < NB; ++b) { xc0[h][b] = make_uint4(0, 0, 0, 0); xc1[h][b] = make_uint2(0, 0); }
    int slot = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        static_for<0, kPairsX>([&](auto PC_) {
            constexpr int P = decltype(PC_)::value;
            constexpr int WS = P & 1, BS = (P >> 1) & 1;
            const int xs = slot * SLOT;
            const int slot_n = slot == kRS - 1 ? 0 : slot + 1;
            uint4 xh[2][NB];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int b = 0; b < NB; ++b) xh[h][b] = *reinterpret_cast<const uint4*>(smem + xs + lxs + ((b * 2) << 10) + h * 256);
            // this pair's blob, for the lanes that take it: lanes 0-31 at an even pair (kept across the barrier), lanes 32-63 at an odd one
            if ((lane >> 5) == (P & 1)) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        xc0[h][b] = *reinterpret_cast<const uint4*>(smem + xs + lpos + ((b * 2 + 1) << 10) + h * 256);
                        xc1[h][b] = *reinterpret_cast<const uint2*>(smem + xs + lpos + (((NB + b) * 2 + 1) << 10) + h * 256);
                    }
            }
            FENCE;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int T = 0; T < 2; ++T)
#pragma unroll
                    for (int b = 0; b < NB; ++b)
#pragma unroll
                        for (int g = 0; g < 2; ++g) acc[g][T][b][h] = m16(wh[WS][2 * T + g], xh[h][b], acc[g][T][b][h]);
            FENCE;
            // younger operations than this wave's part of the next pair's transfer (issued behind the barrier of pair P - 3): the weight slots
            // requested behind pairs P - 3, P - 2, P - 1 (for pairs P - 1, P, P + 1) and two refills
            {
                constexpr int W3 = (P & 1) ? WE + WO + WE : WO + WE + WO;
                if (wave < NHIW) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W3 + (kRS - 2) * DHI) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W3 + (kRS - 2) * DLO) : "memory");
            }
            __syncthreads();
            dma_pair(slot, s + (P + kRS) / kPairsX, (P + kRS) % kPairsX);
            FENCE;
            if constexpr (P & 1) {                                  // the double pair's correction products
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int T = 0; T < 2; ++T)
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            acc[0][T][b][h] = c16<0>(wb[BS][2 * T + 0], wsc[BS][T], xc0[h][b], xc1[h][b], sbv, acc[0][T][b][h]);
                            acc[1][T][b][h] = c16<1>(wb[BS][2 * T + 1], wsc[BS][T], xc0[h][b], xc1[h][b], sbv, acc[1][T][b][h]);
                        }
                FENCE;
            }
            ld_slot(std::integral_constant<int, P + 2>{});         // the slot just consumed: two pairs ahead
            FENCE;
            slot = slot_n;
        });
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15" ::: "memory");
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int h = 0; h < 2; ++h) sum += acc[g][T][b][h][0] + acc[g][T][b][h][1] + acc[g][T][b][h][2] + acc[g][T][b][h][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); std::exit(1); } } while (0)

int main(int argc, char** argv) {
    const int steps = argc > 1 ? std::atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    uint4 *g_w, *g_rnd;
    float* g_out;
    unsigned long long* g_cyc;
    const size_t wbytes = (size_t)2 * 8 * 8 * 20 * 1024;
    CK(hipMalloc(&g_w, wbytes)); CK(hipMalloc(&g_rnd, 65536 * 16)); CK(hipMalloc(&g_out, (size_t)n_cu * 512 * 4)); CK(hipMalloc(&g_cyc, 64));
    std::vector<uint32_t> h(wbytes / 4);
    uint32_t sd = 12345u;
    for (auto& v : h) { sd = sd * 1664525u + 1013904223u; v = (sd & 0x83ff83ffu) | 0x30003000u; }       // fp16 values of ordinary size
    CK(hipMemcpy(g_w, h.data(), wbytes, hipMemcpyHostToDevice));
    std::vector<uint32_t> hr(65536 * 4);
    for (auto& v : hr) { sd = sd * 1664525u + 1013904223u; v = (sd & 0x83ff83ffu) | 0x30003000u; }
    CK(hipMemcpy(g_rnd, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    uint4* g_x;
    const size_t xbytes = (size_t)n_cu * 21 * kPairsX * SLOT;
    CK(hipMalloc(&g_x, xbytes));
    {
        std::vector<uint32_t> hx(xbytes / 4);
        for (auto& v : hx) { sd = sd * 1664525u + 1013904223u; v = (sd & 0x83ff83ffu) | 0x30003000u; }
        CK(hipMemcpy(g_x, hx.data(), xbytes, hipMemcpyHostToDevice));
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&px32), hipFuncAttributeMaxDynamicSharedMemorySize, kRS * SLOT));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&px16), hipFuncAttributeMaxDynamicSharedMemorySize, kRS * SLOT));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ph32), hipFuncAttributeMaxDynamicSharedMemorySize, kStateBytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ph16), hipFuncAttributeMaxDynamicSharedMemorySize, kStateBytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::printf("# phase_h_shapes: %d CUs, %d steps of 8 pairs, 96 rows per workgroup, random state and weights\n", n_cu, steps);
    for (int rep = 0; rep < 4; ++rep)
        for (int shape = 0; shape < 2; ++shape) {
            float ms = 0.f;
            CK(hipEventRecord(e0, 0));
            if (shape == 0) hipLaunchKernelGGL(ph32, dim3(n_cu), dim3(512), kStateBytes, 0, g_w, g_rnd, g_out, g_cyc, steps);
            else hipLaunchKernelGGL(ph16, dim3(n_cu), dim3(512), kStateBytes, 0, g_w, g_rnd, g_out, g_cyc, steps);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long c[8] = {};
            CK(hipMemcpy(c, g_cyc, sizeof(c), hipMemcpyDeviceToHost));
            const double cyc_pair = (double)c[0] / ((double)steps * kPairsH);
            std::printf("%s run %d: %8.0f cycles per pair (wave 0; MFMA time of the SIMD's two waves: 1728), %7.3f ns per (row, pair), %.1f ms\n",
                        shape == 0 ? "ph32 (32x32x16 + 32x32x64) " : "ph16 (16x16x32 + 16x16x128)", rep, cyc_pair, ms * 1e6 / ((double)steps * kPairsH * 96), ms);
            std::fflush(stdout);
        }
    std::printf("# input-part phase (r, z gates): 16 pairs per step, x_t by LDS-DMA through a four-slot ring, one barrier per pair\n");
    for (int rep = 0; rep < 4; ++rep)
        for (int shape = 0; shape < 2; ++shape) {
            float ms = 0.f;
            CK(hipEventRecord(e0, 0));
            if (shape == 0) hipLaunchKernelGGL(px32, dim3(n_cu), dim3(512), kRS * SLOT, 0, g_x, g_w, g_out, g_cyc, steps);
            else hipLaunchKernelGGL(px16, dim3(n_cu), dim3(512), kRS * SLOT, 0, g_x, g_w, g_out, g_cyc, steps);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long c[8] = {};
            CK(hipMemcpy(c, g_cyc, sizeof(c), hipMemcpyDeviceToHost));
            std::printf("%s run %d: %8.0f cycles per pair (wave 0; MFMA time of the SIMD's two waves: 1152), %7.3f ns per (row, pair), %.1f ms\n",
                        shape == 0 ? "px32 (32x32x16 + 32x32x64) " : "px16 (16x16x32 + 16x16x128)", rep, (double)c[0] / ((double)steps * kPairsX),
                        ms * 1e6 / ((double)steps * kPairsX * 96), ms);
            std::fflush(stdout);
        }
    return 0;
}
