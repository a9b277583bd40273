// VERDICT r04 item 2 ("build the kernel your own budget points at, or kill it with a diagnostic build"): DESIGN 10.1 sizes a layer-1/2 GRU
// kernel with 128 rows per workgroup, ONE wave per SIMD and 64 hidden units per wave (a third more rows per weight fragment, half the
// B-operand LDS reads per MFMA) against the shipping 96 rows / two waves per SIMD / 32 units per wave.  This is the inner loop of both
// shapes in isolation - the same instruction mix as gru_layer12_mx_kernel's phases, nothing else of the kernel, results meaningless:
//   phase X (the kernel's phase A: input part of the r and z gates; 16 pairs of k-blocks per step)
//       per pair and wave: 2 kb x 2 gates x UT hi weight fragments + 2 x UT fp4 blobs + UT scale dwords from an L2-resident stream, three
//       pairs ahead (register slots); the pair's x_t fragments (2 kb x [hi | blob] x NB KiB) HBM -> LDS by LDS-DMA into a four-slot ring,
//       refilled right behind the pair's barrier, counted s_waitcnt; B operands read from the ring; 4 UT NB main MFMAs
//       (v_mfma_f32_32x32x16_f16) + 2 UT NB correction MFMAs (v_mfma_scale_f32_32x32x64_f8f6f4, fp4 x fp6); ONE barrier per pair
//   phase H (the kernel's phase B: recurrent part of r, z, n; 8 pairs per step)
//       3 gates, B operands from the state in LDS (no transfers, no barrier), one pair of weights resident, each k-block's fragments
//       refilled with the next pair's right behind their MFMAs
// Shapes:  S96 = 512 threads, UT 1, NB 3 (shipping)   S128 = 256 threads, UT 2, NB 4 (proposed)   S96w = 256 threads, UT 2, NB 3 (one
// wave per SIMD at the shipping row count: isolates the occupancy effect)   S64 = 512 threads, UT 1, NB 2 (the shipping 64-row form)
// One workgroup per CU; cycles per pair from the wave's own counter, ns per (row, pair) from HIP events; the ISA's register / scratch
// figures are printed by tools/isa_report.py on the object.   build + run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 rows128_phase.hip -o rows128_phase && ./rows128_phase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
template <int G>
__device__ __forceinline__ f32x16 mfma_mx(uint4 w, uint32_t ws, uint4 x0, uint2 x1, f32x16 c, int sb) {
    const i32x8 a = {(int)w.x, (int)w.y, (int)w.z, (int)w.w, 0, 0, 0, 0};
    const i32x8 b = {(int)x0.x, (int)x0.y, (int)x0.z, (int)x0.w, (int)x1.x, (int)x1.y, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4 /* fp4 */, 2 /* fp6 */, G, (int)ws, 0, sb);
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

constexpr int kPairsX = 16, kPairsH = 8, kRS = 4;

// ---------------------------------------------------------------------------------------------------------------------------------
// phase X
// DIAG (results meaningless, hazards ignored on purpose): 1 = no barrier (each wave still waits for its own transfers); 2 = no transfers either
// (x stays what the prologue put into the ring): where phase X's idle cycles come from
template <int NW, int UT, int NB, int G = 2, int DIAG = 0>
__global__ __launch_bounds__(NW * 64, 1) void phase_x(const uint4* __restrict__ xin, const uint4* __restrict__ wst, float* __restrict__ out,
                                                      unsigned long long* __restrict__ cyc, int steps) {
    constexpr int FR = 4 * NB;                          // fragments (1 KiB) of one x pair: [kbl 2][bt NB][hi | blob]
    constexpr int SLOT = FR * 1024;
    constexpr int PW = (2 * G * UT + G * UT) * 1024 + UT * 256;    // weight bytes of one pair and wave
    constexpr int WOPS = 2 * G * UT + G * UT + UT;      // vector-memory requests of one weight slot
    constexpr int DHI = (FR + NW - 1) / NW, DLO = FR / NW, NHIW = FR - DLO * NW;     // transfers per wave: waves < NHIW move DHI, the others DLO
    constexpr int NSA = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane16 = lane * 16;
    const size_t x_rows = (size_t)kPairsX * SLOT;       // one step's x_t of this workgroup
    const u32x4_t xrs = dma_rsrc(reinterpret_cast<const char*>(xin) + (size_t)blockIdx.x * 21 * x_rows);
    const unsigned sx_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const char*>(wst) + (size_t)((blockIdx.x & 1) * NW + wave) * kPairsX * PW);
    auto dma_pair = [&](int slot, int s, int p) {
        const int base = ((s % 21) * kPairsX + p) * SLOT;
#pragma unroll
        for (int i = 0; i < DHI; ++i) {
            const int f = wave + i * NW;
            if (f < FR) dma16_buf(xrs, lane16, __builtin_amdgcn_readfirstlane(base + (f << 10)), __builtin_amdgcn_readfirstlane((int)(sx_base + slot * SLOT + (f << 10))));
        }
    };
    uint4 wh[NSA][2][G][UT], wb[NSA][G][UT];
    uint32_t wsc[NSA][UT];
    auto ld_slot = [&](int ws, int p) {
#pragma unroll
        for (int u = 0; u < UT; ++u) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int g = 0; g < G; ++g) wh[ws][k][g][u] = buf_load(wrs, lane16, p * PW + (((u * 2 + k) * G + g) << 10));
#pragma unroll
            for (int g = 0; g < G; ++g) wb[ws][g][u] = buf_load(wrs, lane16, p * PW + ((2 * G * UT + u * G + g) << 10));
            wsc[ws][u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, lane * 4, p * PW + ((3 * G * UT) << 10) + u * 256, 0);
        }
    };
#pragma unroll
    for (int g = 0; g < kRS; ++g) dma_pair(g, 0, g);
    ld_slot(0, 0); ld_slot(1, 1); ld_slot(2, 2);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSA * WOPS) : "memory");
    __syncthreads();
    f32x16 acc[G][UT][NB];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][u][b][r] = 0.f;
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
                for (int u = 0; u < UT; ++u)
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[g][u][b] = mfma16(wh[WS][0][g][u], xh[b], acc[g][u][b]);
            FENCE;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                xc0[b] = *reinterpret_cast<const uint4*>(smem + xs + (((0 * NB + b) * 2 + 1) << 10));
                xc1[b] = *reinterpret_cast<const uint2*>(smem + xs + (((1 * NB + b) * 2 + 1) << 10));
            }
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int u = 0; u < UT; ++u)
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[g][u][b] = mfma16(wh[WS][1][g][u], xh1[b], acc[g][u][b]);
            FENCE;
            // this wave's part of the next pair's transfer has landed: younger operations = (RS - 1) weight slots + (RS - 2) refills
            if constexpr (DIAG < 2) {
                if (wave < NHIW) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((kRS - 1) * WOPS + (kRS - 2) * DHI) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((kRS - 1) * WOPS + (kRS - 2) * DLO) : "memory");
            }
            if constexpr (DIAG == 0) __syncthreads();
            if constexpr (DIAG < 2) dma_pair(slot, s + (P + kRS) / kPairsX, (P + kRS) % kPairsX);
            FENCE;
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int u = 0; u < UT; ++u) {
                    acc[0][u][b] = mfma_mx<0>(wb[WS][0][u], wsc[WS][u], xc0[b], xc1[b], acc[0][u][b], 125);
                    if constexpr (G > 1) acc[1][u][b] = mfma_mx<1>(wb[WS][1][u], wsc[WS][u], xc0[b], xc1[b], acc[1][u][b], 125);
                    if constexpr (G > 2) acc[2][u][b] = mfma_mx<2>(wb[WS][2][u], wsc[WS][u], xc0[b], xc1[b], acc[2][u][b], 125);
                }
            FENCE;
            ld_slot(WS, (P + NSA) % kPairsX);           // the slot just consumed: three pairs ahead
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
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[g][u][b][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// phase H: one unit tile of the wave at a time (UT passes over the state): with both unit tiles' weights resident the 64-unit shape needs
// 384 accumulator + 74 weight + 40 operand registers and the compiler spills 2840 of them (first version of this file); per unit tile the
// resident weights are the shipping kernel's 37 registers - at the price of reading the B operands once per unit tile, i.e. the
// "half the B-operand LDS reads" of the proposal do not exist in this phase
template <int NW, int UT, int NB>
__global__ __launch_bounds__(NW * 64, 1) void phase_h(const uint4* __restrict__ wst, float* __restrict__ out, unsigned long long* __restrict__ cyc, int steps) {
    constexpr int PW = (6 + 3) * 1024 + 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];          // state: [kb 16][bt NB][hi | blob] x 1 KiB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane16 = lane * 16;
    for (int i = threadIdx.x; i < 16 * NB * 2 * 64; i += blockDim.x) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0x3c003c00u + i, 0x38003800u, 0x34003400u + (i >> 6), 0x30003000u);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const char*>(wst) + (size_t)((blockIdx.x & 1) * NW + wave) * UT * kPairsH * PW);
    uint4 wh[2][3], wb[3];
    uint32_t wsc;
    auto ld_h = [&](int k, int p) {             // p: pair index over the wave's UT x 8 pairs
#pragma unroll
        for (int g = 0; g < 3; ++g) wh[k][g] = buf_load(wrs, lane16, p * PW + ((k * 3 + g) << 10));
    };
    auto ld_b = [&](int p) {
#pragma unroll
        for (int g = 0; g < 3; ++g) wb[g] = buf_load(wrs, lane16, p * PW + ((6 + g) << 10));
        wsc = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, lane * 4, p * PW + (9 << 10), 0);
    };
    ld_h(0, 0); ld_h(1, 0); ld_b(0);
    f32x16 acc[3][UT][NB];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][u][b][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        uint4 xh[NB], xc0[NB];
        uint2 xc1[NB];
        static_for<0, UT * kPairsH>([&](auto PC_) {
            constexpr int Q = decltype(PC_)::value;
            constexpr int U = Q / kPairsH, P = Q % kPairsH;
            constexpr int QN = (Q + 1) % (UT * kPairsH);
            static_for<0, 2>([&](auto KC) {
                constexpr int K = decltype(KC)::value;
#pragma unroll
                for (int b = 0; b < NB; ++b) xh[b] = *reinterpret_cast<const uint4*>(smem + lane16 + ((((2 * P + K) * NB + b) * 2 + 0) << 10));
                FENCE;
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g][U][b] = mfma16(wh[K][g], xh[b], acc[g][U][b]);
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
                acc[0][U][b] = mfma_mx<0>(wb[0], wsc, xc0[b], xc1[b], acc[0][U][b], 125);
                acc[1][U][b] = mfma_mx<1>(wb[1], wsc, xc0[b], xc1[b], acc[1][U][b], 125);
                acc[2][U][b] = mfma_mx<2>(wb[2], wsc, xc0[b], xc1[b], acc[2][U][b], 125);
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
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[g][u][b][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); std::exit(1); } } while (0)

static uint4 *g_x, *g_w;
static float* g_out;
static unsigned long long* g_cyc;
static double g_ref[2] = {0.0, 0.0};        // S96's ns per (row, pair), phase X / H

template <int NW, int UT, int NB>
void run(const char* name, int n_cu, int steps) {
    const int rows = 32 * NB;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int phase = 0; phase < 2; ++phase) {
        const int lds = phase == 0 ? kRS * 4 * NB * 1024 : 16 * NB * 2 * 1024;
        const void* fn = phase == 0 ? reinterpret_cast<const void*>(&phase_x<NW, UT, NB>) : reinterpret_cast<const void*>(&phase_h<NW, UT, NB>);
        CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (phase == 0) hipLaunchKernelGGL((phase_x<NW, UT, NB>), dim3(n_cu), dim3(NW * 64), lds, 0, g_x, g_w, g_out, g_cyc, steps);
            else hipLaunchKernelGGL((phase_h<NW, UT, NB>), dim3(n_cu), dim3(NW * 64), lds, 0, g_w, g_out, g_cyc, steps);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        unsigned long long c[8] = {};
        CK(hipMemcpy(c, g_cyc, sizeof(c), hipMemcpyDeviceToHost));
        const int pairs = phase == 0 ? kPairsX : kPairsH;
        const int gates = phase == 0 ? 2 : 3;
        const double mfma = (4.0 + 2.0) * gates / 2.0 * UT * NB;            // per pair and wave: 2 kb x gates main + gates corr
        const double cyc_pair = (double)c[0] / ((double)steps * pairs);
        const double ns_row_pair = ms * 1e6 / ((double)steps * pairs * rows);
        if (g_ref[phase] == 0.0) g_ref[phase] = ns_row_pair;
        const double waves_per_simd = NW / 4.0;
        std::printf("%-5s phase %s: %3d rows, %d waves x %d units: %8.0f cycles per pair (wave 0; %5.1f MFMAs per pair and wave = %4.0f %% of the SIMD's MFMA time), "
                    "%7.3f ns per (row, pair) = %5.3f of S96\n", name, phase == 0 ? "X" : "H", rows, NW, 32 * UT, cyc_pair, mfma,
                    100.0 * mfma * waves_per_simd * 32.0 / cyc_pair, ns_row_pair, ns_row_pair / g_ref[phase]);
        std::fflush(stdout);
    }
}

// the input part in one pass (three gates per operand read, as the recurrent phase has them) against the shipping two passes (r, z | n)
template <int NW, int UT, int NB, int G, int DIAG = 0>
void run_x(const char* name, int n_cu, int steps) {
    const int rows = 32 * NB, lds = kRS * 4 * NB * 1024;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&phase_x<NW, UT, NB, G, DIAG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((phase_x<NW, UT, NB, G, DIAG>), dim3(n_cu), dim3(NW * 64), lds, 0, g_x, g_w, g_out, g_cyc, steps);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned long long c[8] = {};
    CK(hipMemcpy(c, g_cyc, sizeof(c), hipMemcpyDeviceToHost));
    const double mfma = 3.0 * G * UT * NB, cyc_pair = (double)c[0] / ((double)steps * kPairsX);
    std::printf("%-18s input part, %d gate(s) per pass: %3d rows: %8.0f cycles per pair (%5.1f MFMAs per pair and wave = %4.0f %% of the SIMD's MFMA time), %7.3f ns per (row, pair)\n",
                name, G, rows, cyc_pair, mfma, 100.0 * mfma * (NW / 4.0) * 32.0 / cyc_pair, ms * 1e6 / ((double)steps * kPairsX * rows));
    std::fflush(stdout);
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? std::atoi(argv[1]) : 2000;
    int dev = 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int n_cu = prop.multiProcessorCount;
    // x: per workgroup 21 steps x 16 pairs x 16 KiB (128 rows) = 5.25 MiB -> 1.3 GiB for 256 workgroups: from HBM, like the kernel's layer input
    const size_t xbytes = (size_t)n_cu * 21 * kPairsX * 16 * 1024, wbytes = (size_t)2 * 8 * 16 * 20 * 1024;
    CK(hipMalloc(&g_x, xbytes)); CK(hipMalloc(&g_w, wbytes)); CK(hipMalloc(&g_out, (size_t)n_cu * 512 * 4)); CK(hipMalloc(&g_cyc, 64));
    std::vector<uint32_t> h(wbytes / 4);
    uint32_t sd = 12345u;
    for (auto& v : h) { sd = sd * 1664525u + 1013904223u; v = (sd & 0x83ff83ffu) | 0x30003000u; }       // fp16 values of ordinary size
    CK(hipMemcpy(g_w, h.data(), wbytes, hipMemcpyHostToDevice));
    std::vector<uint32_t> hx(xbytes / 4);
    for (auto& v : hx) { sd = sd * 1664525u + 1013904223u; v = (sd & 0x83ff83ffu) | 0x30003000u; }
    CK(hipMemcpy(g_x, hx.data(), xbytes, hipMemcpyHostToDevice));
    std::printf("# rows128_phase: %d CUs, %d steps; one workgroup per CU\n", n_cu, steps);
    run<8, 1, 3>("S96", n_cu, steps);
    run<4, 2, 4>("S128", n_cu, steps);
    run<4, 2, 3>("S96w", n_cu, steps);
    run<8, 1, 2>("S64", n_cu, steps);
    std::printf("# one pass over x_t (r, z, n together: a fourth accumulator set, which fits at 64 rows only) against the shipping two passes\n");
    run_x<8, 1, 3, 2>("S96 rz", n_cu, steps);
    run_x<8, 1, 3, 1>("S96 n", n_cu, steps);
    run_x<8, 1, 2, 3>("S64 rzn", n_cu, steps);
    run_x<8, 1, 2, 2>("S64 rz", n_cu, steps);
    run_x<8, 1, 2, 1>("S64 n", n_cu, steps);
    std::printf("# where the input-part phase's idle cycles come from (diagnostic variants, hazards ignored): no barrier / no barrier and no transfers\n");
    run_x<8, 1, 3, 2, 1>("S96 rz nobar", n_cu, steps);
    run_x<8, 1, 3, 2, 2>("S96 rz nobar nodma", n_cu, steps);
    run_x<8, 1, 3, 1, 1>("S96 n nobar", n_cu, steps);
    run_x<8, 1, 3, 1, 2>("S96 n nobar nodma", n_cu, steps);
    return 0;
}
