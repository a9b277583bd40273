// Power-capped MFMA ceiling probe (diagnostics: ccsm_measure_mfma_ceiling in include/ccsm.h; tools/ubench/mfma_power_ceiling.hip is
// the stand-alone sweep over the same kernel).
//
// bench.py prices gru_layer12_mx_kernel against the 2.5 PFLOP/s dense fp16 peak, an unthrottled figure; under that kernel the package
// sits at its power cap with sclk at 1.64 of 2.4 GHz.  This loop is the kernel's matrix work and nothing else: every wave keeps its
// operands in registers (random fp16 values / random fp6, fp4 codes, a different A and B register set for consecutive instructions,
// so that the operand buses toggle as they do on real data) and issues
//   mode 0 "f16": v_mfma_f32_32x32x16_f16 only
//   mode 1 "mix": the product kernel's 64 : 35 issue mix - per two f16 MFMAs one v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 A x fp6 B)
//   mode 2 "lds": mix, with every B operand re-read from LDS (ds_read_b128) as the kernel's activations are
// on 4 independent accumulators, one workgroup per CU.  Rates are counted in fp16-MFMA flops only (the corrections are overhead, as
// in bench.py's `roofline.achieved`).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <vector>

namespace ccsm_ceiling {
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mf(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
// A: fp4 e2m1 (cbsz 4: the first 4 dwords are used), B: fp6 e2m3 (blgp 2: 6 dwords), block scales 2^-4
__device__ __forceinline__ f32x16 mc(uint4 a0, uint4 b0, uint2 b1, f32x16 c) {
    const i32x8 a = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, 0, 0, 0, 0};
    const i32x8 b = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 2, 0, 123, 0, 123);
}

template <int T, int MODE>   // MODE 0: f16, 1: mix, 2: mix with B from LDS
__global__ __launch_bounds__(T) void k(const uint4* __restrict__ rnd, float* out, int iters) {
    __shared__ uint4 lds[4 * 2 * 64];     // 4 B fragments (hi, blob) per wave slot: every wave reads the same 8 KiB like the kernel's 96-row state
    const int lane = threadIdx.x & 63, tid = blockIdx.x * T + threadIdx.x;
    uint4 w[6], x[4], xb[4];
    for (int i = 0; i < 6; ++i) w[i] = rnd[(tid * 14 + i) & 0xffff];
    for (int i = 0; i < 4; ++i) { x[i] = rnd[(tid * 14 + 6 + i) & 0xffff]; xb[i] = rnd[(tid * 14 + 10 + i) & 0xffff]; }
    if (threadIdx.x < 64) for (int i = 0; i < 4; ++i) { lds[(2 * i) * 64 + lane] = x[i]; lds[(2 * i + 1) * 64 + lane] = xb[i]; }
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {      // the weights alternate between two register sets
            asm volatile("" ::: "memory");
            if (MODE == 2) {
#pragma unroll
                for (int t = 0; t < 4; ++t) { x[t] = lds[(2 * t) * 64 + lane]; xb[t] = lds[(2 * t + 1) * 64 + lane]; }
            }
            // one "pair of k-blocks" for four 32-row tiles: 2 x 4 main products, 4 correction products
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mf(w[half * 3 + 0], x[t], acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mf(w[half * 3 + 1], x[(t + 1) & 3], acc[t]);
            if (MODE != 0) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = mc(w[half * 3 + 2], xb[t], make_uint2(xb[(t + 2) & 3].x, xb[(t + 2) & 3].y), acc[t]);
            }
            asm volatile("" ::: "memory");
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[tid] = s;
}

// The same on the 16-wide instructions (round 5: tools/ubench/mfma_power_shapes.hip found them 15-17 % cheaper per flop at the cap):
//   MIX 0: v_mfma_f32_16x16x32_f16 only          MIX 1: per two of them one fp4 x fp6 v_mfma_scale_f32_16x16x128_f8f6f4 (the same MACs per product)
// 8 independent accumulator tiles; one iteration = the flops of k<T, MODE>'s iteration (8 x 32x32x16 = 16 x 16x16x32).
typedef float f32x4c __attribute__((ext_vector_type(4)));
template <int T, int MIX>
__global__ __launch_bounds__(T) void k16(const uint4* __restrict__ rnd, float* out, int iters) {
    const int tid = blockIdx.x * T + threadIdx.x;
    uint4 w[6], x[4], xb[4];
    for (int i = 0; i < 6; ++i) w[i] = rnd[(tid * 14 + i) & 0xffff];
    for (int i = 0; i < 4; ++i) { x[i] = rnd[(tid * 14 + 6 + i) & 0xffff]; xb[i] = rnd[(tid * 14 + 10 + i) & 0xffff]; }
    f32x4c acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    auto m16 = [](uint4 a, uint4 b, f32x4c c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0); };
    auto c16 = [](uint4 a0, uint4 b0, uint2 b1, f32x4c c) {
        const i32x8 a = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, 0, 0, 0, 0};
        const i32x8 b = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, 0, 0};
        return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 2, 0, 123, 0, 123);
    };
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = m16(w[half * 3 + 0], x[t & 3], acc[t]);
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = m16(w[half * 3 + 1], x[(t + 1) & 3], acc[t]);
            if (MIX != 0) {
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = c16(w[half * 3 + 2], xb[t & 3], make_uint2(xb[(t + 2) & 3].x, xb[(t + 2) & 3].y), acc[t]);
            }
            asm volatile("" ::: "memory");
        }
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    out[tid] = s;
}

inline std::vector<uint4> random_operands() {
    // fp16 values uniform in (-1, 1) scaled by 1/8 (GRU weights / activations are of that order); the same bit patterns serve as
    // random fp6 / fp4 codes for the block-scaled products (any code is a finite number in those formats)
    std::vector<uint4> v(1 << 16);
    unsigned long long s = 0x9e3779b97f4a7c15ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto h = [&]() -> unsigned {
        const float f = ((float)(next() >> 40) / (float)(1 << 24) * 2.f - 1.f) * 0.125f;
        _Float16 q = (_Float16)f; unsigned short b; memcpy(&b, &q, 2); return b;
    };
    for (auto& q : v) { q.x = h() | (h() << 16); q.y = h() | (h() << 16); q.z = h() | (h() << 16); q.w = h() | (h() << 16); }
    return v;
}

}  // namespace ccsm_ceiling
