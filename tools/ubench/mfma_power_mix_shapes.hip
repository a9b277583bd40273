// The split-mx instruction mix under the package power cap in both MFMA shapes (companion of mfma_power_shapes.hip, which found the
// 16x16x32 fp16 instruction 17 % faster at the cap than 32x32x16):
//   mix32 : per two v_mfma_f32_32x32x16_f16 one v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 A x fp6 B)      - what gru_layer12_mx_kernel issues
//   mix16 : per two v_mfma_f32_16x16x32_f16 one v_mfma_scale_f32_16x16x128_f8f6f4 (fp4 A x fp6 B)     - the same MACs per product
//   f16_32 / f16_16 : the fp16 instructions alone (reference points, as in mfma_power_shapes.hip)
// One 512-thread workgroup per CU, register-resident random operands, independent accumulators; rates in fp16-MFMA flops (the
// corrections are overhead, as in bench.py's roofline.achieved).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power_mix_shapes.hip -o tools/ubench/_build/mfma_power_mix_shapes
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mf32(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mf16(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
// A: fp4 e2m1 (cbsz 4: 4 dwords), B: fp6 e2m3 (blgp 2: 6 dwords), block scales 2^-4
__device__ __forceinline__ f32x16 mc32(uint4 a0, uint4 b0, uint2 b1, f32x16 c) {
    const i32x8 a = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, 0, 0, 0, 0};
    const i32x8 b = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 2, 0, 123, 0, 123);
}
__device__ __forceinline__ f32x4 mc16(uint4 a0, uint4 b0, uint2 b1, f32x4 c) {
    const i32x8 a = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, 0, 0, 0, 0};
    const i32x8 b = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 2, 0, 123, 0, 123);
}

// SHAPE 0: 32-wide, 1: 16-wide.  MIX 0: fp16 only, 1: two fp16 per scaled instruction
// MIX 2 (16-wide only): the scaled instruction's A operand is ZERO in lanes 32-63, i.e. K = 128 with the upper 64 empty, and it is issued once
// per 16x16x32 instead of once per two: the form in which the split-mx correction product of ONE pair of k-blocks (K = 64) fits the 16-wide
// instruction without a second pair's operands - twice the issue time of mix16's corrections for (if zeros are as cheap as they are for the
// fp16 instruction) the same energy
template <int SHAPE, int MIX>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ rnd, float* out, int iters) {
    const int tid = blockIdx.x * 512 + threadIdx.x;
    constexpr int NACC = SHAPE == 0 ? 4 : 16;
    uint4 w[8], x[8];
    for (int i = 0; i < 8; ++i) { w[i] = rnd[(tid * 16 + i) & 0xffff]; x[i] = rnd[(tid * 16 + 8 + i) & 0xffff]; }
    uint4 wz[8];
    for (int i = 0; i < 8; ++i) wz[i] = (threadIdx.x & 32) ? make_uint4(0, 0, 0, 0) : w[i];
    using acc_t = typename std::conditional<SHAPE == 0, f32x16, f32x4>::type;
    constexpr int NR = SHAPE == 0 ? 16 : 4;
    acc_t acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < NR; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
#pragma unroll
            for (int t = 0; t < NACC; ++t) {
                if constexpr (SHAPE == 0) acc[t] = mf32(w[q], x[(t + q) & 7], acc[t]); else acc[t] = mf16(w[q], x[(t + q) & 7], acc[t]);
            }
#pragma unroll
            for (int t = 0; t < NACC; ++t) {
                if constexpr (SHAPE == 0) acc[t] = mf32(w[q + 1], x[(t + q + 3) & 7], acc[t]); else acc[t] = mf16(w[q + 1], x[(t + q + 3) & 7], acc[t]);
            }
            if constexpr (MIX == 1) {
#pragma unroll
                for (int t = 0; t < NACC; ++t) {
                    const uint4 xa = x[(t + q + 5) & 7], xb = x[(t + q + 6) & 7];
                    if constexpr (SHAPE == 0) acc[t] = mc32(w[(q + 2) & 7], xa, make_uint2(xb.x, xb.y), acc[t]);
                    else acc[t] = mc16(w[(q + 2) & 7], xa, make_uint2(xb.x, xb.y), acc[t]);
                }
            }
            if constexpr (MIX == 2) {
#pragma unroll
                for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                    for (int t = 0; t < NACC; ++t) {
                        const uint4 xa = x[(t + q + 5 + rep) & 7], xb = x[(t + q + 6 + rep) & 7];
                        acc[t] = mc16(wz[(q + 2 + rep) & 7], xa, make_uint2(xb.x, xb.y), acc[t]);
                    }
            }
        }
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < NR; ++r) s += acc[t][r];
    out[tid] = s;
}

template <int SHAPE, int MIX>
static void run(const char* name, const uint4* rnd, float* out, double seconds) {
    const int grid = 256;
    constexpr int NACC = SHAPE == 0 ? 4 : 16;
    const double flops_per_inst = SHAPE == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
    const int iters = 10000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<SHAPE, MIX><<<grid, 512>>>(rnd, out, 100);
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    double ms_sum = 0; int launches = 0;
    printf("BEGIN %s\n", name); fflush(stdout);
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < 4; ++r) k<SHAPE, MIX><<<grid, 512>>>(rnd, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.5 * seconds) { ms_sum += ms; launches += 4; }
    }
    const double waves = grid * 8.0, sec = ms_sum * 1e-3;
    const double tf = launches * waves * (double)iters * 8 * NACC * flops_per_inst / sec * 1e-12;
    printf("END %s : %.1f TFLOP/s in fp16-MFMA flops (%.3f of 2500), %d launches of %.2f ms\n", name, tf, tf / 2500.0, launches, ms_sum / (launches ? launches : 1));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    std::vector<uint4> h(1 << 16);
    unsigned long long s = 0x9e3779b97f4a7c15ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto hh = [&]() -> unsigned {
        const float f = ((float)(next() >> 40) / (float)(1 << 24) * 2.f - 1.f) * 0.125f;
        _Float16 q = (_Float16)f; unsigned short b; memcpy(&b, &q, 2); return b;
    };
    for (auto& q : h) { q.x = hh() | (hh() << 16); q.y = hh() | (hh() << 16); q.z = hh() | (hh() << 16); q.w = hh() | (hh() << 16); }
    uint4* rnd; float* out;
    CK(hipMalloc(&rnd, h.size() * sizeof(uint4))); CK(hipMalloc(&out, 256 * 512 * 4));
    CK(hipMemcpy(rnd, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
    run<0, 0>("f16_32 (32x32x16 only)", rnd, out, seconds);
    run<1, 0>("f16_16 (16x16x32 only)", rnd, out, seconds);
    run<0, 1>("mix32 (2 x 32x32x16 f16 + 1 x 32x32x64 fp4 x fp6)", rnd, out, seconds);
    run<1, 1>("mix16 (2 x 16x16x32 f16 + 1 x 16x16x128 fp4 x fp6)", rnd, out, seconds);
    run<1, 2>("mix16z (2 x 16x16x32 f16 + 2 x 16x16x128 fp4 x fp6 with A zero in lanes 32-63: one correction instruction per fp16 one)", rnd, out, seconds);
    run<0, 1>("mix32 again", rnd, out, seconds);
    run<1, 1>("mix16 again", rnd, out, seconds);
    run<1, 2>("mix16z again", rnd, out, seconds);
    return 0;
}
