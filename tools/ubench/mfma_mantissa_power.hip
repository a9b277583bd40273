// Does the fp16 MFMA's energy depend on how many MANTISSA bits of its operands are non-zero?  The GRU kernels run at the package power
// cap (time = energy / cap), and in the three-pass split-fp16 arithmetic two of the three products have a residual operand (lo = v - fp16(v))
// that needs ~8 significant bits, not 11, for fp32-class results (error 2^-12 * 2^-9 = 2^-21 relative).  This probe issues
// v_mfma_f32_32x32x16_f16 back to back (two waves per SIMD, one workgroup per CU, operands resident in registers, A and B register sets
// alternating) with operand classes
//   full   : random fp16, all 10 stored mantissa bits random
//   m7/m3/m0 : the low 3 / 7 / 10 stored mantissa bits cleared (8 / 4 / 1 significant bits)
// in the combinations a three-pass kernel would issue, for a few seconds each, and prints the sustained rate; run under
// tools/ubench/run_power_ceiling.py-style sampling or read the rate alone (at the cap, rate ~ 1 / energy per instruction).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_mantissa_power.hip -o tools/ubench/_build/mfma_mantissa_power
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ f32x16 mf(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
// per iteration: for each of 4 row tiles the three products of one k-block: (A0, B0), (A1, B0), (A0, B1); A0/A1/B0/B1 come from four
// operand classes (buffers); two register sets alternate
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ a0p, const uint4* __restrict__ a1p, const uint4* __restrict__ b0p,
                                         const uint4* __restrict__ b1p, float* out, int iters) {
    const int tid = blockIdx.x * 512 + threadIdx.x;
    uint4 a0[2], a1[2], b0[4], b1[4];
    for (int i = 0; i < 2; ++i) { a0[i] = a0p[(tid * 2 + i) & 0xffff]; a1[i] = a1p[(tid * 2 + i + 77) & 0xffff]; }
    for (int i = 0; i < 4; ++i) { b0[i] = b0p[(tid * 4 + i + 13) & 0xffff]; b1[i] = b1p[(tid * 4 + i + 991) & 0xffff]; }
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mf(a0[h], b0[t], acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mf(a1[h], b0[(t + 1) & 3], acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mf(a0[h], b1[t], acc[t]);
            asm volatile("" ::: "memory");
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[tid] = s;
}
static std::vector<uint4> operands(unsigned mask, float scale, unsigned long long seed) {
    std::vector<uint4> v(1 << 16);
    unsigned long long s = seed;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto h = [&]() -> unsigned {
        const float f = ((float)(next() >> 40) / (float)(1 << 24) * 2.f - 1.f) * scale;
        _Float16 q = (_Float16)f; unsigned short b; memcpy(&b, &q, 2); return b & mask;
    };
    for (auto& q : v) { q.x = h() | (h() << 16); q.y = h() | (h() << 16); q.z = h() | (h() << 16); q.w = h() | (h() << 16); }
    return v;
}
int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const unsigned FULL = 0xffff, M7 = 0xfff8, M3 = 0xff80, M0 = 0xfc00;
    struct Cls { const char* name; unsigned mask; float scale; };
    const Cls cls[] = {{"full", FULL, 0.125f}, {"m7", M7, 0.125f}, {"m3", M3, 0.125f}, {"m0", M0, 0.125f}, {"full_small", FULL, 0.125f / 2048.f}, {"m7_small", M7, 0.125f / 2048.f}, {"zero", 0, 0.f}};
    const int NC = sizeof(cls) / sizeof(cls[0]);
    uint4* buf[NC];
    for (int i = 0; i < NC; ++i) {
        std::vector<uint4> h = operands(cls[i].mask, cls[i].scale, 0x9e3779b97f4a7c15ull + 1234567ull * i);
        CK(hipMalloc(&buf[i], h.size() * sizeof(uint4))); CK(hipMemcpy(buf[i], h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
    }
    float* out; CK(hipMalloc(&out, 256 * 512 * 4));
    // (A0 = W_hi, A1 = W_lo, B0 = x_hi, B1 = x_lo)
    struct Run { const char* name; int a0, a1, b0, b1; };
    const Run runs[] = {{"all full (three passes on random operands)", 0, 0, 0, 0},
                        {"split3 as shipped: lo operands full mantissa, small exponent", 0, 4, 0, 4},
                        {"split3 with lo operands cut to 8 significant bits", 0, 5, 0, 5},
                        {"lo operands 8 bits (same exponent range as hi)", 0, 1, 0, 1},
                        {"lo operands 4 bits", 0, 2, 0, 2},
                        {"lo operands 1 bit (power of two)", 0, 3, 0, 3},
                        {"everything 8 bits", 1, 1, 1, 1},
                        {"everything 4 bits", 2, 2, 2, 2},
                        {"lo operands zero", 0, 6, 0, 6},
                        {"all zero", 6, 6, 6, 6}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000, grid = 256;
    for (const Run& r : runs) {
        k<<<grid, 512>>>(buf[r.a0], buf[r.a1], buf[r.b0], buf[r.b1], out, 200);
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        double ms_sum = 0; int launches = 0;
        printf("BEGIN %s\n", r.name); fflush(stdout);
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            CK(hipEventRecord(e0));
            for (int q = 0; q < 4; ++q) k<<<grid, 512>>>(buf[r.a0], buf[r.a1], buf[r.b0], buf[r.b1], out, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.5 * seconds) { ms_sum += ms; launches += 4; }
        }
        const double waves = (double)grid * 8, flops_it = 12.0 * 2 * 32 * 32 * 16;
        const double tf = launches * waves * iters * flops_it / (ms_sum * 1e-3) * 1e-12;
        printf("END %-62s : %.1f TFLOP/s of fp16 MFMA issued (%.3f of 2500)\n", r.name, tf, tf / 2500.0); fflush(stdout);
    }
    return 0;
}
