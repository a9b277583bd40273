// What decides the MFMA pipe's rate under the package power cap: the instruction's SHAPE, the ORDER in which consecutive instructions
// re-use an operand, or the DATA?
//
// profiles/r03_a_power_ceiling.log: v_mfma_f32_32x32x16_f16 on random register-resident operands sustains 1709 TFLOP/s (0.68 of the
// datasheet peak) at the cap, on all-zero operands 0.99 of it at 885 W - the operands' activity is where the watts go, and every GRU
// kernel of this library sits on that cap.  This sweep holds everything else fixed (one 512-thread workgroup per CU, two waves per SIMD,
// register-resident operands, independent accumulators) and varies one thing at a time:
//   shape : 32x32x16 (8 passes, 16 accumulator registers per tile) | 16x16x32 (4 accumulator registers per tile: per flop twice the
//           operand bytes, half the accumulator traffic)
//   order : 'a' consecutive instructions share A (the product kernels' order: one weight fragment against the row tiles)
//           'b' consecutive instructions share B | 'n' neither (both operands change with every instruction)
//           'r' the SAME A and B for a run of four instructions on four accumulators (what an operand latch would reward)
//   data  : uniform fp16 in (-1/8, 1/8) | "lo-like" (hi operands as before on one side, residuals = uniform x 2^-12 on the other)
//           | A zero | B zero | both zero | all mantissas cut to 4 bits | positive only (no sign toggles)
//   accs  : 4 | 8 | 12 independent accumulator tiles of 32x32 (the product kernels carry 9)
// Prints the sustained rate (second half of each run; the governor needs ~1 s to settle); tools/ubench/run_power_ceiling.py samples
// sclk and power beside it (CCSM_UBENCH_EXE=mfma_power_shapes).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power_shapes.hip -o tools/ubench/_build/mfma_power_shapes
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

__device__ __forceinline__ f32x16 mf32(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mf16(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}

// SHAPE 0: 32x32x16, 1: 16x16x32.  ORDER 0 'a', 1 'b', 2 'n', 3 'r'.  NACC accumulator tiles (a multiple of 4).
template <int SHAPE, int ORDER, int NACC>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ ra, const uint4* __restrict__ rb, float* out, int iters) {
    const int tid = blockIdx.x * 512 + threadIdx.x;
    constexpr int NW = 8, NX = 8;
    uint4 w[NW], x[NX];
    for (int i = 0; i < NW; ++i) w[i] = ra[(tid * NW + i) & 0xffff];
    for (int i = 0; i < NX; ++i) x[i] = rb[(tid * NX + i) & 0xffff];
    using acc_t = typename std::conditional<SHAPE == 0, f32x16, f32x4>::type;
    constexpr int NR = SHAPE == 0 ? 16 : 4;
    acc_t acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < NR; ++r) acc[t][r] = 0.f;
    auto mm = [&](uint4 a, uint4 b, acc_t c) -> acc_t {
        if constexpr (SHAPE == 0) return mf32(a, b, c); else return mf16(a, b, c);
    };
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
        // one iteration = 8 groups of NACC instructions; group q uses operand index q
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int t = 0; t < NACC; ++t) {
                if constexpr (ORDER == 0) acc[t] = mm(w[q], x[(t + q) & 7], acc[t]);                 // A shared by the group
                else if constexpr (ORDER == 1) acc[t] = mm(w[(t + q) & 7], x[q], acc[t]);            // B shared by the group
                else if constexpr (ORDER == 2) acc[t] = mm(w[(t + q) & 7], x[(2 * t + q + 3) & 7], acc[t]);   // neither
                else acc[t] = mm(w[(q + (t >> 2)) & 7], x[(q + 3 * (t >> 2)) & 7], acc[t]);          // runs of four with both operands the same
            }
        }
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < NR; ++r) s += acc[t][r];
    out[tid] = s;
}

static std::vector<uint4> fill(int kind) {
    // kind 0: uniform fp16 in (-1/8, 1/8); 1: residual-like (uniform x 2^-12 x 1/8); 2: zero; 3: mantissa cut to 4 bits (the low 7 of 10 cleared);
    // 4: positive only; 5: activations-like: uniform in (-1, 1)
    std::vector<uint4> v(1 << 16);
    unsigned long long s = 0x9e3779b97f4a7c15ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto h = [&]() -> unsigned {
        float f = ((float)(next() >> 40) / (float)(1 << 24) * 2.f - 1.f) * 0.125f;
        if (kind == 1) f *= 1.f / 4096.f;
        if (kind == 2) f = 0.f;
        if (kind == 4) f = f < 0 ? -f : f;
        if (kind == 5) f *= 8.f;
        _Float16 q = (_Float16)f; unsigned short b; memcpy(&b, &q, 2);
        if (kind == 3) b &= 0xff80;
        return b;
    };
    for (auto& q : v) { q.x = h() | (h() << 16); q.y = h() | (h() << 16); q.z = h() | (h() << 16); q.w = h() | (h() << 16); }
    return v;
}

template <int SHAPE, int ORDER, int NACC>
static void run(const char* name, const uint4* ra, const uint4* rb, float* out, double seconds) {
    const int grid = 256;
    const double flops_per_inst = SHAPE == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
    const int iters = (int)(20000.0 * 16 / NACC * (SHAPE == 0 ? 1 : 2) / 4);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<SHAPE, ORDER, NACC><<<grid, 512>>>(ra, rb, out, 100);
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    double ms_sum = 0; int launches = 0;
    printf("BEGIN %s\n", name); fflush(stdout);
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < 4; ++r) k<SHAPE, ORDER, NACC><<<grid, 512>>>(ra, rb, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.5 * seconds) { ms_sum += ms; launches += 4; }
    }
    const double waves = grid * 8.0, sec = ms_sum * 1e-3;
    const double tf = launches * waves * (double)iters * 8 * NACC * flops_per_inst / sec * 1e-12;
    printf("END %s : %.1f TFLOP/s (%.3f of 2500), %d launches of %.2f ms\n", name, tf, tf / 2500.0, launches, ms_sum / (launches ? launches : 1));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    uint4 *buf[6]; float* out;
    for (int kind = 0; kind < 6; ++kind) {
        std::vector<uint4> h = fill(kind);
        CK(hipMalloc(&buf[kind], h.size() * sizeof(uint4)));
        CK(hipMemcpy(buf[kind], h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&out, 256 * 512 * 4));
    // order (shape 32x32x16, 4 and 12 accumulators, uniform data)
    run<0, 0, 4>("32x32x16 order=a accs=4 data=uniform", buf[0], buf[0], out, seconds);
    run<0, 1, 4>("32x32x16 order=b accs=4 data=uniform", buf[0], buf[0], out, seconds);
    run<0, 2, 4>("32x32x16 order=n accs=4 data=uniform", buf[0], buf[0], out, seconds);
    run<0, 3, 4>("32x32x16 order=r accs=4 data=uniform", buf[0], buf[0], out, seconds);
    run<0, 0, 8>("32x32x16 order=a accs=8 data=uniform", buf[0], buf[0], out, seconds);
    run<0, 1, 8>("32x32x16 order=b accs=8 data=uniform", buf[0], buf[0], out, seconds);
    run<0, 3, 8>("32x32x16 order=r accs=8 data=uniform", buf[0], buf[0], out, seconds);
    // shape
    run<1, 0, 16>("16x16x32 order=a accs=16 data=uniform", buf[0], buf[0], out, seconds);
    run<1, 1, 16>("16x16x32 order=b accs=16 data=uniform", buf[0], buf[0], out, seconds);
    run<1, 2, 16>("16x16x32 order=n accs=16 data=uniform", buf[0], buf[0], out, seconds);
    run<1, 0, 32>("16x16x32 order=a accs=32 data=uniform", buf[0], buf[0], out, seconds);
    // data (32x32x16, order a, 4 accumulators)
    run<0, 0, 4>("32x32x16 order=a accs=4 data=A:weights B:activations(-1,1)", buf[0], buf[5], out, seconds);
    run<0, 0, 4>("32x32x16 order=a accs=4 data=A:residual B:uniform", buf[1], buf[0], out, seconds);
    run<0, 0, 4>("32x32x16 order=a accs=4 data=A:uniform B:residual", buf[0], buf[1], out, seconds);
    run<0, 0, 4>("32x32x16 order=a accs=4 data=A:zero B:uniform", buf[2], buf[0], out, seconds);
    run<0, 0, 4>("32x32x16 order=a accs=4 data=A:uniform B:zero", buf[0], buf[2], out, seconds);
    run<0, 0, 4>("32x32x16 order=a accs=4 data=zero", buf[2], buf[2], out, seconds);
    run<0, 0, 4>("32x32x16 order=a accs=4 data=mantissa4", buf[3], buf[3], out, seconds);
    run<0, 0, 4>("32x32x16 order=a accs=4 data=positive", buf[4], buf[4], out, seconds);
    return 0;
}
