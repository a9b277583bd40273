// Which SIDE of the matrix instruction should carry which operand at the package power cap?  
// profiles/r05_m found a toggling B operand ~3 x as expensive as a toggling A operand (zero B: 1025 W, zero A: 1330 W at full clock) and
// 4-bit mantissas worth +17 %.  The split-mx kernels put the WEIGHTS on A and the ACTIVATIONS on B (C^T[unit][row] = W x^T).  This measures the
// kernels' instruction mix - per two v_mfma_f32_32x32x16_f16 one v_mfma_scale_f32_32x32x64_f8f6f4 - on operands that LOOK like the kernels'
// (weights uniform in +-1/16 as fp16 and as fp4 e2m1 blobs of [W_lo 2^11 | W_hi] under block scales; activations tanh-shaped in (-1, 1) as fp16
// and as fp6 e2m3 blobs of [x_hi 4 | x_lo 2^14]) in both assignments:
//   w_on_A : A = weights (fp16 / fp4 blob), B = activations (fp16 / fp6 blob)          - what ships
//   w_on_B : A = activations (fp16 / fp6 blob: cbsz 2), B = weights (fp16 / fp4 blob: blgp 4)   - the transposed product C[row][unit]
// and, for reference, both sides random (the operands of mfma_power_mix_shapes.hip's mix32).  One 512-thread workgroup per CU, register-resident
// operands, independent accumulators; rates in fp16-MFMA flops.  Run under tools/ubench/run_power_ceiling.py (CCSM_UBENCH_EXE=mfma_operand_roles)
// for the clock and the watts beside each rate.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_operand_roles.hip -o tools/ubench/_build/mfma_operand_roles
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mf32(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
// fp4 (4 dwords) on one side, fp6 (6 dwords) on the other; block scales 2^-4
template <bool FP4_ON_A>
__device__ __forceinline__ f32x16 mc32(uint4 q4, uint4 q6a, uint2 q6b, f32x16 c) {
    const i32x8 v4 = {(int)q4.x, (int)q4.y, (int)q4.z, (int)q4.w, 0, 0, 0, 0};
    const i32x8 v6 = {(int)q6a.x, (int)q6a.y, (int)q6a.z, (int)q6a.w, (int)q6b.x, (int)q6b.y, 0, 0};
    if constexpr (FP4_ON_A) return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v4, v6, c, 4, 2, 0, 123, 0, 123);
    else return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v6, v4, c, 2, 4, 0, 123, 0, 123);
}

// wh / xh: fp16 operand registers of the weights / the activations; wq: fp4 blobs; xq: fp6 blobs (uint4 + the low half of the next)
template <bool W_ON_A>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ wh_, const uint4* __restrict__ xh_, const uint4* __restrict__ wq_,
                                         const uint4* __restrict__ xq_, float* out, int iters) {
    const int tid = blockIdx.x * 512 + threadIdx.x;
    uint4 w[8], x[8], wq[4], xq[8];
    for (int i = 0; i < 8; ++i) { w[i] = wh_[(tid * 8 + i) & 0xffff]; x[i] = xh_[(tid * 8 + i) & 0xffff]; xq[i] = xq_[(tid * 8 + i) & 0xffff]; }
    for (int i = 0; i < 4; ++i) wq[i] = wq_[(tid * 4 + i) & 0xffff];
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = W_ON_A ? mf32(w[q], x[(t + q) & 7], acc[t]) : mf32(x[(t + q) & 7], w[q], acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = W_ON_A ? mf32(w[q + 1], x[(t + q + 3) & 7], acc[t]) : mf32(x[(t + q + 3) & 7], w[q + 1], acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint4 xa = xq[(t + q + 5) & 7], xb = xq[(t + q + 6) & 7];
                acc[t] = mc32<W_ON_A>(wq[(q >> 1) & 3], xa, make_uint2(xb.x, xb.y), acc[t]);
            }
        }
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[tid] = s;
}

template <bool W_ON_A>
static void run(const char* name, const uint4* wh, const uint4* xh, const uint4* wq, const uint4* xq, float* out, double seconds) {
    const int grid = 256, iters = 10000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<W_ON_A><<<grid, 512>>>(wh, xh, wq, xq, out, 100);
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    double ms_sum = 0; int launches = 0;
    printf("BEGIN %s\n", name); fflush(stdout);
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < 4; ++r) k<W_ON_A><<<grid, 512>>>(wh, xh, wq, xq, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.5 * seconds) { ms_sum += ms; launches += 4; }
    }
    const double tf = launches * grid * 8.0 * (double)iters * 8 * 4 * (2.0 * 32 * 32 * 16) / (ms_sum * 1e-3) * 1e-12;
    printf("END %s : %.1f TFLOP/s in fp16-MFMA flops (%.3f of 2500), %d launches of %.2f ms\n", name, tf, tf / 2500.0, launches, ms_sum / (launches ? launches : 1));
    fflush(stdout);
}

static unsigned short f16bits(float f) { _Float16 q = (_Float16)f; unsigned short b; memcpy(&b, &q, 2); return b; }
// MX element codes (round to nearest): fp4 e2m1 {0, .5, 1, 1.5, 2, 3, 4, 6}, fp6 e2m3 (bias 1)
static unsigned mx_code(float v, int mb) {
    const float vmax = mb == 3 ? 7.5f : 6.0f;
    const unsigned sign = std::signbit(v) ? (1u << (2 + mb)) : 0u;
    float a = std::fmin(std::fabs(v), vmax);
    int e; (void)std::frexp(a, &e);
    int ex = a > 0.f ? e - 1 : 0;
    if (ex < 0) ex = 0;
    const float step = std::ldexp(1.0f, ex - mb);
    float r = std::fmin(std::nearbyint(a / step) * step, vmax);
    if (r < 1.0f) return sign | (unsigned)std::lrint(r * (float)(1 << mb));
    int e2; const float m2 = std::frexp(r, &e2);
    return sign | (unsigned)((e2 << mb) | (int)std::lrint((m2 * 2.0f - 1.0f) * (float)(1 << mb)));
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const size_t N = 1 << 16;
    unsigned long long s = 0x9e3779b97f4a7c15ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto uni = [&]() { return (float)(next() >> 40) / (float)(1 << 24) * 2.f - 1.f; };
    auto gauss = [&]() { float a = 0.f; for (int i = 0; i < 6; ++i) a += uni(); return a * 0.7071f; };
    std::vector<uint4> wh(N), xh(N), wq(N), xq(N), rnd(N), rndq(N);
    auto pack16 = [&](auto gen) { uint4 q; unsigned* p = &q.x; for (int i = 0; i < 4; ++i) { p[i] = f16bits(gen()) | ((unsigned)f16bits(gen()) << 16); } return q; };
    for (size_t i = 0; i < N; ++i) {
        wh[i] = pack16([&]() { return uni() * 0.0625f; });                  // nn.GRU's default initialisation: uniform(-1/16, 1/16)
        xh[i] = pack16([&]() { return std::tanh(gauss()); });               // a GRU state / layer output
        rnd[i] = pack16([&]() { return uni() * 0.125f; });                  // mfma_power_mix_shapes.hip's operands
        // one lane's fp4 blob: 32 values of [W_lo 2^11] (even uint4) or [W_hi] (odd) divided by the block's power-of-two scale
        unsigned char b4[16] = {0};
        float v[32], mx = 0.f;
        for (int j = 0; j < 32; ++j) {
            const float w = uni() * 0.0625f, h = (float)(_Float16)w;
            v[j] = (i & 1) ? h : std::ldexp(w - h, 11);
            mx = std::fmax(mx, std::fabs(v[j]));
        }
        const int lg = mx > 0.f ? (int)std::floor(std::log2(6.0f / mx)) : 0;
        for (int j = 0; j < 32; ++j) b4[j >> 1] |= (unsigned char)(mx_code(std::ldexp(v[j], lg), 1) << ((j & 1) * 4));
        memcpy(&wq[i], b4, 16);
        // fp6 blob pieces: x_hi * 4 (even) / x_lo * 2^14 (odd), 6-bit codes packed back to back (a blob is a uint4 and half of the next)
        unsigned char b6[16] = {0};
        for (int j = 0; j < 21; ++j) {
            const float x = std::tanh(gauss()), h = (float)(_Float16)x;
            const unsigned c = mx_code((i & 1) ? std::ldexp(x - h, 14) : h * 4.0f, 3);
            const int bit = j * 6;
            b6[bit >> 3] |= (unsigned char)(c << (bit & 7));
            if ((bit & 7) + 6 > 8 && (bit >> 3) + 1 < 16) b6[(bit >> 3) + 1] |= (unsigned char)(c >> (8 - (bit & 7)));
        }
        memcpy(&xq[i], b6, 16);
        rndq[i] = make_uint4((unsigned)next(), (unsigned)next(), (unsigned)next(), (unsigned)next());
    }
    uint4 *d[6]; float* out;
    const std::vector<uint4>* src[6] = {&wh, &xh, &wq, &xq, &rnd, &rndq};
    for (int i = 0; i < 6; ++i) { CK(hipMalloc(&d[i], N * sizeof(uint4))); CK(hipMemcpy(d[i], src[i]->data(), N * sizeof(uint4), hipMemcpyHostToDevice)); }
    CK(hipMalloc(&out, 256 * 512 * 4));
    for (int rep = 0; rep < 2; ++rep) {
        run<true>(rep ? "w_on_A again" : "w_on_A (ships: A = weights fp16 / fp4 blob, B = activations fp16 / fp6 blob)", d[0], d[1], d[2], d[3], out, seconds);
        run<false>(rep ? "w_on_B again" : "w_on_B (transposed: A = activations fp16 / fp6 blob, B = weights fp16 / fp4 blob)", d[0], d[1], d[2], d[3], out, seconds);
    }
    run<true>("random operands on both sides, fp4 on A (mfma_power_mix_shapes.hip's mix32)", d[4], d[4], d[5], d[5], out, seconds);
    run<false>("random operands on both sides, fp4 on B", d[4], d[4], d[5], d[5], out, seconds);
    // the main product alone in both assignments (narrow-exponent weights against tanh-shaped activations): scaled instruction fed zeros
    CK(hipMemset(d[5], 0, N * sizeof(uint4)));
    run<true>("w_on_A, correction blobs all zero", d[0], d[1], d[5], d[5], out, seconds);
    run<false>("w_on_B, correction blobs all zero", d[0], d[1], d[5], d[5], out, seconds);
    return 0;
}
