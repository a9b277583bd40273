// How many waves per SIMD does it take to fill the matrix pipe?  Each wave issues independent MFMAs on 4 accumulators
// (8 f16 32x32x16 + 4 fp8 32x32x64 per iteration); workgroups of 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD, one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mf(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mc(uint4 a0, uint4 a1, uint4 b0, uint4 b1, f32x16 c) {
    const i32x8 a = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
    const i32x8 b = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 120, 0, 121);
}
template <int T, int KIND>   // KIND 0: f16 + fp8 mix, 1: f16 only, 2: fp8 only
__global__ __launch_bounds__(T) void k(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    uint4 w0 = make_uint4(0x3c003c00 + lane, 0x3c003c00, 0x34003400, 0x30003000), w1 = make_uint4(0x38003800, 0x3c003c00 + lane, 0x34003400, 0x30003000);
    uint4 x[4];
    for (int t = 0; t < 4; ++t) x[t] = make_uint4(0x3c003c00 + t, 0x38003800, 0x34003400 + lane, 0x30003000);
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
        if (KIND != 2) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mf(w0, x[t], acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mf(w1, x[t], acc[t]);
        }
        if (KIND != 1) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mc(w0, w1, x[t], x[(t + 1) % 4], acc[t]);
        }
        asm volatile("" ::: "memory");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int T, int KIND>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 256);
    const int iters = 4000;
    k<T, KIND><<<256, T>>>(out, cyc, iters);
    k<T, KIND><<<256, T>>>(out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[16], first = ~0ull, last = 0;
    (void)hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
    for (int w = 0; w < T / 64; ++w) { first = h[w] < first ? h[w] : first; last = h[w] > last ? h[w] : last; }
    // the oldest wave of a SIMD is served first and runs at the single-wave rate; the slowest wave shows the shared pipe
    printf("%-10s %d wave(s) per SIMD: %.1f ticks per iteration for the fastest wave, %.1f for the slowest\n", name, T / 256, (double)first / iters,
           (double)last / iters);
}
int main() {
    run<256, 1>("8 x f16"); run<512, 1>("8 x f16"); run<1024, 1>("8 x f16");
    run<256, 2>("4 x fp8"); run<512, 2>("4 x fp8"); run<1024, 2>("4 x fp8");
    run<256, 0>("8 + 4 mix"); run<512, 0>("8 + 4 mix"); run<1024, 0>("8 + 4 mix");
    return 0;
}
