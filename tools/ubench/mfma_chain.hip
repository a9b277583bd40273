// Does a dependent chain of MFMAs on one accumulator cost issue slots?  7 accumulators, per accumulator [f16 main, f16 main, fp8 scaled corr]
// as in attn_fc_f8_kernel's chunk loop: CHAIN = the three MFMAs of an accumulator back to back (the kernel's order);
// INTERLEAVED = main over all 7, main over all 7, corr over all 7.  512 threads per workgroup (2 waves per SIMD), one workgroup per CU.
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
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    uint4 w0 = make_uint4(0x3c003c00 + lane, 0x3c003c00, 0x34003400, 0x30003000), w1 = make_uint4(0x38003800, 0x3c003c00 + lane, 0x34003400, 0x30003000);
    uint4 x[7];
    for (int t = 0; t < 7; ++t) x[t] = make_uint4(0x3c003c00 + t, 0x38003800, 0x34003400 + lane, 0x30003000);
    f32x16 acc[7];
    for (int t = 0; t < 7; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                acc[t] = mf(w0, x[t], acc[t]);
                acc[t] = mf(w1, x[t], acc[t]);
                acc[t] = mc(w0, w1, x[t], x[(t + 1) % 7], acc[t]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 7; ++t) acc[t] = mf(w0, x[t], acc[t]);
#pragma unroll
            for (int t = 0; t < 7; ++t) acc[t] = mf(w1, x[t], acc[t]);
#pragma unroll
            for (int t = 0; t < 7; ++t) acc[t] = mc(w0, w1, x[t], x[(t + 1) % 7], acc[t]);
        }
        asm volatile("" ::: "memory");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < 7; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int MODE>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
    const int iters = 2000;
    k<MODE><<<256, 512>>>(out, cyc, iters);
    k<MODE><<<256, 512>>>(out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[8], last = 0;
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    for (int w = 0; w < 8; ++w) last = h[w] > last ? h[w] : last;      // the oldest wave of a SIMD is served first: report the slowest
    printf("%-12s cycles per iteration (21 MFMAs per wave, 2 waves per SIMD): %.0f  (issue 2 x 7 x (32 + 32 + 64) = 1792)\n", name, (double)last / iters);
}
int main() { run<0>("chain"); run<1>("interleaved"); return 0; }
