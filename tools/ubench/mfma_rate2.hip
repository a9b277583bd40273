// Mimic of the GRU phase-A MFMA group: 9 accumulator tiles resident (6 used per group), operands as in CCSM_MM
// (2 gates x hi/lo weights, 3 batch tiles x hi/lo activations), optionally refreshed from LDS / global each group.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mf(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
template <int MODE>   // 0: static operands; 1: x from LDS each group; 2: + w from global each group (double-buffered)
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, const uint4* wsrc, int iters) {
    __shared__ uint4 lds[6 * 64 * 4];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 6 * 64 * 4; i += blockDim.x) lds[i] = make_uint4(i, i * 3, 0x3c003c00, 0x38003800);
    __syncthreads();
    uint4 w[2][2][2], x[3][2];
    for (int b = 0; b < 2; ++b) for (int g = 0; g < 2; ++g) for (int h = 0; h < 2; ++h) w[b][g][h] = make_uint4(0x3c003c00 + lane, 0x3c003c00, 0x34003400 + g, 0x30003000 + h);
    for (int b = 0; b < 3; ++b) for (int h = 0; h < 2; ++h) x[b][h] = make_uint4(0x3c003c00, 0x38003800 + b, 0x34003400 + h, 0x30003000 + lane);
    f32x16 acc[3][3];
    for (int s = 0; s < 3; ++s) for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) acc[s][i][r] = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (MODE >= 2) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int h = 0; h < 2; ++h) w[j ^ 1][g][h] = wsrc[(((it * 2 + j) & 255) * 4 + g * 2 + h) * 64 + lane];
            }
            if (MODE >= 1) {
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int h = 0; h < 2; ++h) x[b][h] = lds[((j * 3 + b) * 2 + h) * 64 + lane];
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[g][b] = mf(w[j][g][0], x[b][0], acc[g][b]);
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[g][b] = mf(w[j][g][0], x[b][1], acc[g][b]);
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[g][b] = mf(w[j][g][1], x[b][0], acc[g][b]);
            asm volatile("" ::: "memory");
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int q = 0; q < 3; ++q) for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) s += acc[q][i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int MODE>
void run(int threads, const char* name) {
    float* out; unsigned long long* cyc; uint4* wsrc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64); hipMalloc(&wsrc, 256 * 4 * 64 * 16); hipMemset(wsrc, 0x3c, 256 * 4 * 64 * 16);
    const int iters = 1000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, wsrc, iters);
    hipDeviceSynchronize();
    unsigned long long c[8]; hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("%s mode %d threads %d: cycles per MFMA: wave0 %.1f", name, MODE, threads, (double)c[0] / (iters * 36.0));
    if (threads == 512) printf("  wave4 %.1f", (double)c[4] / (iters * 36.0));
    printf("\n");
}
int main() {
    run<0>(256, "lone"); run<0>(512, "pair");
    run<1>(256, "lone"); run<1>(512, "pair");
    run<2>(256, "lone"); run<2>(512, "pair");
    return 0;
}
