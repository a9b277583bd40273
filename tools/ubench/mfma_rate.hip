// Microbenchmark: cycles per v_mfma_f32_32x32x16_f16 for one wave, with NACC independent accumulators issued round-robin,
// 1 or 2 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int NOPER>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    half8 a[NOPER], b[NOPER];
    for (int i = 0; i < NOPER; ++i)
        for (int j = 0; j < 8; ++j) { a[i][j] = (_Float16)(threadIdx.x * 0.001f + i + j); b[i][j] = (_Float16)(j * 0.01f + i); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + p) % NOPER], b[(i * 2 + p) % NOPER], acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int NOPER>
void run(int threads, const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k<NACC, NOPER>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<NACC, NOPER>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s: NACC=%d threads=%d  cycles per MFMA per wave = %.1f\n", name, NACC, threads, (double)c / (iters * 3.0 * NACC));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<6, 4>(256, "1 wave/SIMD");
    run<6, 4>(512, "2 waves/SIMD");
    run<3, 4>(256, "1 wave/SIMD");
    run<3, 4>(512, "2 waves/SIMD");
    run<9, 4>(256, "1 wave/SIMD");
    run<9, 4>(512, "2 waves/SIMD");
    run<2, 2>(256, "1 wave/SIMD");
    run<1, 1>(256, "1 wave/SIMD");
    return 0;
}
