// LDS read bandwidth of the operand pattern of the GRU / attention kernels: every wave of a 512-thread workgroup reads the same
// 24 fragments (1 KiB each, lane-linear ds_read_b128) over and over.  Reports bytes per clock per CU (slowest wave).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES>
__global__ __launch_bounds__(512) void k(unsigned* out, unsigned long long* cyc, int iters) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 24 * 1024 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    unsigned acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < 24; ++f) {
            if (BYTES == 16) {
                const uint4 v = *reinterpret_cast<const uint4*>(smem + f * 1024 + lane * 16);
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            } else {
                const uint2 v = *reinterpret_cast<const uint2*>(smem + f * 1024 + lane * 8);
                const uint2 u = *reinterpret_cast<const uint2*>(smem + f * 1024 + 512 + lane * 8);
                acc ^= v.x ^ v.y ^ u.x ^ u.y;
            }
        }
        asm volatile("" ::: "memory");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int BYTES>
void run(const char* name) {
    unsigned* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
    const int iters = 2000;
    k<BYTES><<<256, 512, 24 * 1024>>>(out, cyc, iters);
    k<BYTES><<<256, 512, 24 * 1024>>>(out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[8], last = 0;
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    for (int w = 0; w < 8; ++w) last = h[w] > last ? h[w] : last;
    printf("%s: %.1f bytes per clock per CU (8 waves x 24 KiB per iteration in %.0f clocks)\n", name, 8.0 * 24 * 1024 * iters / (double)last, (double)last / iters);
}
int main() { run<16>("ds_read_b128"); run<8>("ds_read_b64 "); return 0; }
