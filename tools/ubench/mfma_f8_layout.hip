// Layout / scale / rate probe of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3) on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f8_layout.hip -o /tmp/mfma_f8_layout && /tmp/mfma_f8_layout
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void probe(const int* a, const int* b, float* c, int sa, int sb) {
    i32x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = a[threadIdx.x * 8 + i]; B[i] = b[threadIdx.x * 8 + i]; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0, 0, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) c[threadIdx.x * 16 + i] = acc[i];
}
__global__ void cvt(const float* in, int* out) {
    const float x = in[2 * threadIdx.x], y = in[2 * threadIdx.x + 1];
    out[threadIdx.x] = __builtin_amdgcn_cvt_pk_fp8_f32(x, y, 0x77777777, false);
    out[64 + threadIdx.x] = __builtin_amdgcn_cvt_pk_fp8_f32(x, y, 0x77777777, true);
}
template <int MODE>
__global__ void rate(float* out, int iters, unsigned long long* cyc) {
    i32x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = 0x38383838 + threadIdx.x; B[i] = 0x38383838 + i; }
    f16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * threadIdx.x); hb[i] = (_Float16)(0.5f); }
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE == 0) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[j], 0, 0, 0, 127, 0, 127);
            if (MODE == 1) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[j], 0, 0, 0);
            if (MODE == 2) {   // the intended mix: 2 f16 + 1 scaled fp8 per 32 k
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[j], 0, 0, 0, 116, 0, 127);
            }
            if (MODE == 3) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[j], 2, 2, 0, 127, 0, 127);  // fp6 x fp6
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static float fp8_decode(unsigned v) {   // OCP e4m3fn
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}

int main() {
    std::vector<int> ha(512), hb(512);
    std::vector<float> A(32 * 64), B(32 * 64);
    srand(1);
    // hypothesis: lane l holds row/col l%32, k = 32*(l/32) + 4*reg + byte
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 8; ++r) {
            unsigned wa = 0, wb = 0;
            for (int b = 0; b < 4; ++b) {
                const unsigned va = (rand() % 0x78) | ((rand() & 1) << 7), vb = (rand() % 0x78) | ((rand() & 1) << 7);
                wa |= va << (8 * b); wb |= vb << (8 * b);
                const int k = 32 * (l / 32) + 4 * r + b;
                A[(l % 32) * 64 + k] = fp8_decode(va);
                B[(l % 32) * 64 + k] = fp8_decode(vb);
            }
            ha[l * 8 + r] = (int)wa; hb[l * 8 + r] = (int)wb;
        }
    int *da, *db; float* dc;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dc, 64 * 16 * 4);
    hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice);
    for (int sa : {127, 116}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, sa, 127);
        std::vector<float> hc(1024);
        hipMemcpy(hc.data(), dc, 4096, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                double ref = 0;
                for (int k = 0; k < 64; ++k) ref += (double)A[row * 64 + k] * B[col * 64 + k];
                ref *= std::ldexp(1.0, sa - 127);
                maxerr = std::fmax(maxerr, std::fabs(ref - hc[l * 16 + r]));
                maxref = std::fmax(maxref, std::fabs(ref));
            }
        printf("scale_a=%d: C[row=A idx][col=B idx] max abs err %.3g (max |ref| %.3g)\n", sa, maxerr, maxref);
    }
    // cvt_pk_fp8_f32
    {
        std::vector<float> in = {0.0f, 1.0f, -1.5f, 0.3f, 448.0f, 1000.0f, 1e-3f, 2e-3f, 0.0625f, 0.07f, -0.001953125f, 3.3f};
        in.resize(128, 0.f);
        float* din; int* dout;
        hipMalloc(&din, 512); hipMalloc(&dout, 512);
        hipMemcpy(din, in.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, din, dout);
        std::vector<int> o(128);
        hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
        for (int i = 0; i < 6; ++i)
            printf("cvt(%g, %g): lo-word %08x -> (%g, %g) ; hi-word %08x\n", in[2 * i], in[2 * i + 1], o[i], fp8_decode(o[i] & 255),
                   fp8_decode((o[i] >> 8) & 255), o[64 + i]);
    }
    // issue rate, one wave per SIMD (256 threads = 4 waves) on one CU
    float* dout; unsigned long long* dcyc;
    hipMalloc(&dout, 1 << 20); hipMalloc(&dcyc, 8);
    const int iters = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(1), dim3(256), 0, 0, dout, iters, dcyc);
            if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(1), dim3(256), 0, 0, dout, iters, dcyc);
            if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(1), dim3(256), 0, 0, dout, iters, dcyc);
            if (mode == 3) hipLaunchKernelGGL(rate<3>, dim3(1), dim3(256), 0, 0, dout, iters, dcyc);
            hipDeviceSynchronize();
        }
        unsigned long long c; hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
        const int per_it = mode == 2 ? 12 : 4;
        printf("mode %d (%s): %.1f clk per MFMA (s_memtime-class counter, 100 MHz ticks scaled? raw %llu for %d MFMAs)\n", mode,
               mode == 0 ? "fp8 x fp8 32x32x64" : mode == 1 ? "f16 32x32x16" : mode == 2 ? "2 f16 + 1 fp8" : "fp6 x fp6 32x32x64",
               (double)c / (iters * per_it), c, iters * per_it);
    }
    return 0;
}
