// What does the MFMA pipe of this part sustain under its package power cap with RANDOM operands?
//
// bench.py prices gru_layer12_mx_kernel against the 2.5 PFLOP/s dense fp16 peak, which is an unthrottled figure; under that kernel
// the package sits at its 1400 W cap with sclk at 1.64 of 2.4 GHz (profiles/r02_c_power_attribution.md).  This loop is the kernel's
// matrix work and nothing else: every wave keeps its operands in registers (random fp16 values / random fp6, fp4 codes, a different
// A and B register set for consecutive instructions, so that the operand buses toggle as they do on real data) and issues
//   mode f16 : v_mfma_f32_32x32x16_f16 only
//   mode mix : the product kernel's 64 : 35 issue mix - per two f16 MFMAs one v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 A x fp6 B)
//   mode lds : mix, with every B operand re-read from LDS (ds_read_b128) as the kernel's activations are
// on 4 independent accumulators, 1 or 2 waves per SIMD, one workgroup per CU (256 workgroups).  It runs for the given number of
// seconds and prints the rate in fp16-MFMA flops (the corrections are overhead, as in bench.py's `roofline.achieved`) and in
// issued MFMA cycles; tools/ubench/run_power_ceiling.py samples sclk and power beside it.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power_ceiling.hip -o tools/ubench/_build/mfma_power_ceiling   (kernel: ccsmeth_amd/csrc/ccsm_ceiling.hip)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "../../ccsmeth_amd/csrc/ccsm_ceiling.hip"
using namespace ccsm_ceiling;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int T, int MODE>
static void run(const char* name, const uint4* rnd, float* out, double seconds, bool zero) {
    const int iters = 20000, grid = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<T, MODE><<<grid, T>>>(rnd, out, 200);
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    double ms_sum = 0; int launches = 0;
    printf("BEGIN %s waves_per_simd=%d operands=%s\n", name, T / 256, zero ? "zero" : "random"); fflush(stdout);
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < 4; ++r) k<T, MODE><<<grid, T>>>(rnd, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // the governor needs ~1 s to settle: average the second half only
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.5 * seconds) { ms_sum += ms; launches += 4; }
    }
    const double waves = (double)grid * (T / 64), per_it_f16 = 8.0 * 2 * 32 * 32 * 16, per_it_cyc = 8 * 32 + (MODE ? 4 * 35 : 0);
    const double sec = ms_sum * 1e-3;
    const double tf = launches * waves * iters * per_it_f16 / sec * 1e-12;
    // issue cycles per second and SIMD, as a clock: what sclk would have to be if the pipe never idled
    const double ghz_issue = launches * (waves / 1024.0) * iters * per_it_cyc / sec * 1e-9;
    printf("END %s waves_per_simd=%d operands=%s : %.1f TFLOP/s fp16-MFMA flops (%.3f of 2500), MFMA issue %.3f Gcycles/s per SIMD, %d launches of %.2f ms\n",
           name, T / 256, zero ? "zero" : "random", tf, tf / 2500.0, ghz_issue, launches, ms_sum / (launches ? launches : 1));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    std::vector<uint4> h = random_operands();
    uint4* rnd; float* out;
    CK(hipMalloc(&rnd, h.size() * sizeof(uint4))); CK(hipMalloc(&out, 256 * 512 * 4));
    for (int zero = 0; zero < 2; ++zero) {
        if (zero) CK(hipMemset(rnd, 0, h.size() * sizeof(uint4))); else CK(hipMemcpy(rnd, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
        run<256, 0>("f16", rnd, out, seconds, zero); run<512, 0>("f16", rnd, out, seconds, zero);
        run<256, 1>("mix", rnd, out, seconds, zero); run<512, 1>("mix", rnd, out, seconds, zero);
        run<512, 2>("lds", rnd, out, seconds, zero);
    }
    return 0;
}
