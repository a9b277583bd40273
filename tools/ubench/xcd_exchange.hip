// Can the workgroups of ONE XCD exchange a GRU state every timestep through their shared L2 cheaply?
//
// the attbigru2s GRU kernels stream 2 MB of weights per workgroup and timestep because a workgroup owns all 768 gate
// rows of its 96 batch rows.  The alternative - a CLUSTER of 8 workgroups that each keep 1/8 of the weights resident and exchange the
// 256-unit state every step - died in round 2 on the cost of an agent-scope release / acquire (an L2 write-back + invalidate: the eight
// XCD L2s are not coherent with each other; 86 us per step in tools/experiments/train_cluster_gru).  Workgroups are dealt to the XCDs
// round-robin (workgroup i -> XCD i mod 8), so a cluster made of workgroups i, i + 8, i + 16, ... shares one L2: no write-back is
// needed if (a) the data stores and the flag reach L2 (stores write through the CU's L1), and (b) the consumers' loads bypass their L1
// (sc1 on the load), with plain s_waitcnt ordering in between and NO fence instruction.  This probe measures that exchange:
//   each of 256 workgroups (one per CU) per step: writes its 12 KiB slice, waits for its stores, bumps the cluster's counter (an L2
//   atomic), polls the counter until all 8 members have arrived, reads the cluster's 96 KiB back, checks one word per slice;
// for clusters inside an XCD (members 8 apart) and, for contrast, clusters across XCDs (8 consecutive workgroups) with agent-scope
// fences.  Every poll loop is bounded: a protocol that does not work reports errors / time-outs instead of hanging the GPU.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/xcd_exchange.hip -o tools/ubench/_build/xcd_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kMembers = 8, kSliceU4 = 12 * 1024 / 16, kThreads = 512;     // 12 KiB per member and step = 96 rows x 32 units x (hi + blob)

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_l2(uint4* p, uint4 v) {       // through the L1 to the XCD's L2
    const u32x4v r = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ uint4 load_l2(const uint4* p) {          // past the L1
    u32x4v r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return make_uint4(r.x, r.y, r.z, r.w);
}

// MODE 0: same-XCD clusters, no fences.  MODE 1: cross-XCD clusters (consecutive workgroups), agent-scope release / acquire fences.
template <int MODE>
__global__ __launch_bounds__(kThreads) void exchange(uint4* __restrict__ buf, unsigned* __restrict__ counters, int steps, unsigned long long* cycles,
                                                     unsigned* errors) {
    const int wg = blockIdx.x;
    int cluster, member;
    if (MODE == 0) { const int xcd = wg & 7, local = wg >> 3; cluster = xcd * 4 + (local >> 3); member = local & 7; }
    else { cluster = wg >> 3; member = wg & 7; }
    unsigned* ctr = counters + cluster * 32;                          // one counter per 128-byte line
    uint4* cbuf = buf + (size_t)cluster * 2 * kMembers * kSliceU4;     // [parity][member][slice]
    unsigned bad = 0, timeouts = 0;
    __shared__ int s_abort;
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        if (s_abort) break;                                               // a protocol that stalls gives up after a few time-outs
        uint4* mine = cbuf + ((size_t)(s & 1) * kMembers + member) * kSliceU4;
        const unsigned tag = (unsigned)(s * 64 + member + 1);
        for (int i = threadIdx.x; i < kSliceU4; i += kThreads) {
            const uint4 v = make_uint4(tag, (unsigned)i, tag ^ 0x5a5a5a5au, (unsigned)wg);
            if (MODE == 0) store_l2(mine + i, v); else mine[i] = v;
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's stores are at the L2
        else __atomic_thread_fence(__ATOMIC_RELEASE);                      // (agent scope: writes the L2 back)
        __syncthreads();                                                  // ... and so are every thread's
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(s + 1) * kMembers;
            int polls = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && polls < (1 << 18)) { ++polls; __builtin_amdgcn_s_sleep(1); }
            if (polls >= (1 << 18)) { ++timeouts; if (timeouts > 2) s_abort = 1; }
        }
        __syncthreads();
        if (MODE == 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);           // (agent scope: invalidates)
        const uint4* all = cbuf + (size_t)(s & 1) * kMembers * kSliceU4;
        unsigned acc = 0;
        for (int i = threadIdx.x; i < kMembers * kSliceU4; i += kThreads) {
            const uint4 v = MODE == 0 ? load_l2(all + i) : all[i];
            const unsigned m = (unsigned)(i / kSliceU4);
            acc |= (v.x != (unsigned)(s * 64 + m + 1)) | (v.y != (unsigned)(i % kSliceU4));
        }
        bad += acc;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (bad) atomicAdd(errors, 1u);
    if (timeouts) atomicAdd(errors + 1, timeouts);
    if (threadIdx.x == 0) cycles[wg] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int steps) {
    const int grid = 256, clusters = grid / kMembers;
    uint4* buf; unsigned* ctr; unsigned long long* cyc; unsigned* err;
    CK(hipMalloc(&buf, (size_t)clusters * 2 * kMembers * kSliceU4 * sizeof(uint4)));
    CK(hipMalloc(&ctr, clusters * 32 * sizeof(unsigned)));
    CK(hipMalloc(&cyc, grid * sizeof(unsigned long long)));
    CK(hipMalloc(&err, 2 * sizeof(unsigned)));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(ctr, 0, clusters * 32 * sizeof(unsigned)));
        CK(hipMemset(err, 0, 2 * sizeof(unsigned)));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        exchange<MODE><<<grid, kThreads>>>(buf, ctr, steps, cyc, err);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(grid); unsigned he[2];
        CK(hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        CK(hipMemcpy(he, err, sizeof(he), hipMemcpyDeviceToHost));
        unsigned long long mx = 0; for (auto c : h) mx = c > mx ? c : mx;
        if (rep == 1)
            printf("%-34s %d steps: %.3f ms = %.2f us per step (%.0f cycle-counter ticks per step on the slowest workgroup); workgroups with "
                   "wrong data: %u, poll time-outs: %u\n", name, steps, ms, 1e3 * ms / steps, (double)mx / steps, he[0], he[1]);
    }
    CK(hipFree(buf)); CK(hipFree(ctr)); CK(hipFree(cyc)); CK(hipFree(err));
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2000;
    // all 256 workgroups must be resident at once (they wait for each other): one 512-thread workgroup per CU
    run<0>("same-XCD clusters, sc1, no fences", steps);
    run<1>("cross-XCD clusters, agent fences", steps);
    return 0;
}
