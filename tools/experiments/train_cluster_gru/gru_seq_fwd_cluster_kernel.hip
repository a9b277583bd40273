// Experiment (round 2), NOT part of libccsm_train: see README.md.  Was appended to ccsmeth_amd/csrc/ccsm_train_seq.hip.
// ---------------------------------------------------------------------------------------------------------------------------------
// Cluster variant: W_hh RESIDENT in registers, the hidden state exchanged between four workgroups per timestep.
//
// The kernel above streams 768 KiB of weight fragments per step and workgroup from L2 with two k-blocks of look-ahead: at the L2's
// loaded latency that is ~13 us of the ~15 us a step takes, whatever the batch size — and at the reference's batch size (512 sites =
// 32 row tiles per direction) seven eighths of the chip idle meanwhile.  Here a 32-row tile is served by a CLUSTER of four workgroups:
// member q owns hidden units [64q, 64q + 64) of all three gates, i.e. 192 of the 768 columns of W_hh^T = 96 VGPRs per lane of split
// fp16 fragments, loaded ONCE per launch.  Per step a member multiplies the whole h_{t-1} tile (LDS) with its columns, evaluates the
// gates of its 64 units, writes its slice of h_t (fp16 hi / lo) to a global exchange buffer, signals, waits for the other three
// members, and pulls their slices into its LDS copy of the tile.  Both directions of a layer run in the same launch.
//   wave w = (hf = w >> 2: unit half of the member, kq = w & 3: K quarter = k-blocks [4 kq, 4 kq + 4)): 3 gates x 4 k-blocks x 3
//     passes = 36 MFMAs per step; partial sums of the four K quarters meet in LDS;
//   gate arithmetic: thread (w, lane) owns rows 8 (w & 3) + 4 hh + e (e < 4) of unit 64 q + 32 (w >> 2) + (lane & 31) and keeps their
//     h in fp32 registers for the whole sequence; all global accesses are 128-byte row segments;
//   exchange: release fence + atomic counter per (cluster, step) at agent scope, bounded spin (a launch that could not make progress
//     sets *err instead of hanging), acquire fence, 16-byte loads; two exchange buffers alternate by step parity;
//   launched with hipLaunchCooperativeKernel so that all members are resident together (at most 32 row tiles per launch).
// LDS: h tile hi | lo 33 KiB + K-quarter partial sums 96 KiB.
constexpr int kCqTileHalfs = 32 * kSqRowHalfs;                   // one fp16 plane of the h tile
constexpr int kCqPartFloats = 2 * 4 * 3 * 16 * 64;               // [hf][kq][gate][register][lane]
constexpr int kCqLds = 2 * kCqTileHalfs * 2 + kCqPartFloats * 4;
constexpr int kCqXbufHalfs = 2 * 32 * H;                         // per (parity, cluster): [hi | lo][32 rows][256]
constexpr unsigned kCqSpinLimit = 4u << 20;

struct CqArgs {
    const float* gi[2];        // per direction: (T, M, 768)
    const float* h0[2];        // (M, 256)
    const float* w_hh[2];      // (768, 256) fp32 parameters
    const float* b_ih[2];
    const float* b_hh[2];
    float* out;                // (T, M, 512) of the layer (direction = column half)
    float* sav[2][4];          // R, Z, N, HP per direction: (T, M, 256)
    _Float16* xbuf;            // [2][clusters][kCqXbufHalfs]
    int* flags;                // [clusters][T], zero before the launch
    int* err;
    int M, row_base, tiles, save;
};

__global__ __launch_bounds__(512, 1) void gru_seq_fwd_cluster_kernel(CqArgs a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 cq_lds[];
    _Float16* t_hi = cq_lds;
    _Float16* t_lo = cq_lds + kCqTileHalfs;
    float* part = reinterpret_cast<float*>(cq_lds + 2 * kCqTileHalfs);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, hh = lane >> 5;
    const int q = blockIdx.x & 3, cl = blockIdx.x >> 2;             // member, cluster
    const int d = cl & 1, tile = cl >> 1;
    const int M = a.M;
    const int row0 = a.row_base + tile * 32;
    const int hf = wave >> 2, kq = wave & 3;

    // ---- this wave's resident B operands: gate g, k-block 4 kq + i: lane (j, hh) = split(W_hh[g * 256 + 64 q + 32 hf + j][16 kb + 8 hh + 0..7])
    sq_half8 w_hi[3][4], w_lo[3][4];
    {
        const float* W = a.w_hh[d];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* src = W + (size_t)(g * H + 64 * q + 32 * hf + j) * H + 16 * (4 * kq + i) + 8 * hh;
                const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
                const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    w_hi[g][i][k] = (_Float16)v[k];
                    w_lo[g][i][k] = (_Float16)(v[k] - (float)w_hi[g][i][k]);
                }
            }
    }
    // ---- gate-arithmetic ownership: rows 8 rs + 4 hh + e of unit ug
    const int rs = wave & 3;
    const int ul = 32 * hf + j;                                     // unit within the member's 64
    const int ug = 64 * q + ul;                                     // unit within the layer direction's 256
    const int lrow = 8 * rs + 4 * hh;                               // first of this thread's 4 tile rows
    float h[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int m = row0 + lrow + e;
        h[e] = m < M ? a.h0[d][(size_t)m * H + ug] : 0.f;
    }
    // the whole h0 tile (all 256 units) into LDS: thread i -> row i >> 4, 16 units
    {
        const int row = threadIdx.x >> 4, c0 = (threadIdx.x & 15) * 16;
        const int m = row0 + row;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float v = m < M ? a.h0[d][(size_t)m * H + c0 + k] : 0.f;
            const _Float16 x = (_Float16)v;
            t_hi[row * kSqRowHalfs + c0 + k] = x;
            t_lo[row * kSqRowHalfs + c0 + k] = (_Float16)(v - (float)x);
        }
    }
    const float bir = a.b_ih[d][ug] + a.b_hh[d][ug], biz = a.b_ih[d][H + ug] + a.b_hh[d][H + ug], bin = a.b_ih[d][2 * H + ug],
                bhn = a.b_hh[d][2 * H + ug];
    auto opaque = [](unsigned v) { asm volatile("" : "+v"(v)); return v; };
    const int rows_left = M - (row0 + lrow);
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = d ? T - 1 - s : s;
        // this step's input projections for the thread's 4 elements: requested first, used after the products
        float gr[4], gz[4], gn[4];
        {
            const float* gt = a.gi[d] + (size_t)t * M * G;
            const unsigned o = opaque((unsigned)(row0 + lrow) * G + ug);
#pragma unroll
            for (int e = 0; e < 4; ++e) { gr[e] = gt[o + e * G]; gz[e] = gt[o + e * G + H]; gn[e] = gt[o + e * G + 2 * H]; }
        }
        // ---- products of this wave's K quarter
        sq_f32x16 acc[3];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; acc[2][r] = 0.f; }
        {
            const _Float16* ph = t_hi + j * kSqRowHalfs + 8 * hh + 64 * kq;
            const _Float16* pl = t_lo + j * kSqRowHalfs + 8 * hh + 64 * kq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const sq_half8 a_hi = *reinterpret_cast<const sq_half8*>(ph + 16 * i);
                const sq_half8 a_lo = *reinterpret_cast<const sq_half8*>(pl + 16 * i);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    acc[g] = sq_mfma(a_hi, w_hi[g][i], acc[g]);
                    acc[g] = sq_mfma(a_lo, w_hi[g][i], acc[g]);
                    acc[g] = sq_mfma(a_hi, w_lo[g][i], acc[g]);
                }
            }
        }
        {
            float* pw = part + (size_t)((hf * 4 + kq) * 3) * 16 * 64 + lane;
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) pw[(g * 16 + r) * 64] = acc[g][r];
        }
        __syncthreads();                                            // A: every product of the step is in LDS; the h tile is free to change
        // ---- gates of this thread's 4 elements
        const int par = s & 1;
        _Float16* xb = a.xbuf + ((size_t)par * (gridDim.x >> 2) + cl) * kCqXbufHalfs;
        {
            float* ot = a.out + (size_t)t * M * H2 + d * H;
            const size_t st = (size_t)t * M * H;
            const unsigned oo = opaque((unsigned)(row0 + lrow) * H2 + ug), os = opaque((unsigned)(row0 + lrow) * H + ug);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * rs + e;
                float pr = 0.f, pz = 0.f, pn = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float* pp = part + (size_t)((hf * 4 + k) * 3) * 16 * 64 + lane;
                    pr += pp[(0 * 16 + r) * 64];
                    pz += pp[(1 * 16 + r) * 64];
                    pn += pp[(2 * 16 + r) * 64];
                }
                const float rr = sigmoidf_(gr[e] + pr + bir);
                const float zz = sigmoidf_(gz[e] + pz + biz);
                const float hp = pn + bhn;
                const float nn = tanhf(gn[e] + bin + rr * hp);
                h[e] = (1.0f - zz) * nn + zz * h[e];
                const _Float16 x = (_Float16)h[e];
                const _Float16 y = (_Float16)(h[e] - (float)x);
                const int row = lrow + e;
                t_hi[row * kSqRowHalfs + ug] = x;
                t_lo[row * kSqRowHalfs + ug] = y;
                xb[row * H + ug] = x;
                xb[32 * H + row * H + ug] = y;
                if (e < rows_left) {
                    ot[oo + (unsigned)e * H2] = h[e];
                    if (a.save) {
                        const unsigned o = os + (unsigned)e * H;
                        (a.sav[d][0] + st)[o] = rr; (a.sav[d][1] + st)[o] = zz; (a.sav[d][2] + st)[o] = nn; (a.sav[d][3] + st)[o] = hp;
                    }
                }
            }
        }
        if (s + 1 == T) break;                                      // nobody needs the last state's exchange
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // this thread's slice stores are visible device-wide ...
        __syncthreads();                                            // B: ... for every thread of the member
        if (threadIdx.x == 0) {
            int* f = a.flags + (size_t)cl * T + s;
            __hip_atomic_fetch_add(f, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            unsigned n = 0;
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 4) {
                if (++n > kCqSpinLimit) { *a.err = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();                                            // C: all four slices of h_t are in the exchange buffer
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        {   // the other members' slices: one 16-byte chunk per thread and member: plane i >> 8, row (i & 255) >> 3, chunk i & 7
            const int i = threadIdx.x, plane = i >> 8, row = (i & 255) >> 3, ch = i & 7;
            _Float16* dst = (plane ? t_lo : t_hi) + row * kSqRowHalfs + 8 * ch;
            const _Float16* src = xb + plane * 32 * H + row * H + 8 * ch;
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const int m2 = (q + k) & 3;
                *reinterpret_cast<uint4*>(dst + 64 * m2) = *reinterpret_cast<const uint4*>(src + 64 * m2);
            }
        }
        __syncthreads();                                            // D: the tile holds h_t
    }
}
