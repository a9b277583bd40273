// libccsm_train: persistent recurrent kernels — all 21 timesteps of one GRU layer and direction in ONE launch, gate arithmetic fused.
// Included by ccsm_train.hip inside its anonymous namespace (uses T, H, G, H2, sigmoidf_).
//
// What they replace (train_multigpu.py:283-286 -> torch.nn.GRU forward / backward): per timestep one rocBLAS product
// h_{t-1} W_hh^T (M x 768 x 256) plus one gate kernel — 126 dependent launch pairs per forward, each 8-35 us, latency- rather than
// throughput-bound at the reference's batch sizes.  Here a workgroup owns 32 batch rows for the whole sequence:
//   * arithmetic: every fp32 operand is split into fp16 hi + fp16 lo and the product taken as hi*hi + lo*hi + hi*lo on
//     v_mfma_f32_32x32x16_f16 with fp32 accumulation (the dropped lo*lo term is 2^-22 relative): fp32-class results (the training
//     parity tests keep their tolerances) at the fp16 matrix rate instead of the 16x slower fp32 one;
//   * orientation: batch rows are the MFMA's M, hidden / gate units its N, so a result register is 32 CONSECUTIVE units of one row
//     — every global access of the epilogue (gi, out, the saved r / z / n / W_hn h) is a 128-byte coalesced row segment of the
//     (T, M, features) fp32 arrays the rest of the library uses;
//   * the state: each lane keeps h_t of its own (row, unit) pairs in fp32 registers for the whole sequence (the recurrence itself
//     is exact fp32); the fp16 hi / lo copy every wave needs as the next step's A operand goes through LDS (row-major, padded rows,
//     double-buffered: one barrier per step);
//   * W_hh streams from L2 as pre-split B-operand fragments (768 KiB per step and workgroup), packed from the fp32 parameters by
//     pack_whh_kernel at the start of every forward (the optimiser changes them every step).
// Wave w of 8 owns hidden units [32w, 32w + 32) of all three gates.

typedef _Float16 sq_half8 __attribute__((ext_vector_type(8)));
typedef float sq_f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSqRowHalfs = H + 8;                       // LDS row stride of the fp16 state copies: 528 B, conflict-free 16-byte row reads
constexpr int kSqBufHalfs = 32 * kSqRowHalfs;
constexpr int kSqLds = 2 * 2 * kSqBufHalfs * 2;          // [buffer][hi | lo][32 rows][264] halfs = 67584 B
constexpr int kSqFragPerDir = 8 * (H / 16) * 3 * 2 * 64; // uint4 per (layer, direction): [wave][kb][gate][hi | lo][lane]

__device__ __forceinline__ sq_f32x16 sq_mfma(sq_half8 a, sq_half8 b, sq_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// B-operand fragments of W_hh (768 x 256, fp32 row-major): fragment (wave, kb, gate, hl), lane (j, g) = the 8 halfs
// split(W_hh[gate * 256 + 32 wave + j][16 kb + 8 g + 0..7]).  One thread per (fragment pair, lane).
__global__ void pack_whh_kernel(const float* __restrict__ w, uint4* __restrict__ frag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // ((wave * 16 + kb) * 3 + gate) * 64 + lane
    if (i >= 8 * (H / 16) * 3 * 64) return;
    const int lane = i & 63, gate = (i >> 6) % 3, kb = ((i >> 6) / 3) % (H / 16), wave = (i >> 6) / (3 * (H / 16));
    const float* src = w + (size_t)(gate * H + 32 * wave + (lane & 31)) * H + 16 * kb + 8 * (lane >> 5);
    sq_half8 hi, lo;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float v = src[k];
        hi[k] = (_Float16)v;
        lo[k] = (_Float16)(v - (float)hi[k]);
    }
    const size_t o = (size_t)(i >> 6) * 2 * 64 + lane;
    frag[o] = __builtin_bit_cast(uint4, hi);
    frag[o + 64] = __builtin_bit_cast(uint4, lo);
}

// Forward.  gi: (T, M, 768) = x_t W_ih^T (no bias); h0: (M, 256); out: (T, M, 512) + direction column offset already applied;
// R, Z, Nn, HP: (T, M, 256) saved for the backward pass when `save`.  grid = ceil(M / 32), block = 512.
__global__ __launch_bounds__(512, 1) void gru_seq_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ h0,
                                                            const uint4* __restrict__ wfrag, const float* __restrict__ b_ih,
                                                            const float* __restrict__ b_hh, float* __restrict__ out, float* __restrict__ R,
                                                            float* __restrict__ Z, float* __restrict__ Nn, float* __restrict__ HP, int M,
                                                            int reverse, int save) {
    extern __shared__ __attribute__((aligned(16))) _Float16 sq_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, hh = lane >> 5;
    const int row0 = blockIdx.x * 32;
    const int u = 32 * wave + j;                                   // this lane's hidden unit
    // register r = 4q + e of an accumulator is batch row 8q + 4hh + e of the tile
    auto row_of = [&](int r) { return 8 * (r >> 2) + 4 * hh + (r & 3); };

    float h[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = row0 + row_of(r);
        h[r] = m < M ? h0[(size_t)m * H + u] : 0.f;
    }
    auto publish = [&](int buf) {                                  // this lane's 16 (row, u) values -> fp16 hi / lo copies in LDS
        _Float16* hi = sq_lds + (size_t)buf * 2 * kSqBufHalfs;
        _Float16* lo = hi + kSqBufHalfs;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const _Float16 a = (_Float16)h[r];
            hi[row_of(r) * kSqRowHalfs + u] = a;
            lo[row_of(r) * kSqRowHalfs + u] = (_Float16)(h[r] - (float)a);
        }
    };
    publish(0);
    const float bir = b_ih[u] + b_hh[u], biz = b_ih[H + u] + b_hh[H + u], bin = b_ih[2 * H + u], bhn = b_hh[2 * H + u];
    const uint4* wf = wfrag + (size_t)wave * (H / 16) * 3 * 2 * 64 + lane;
    __syncthreads();

    // input projections of one step (three gates x 16 rows per lane) are requested one step AHEAD, behind the k-block loop of the step
    // before, and folded into the accumulators' initial values: they are never live next to the weight fragments
    float nr[16], nz[16], nn_[16];
    // A lane's 16 rows are row0 + 4 hh + c, c = 8 q + e a compile-time constant per register: every per-lane address is ONE 32-bit
    // offset (made opaque once per step, or the compiler hoists 16 x 5 row addresses out of the step loop and spills them) plus a
    // constant.  Loads are not guarded: the arrays are allocated with 32 rows of slack and batch rows never mix.
    const int lrow = row0 + 4 * hh;
    auto opaque = [](unsigned v) { asm volatile("" : "+v"(v)); return v; };
    auto gi_load = [&](int s) {
        const int t = reverse ? T - 1 - s : s;
        const float* gt = gi + (size_t)t * M * G;                      // wave-uniform base
        const unsigned o = opaque((unsigned)lrow * G + u);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned c = (unsigned)(8 * (r >> 2) + (r & 3)) * G;
            nr[r] = gt[o + c];
            nz[r] = gt[o + c + H];
            nn_[r] = gt[o + c + 2 * H];
        }
    };
    gi_load(0);
    for (int s = 0; s < T; ++s) {
        const int t = reverse ? T - 1 - s : s;
        const int cur = s & 1;
        sq_f32x16 acc[3];
        float gn[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = nr[r] + bir; acc[1][r] = nz[r] + biz; acc[2][r] = bhn; gn[r] = nn_[r] + bin; }
        const _Float16* hi = sq_lds + (size_t)cur * 2 * kSqBufHalfs + j * kSqRowHalfs + 8 * hh;
        const _Float16* lo = hi + kSqBufHalfs;
        // weight fragments two k-blocks ahead (three register sets; a fourth spills: the stream is latency-bound, 768 KiB per step and
        // workgroup from L2); the k-block loop stays rolled: fully unrolled, the compiler hoists all 96 fragment loads to the top of the
        // step and spills them
        uint4 wq[3][6];
        auto wload = [&](uint4 (&w)[6], int kb) {
#pragma unroll
            for (int f = 0; f < 6; ++f) w[f] = wf[(size_t)(kb * 6 + f) * 64];
        };
        auto kblock = [&](int kb, const uint4 (&w)[6]) {
            const sq_half8 a_hi = *reinterpret_cast<const sq_half8*>(hi + 16 * kb);
            const sq_half8 a_lo = *reinterpret_cast<const sq_half8*>(lo + 16 * kb);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const sq_half8 w_hi = __builtin_bit_cast(sq_half8, w[2 * g]);
                const sq_half8 w_lo = __builtin_bit_cast(sq_half8, w[2 * g + 1]);
                acc[g] = sq_mfma(a_hi, w_hi, acc[g]);
                acc[g] = sq_mfma(a_lo, w_hi, acc[g]);
                acc[g] = sq_mfma(a_hi, w_lo, acc[g]);
            }
        };
        asm volatile("" ::: "memory");
        wload(wq[0], 0); wload(wq[1], 1);
#pragma unroll 1
        for (int kb = 0; kb + 3 <= H / 16; kb += 3) {                  // k-blocks 0..14, three per trip, loads two k-blocks ahead
            wload(wq[2], kb + 2);
            kblock(kb, wq[0]);
            wload(wq[0], kb + 3);
            kblock(kb + 1, wq[1]);
            wload(wq[1], kb + 4 < H / 16 ? kb + 4 : H / 16 - 1);       // past the end: a harmless reload
            kblock(kb + 2, wq[2]);
        }
        kblock(H / 16 - 1, wq[0]);
        asm volatile("" ::: "memory");                                 // memory operations stay on their side: the compiler otherwise hoists the
        if (s + 1 < T) gi_load(s + 1);                                 // next step's 48 loads above the k-block loop and spills
        asm volatile("" ::: "memory");
        // gates (torch.nn.GRU cell): r = s(gi_r + b_ir + gh_r + b_hr), z likewise, hp = gh_n + b_hn, n = tanh(gi_n + b_in + r hp),
        // h = (1 - z) n + z h_{t-1}
        float* ot = out + (size_t)t * M * H2;
        const size_t st = (size_t)t * M * H;
        const unsigned oo = opaque((unsigned)lrow * H2 + u), os = opaque((unsigned)lrow * H + u);
        const int rows_left = M - lrow;                                // rows of this lane's group that exist
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = 8 * (r >> 2) + (r & 3);
            const float rr = sigmoidf_(acc[0][r]);
            const float zz = sigmoidf_(acc[1][r]);
            const float hp = acc[2][r];
#ifdef CCSM_SEQ_FAST_TANH
            const float nn = 1.0f - 2.0f / (__expf(2.0f * (gn[r] + rr * hp)) + 1.0f);
#else
            const float nn = tanhf(gn[r] + rr * hp);
#endif
            h[r] = (1.0f - zz) * nn + zz * h[r];
            if (c < rows_left) {
                ot[oo + (unsigned)c * H2] = h[r];
                if (save) {
                    const unsigned o = os + (unsigned)c * H;
                    (R + st)[o] = rr; (Z + st)[o] = zz; (Nn + st)[o] = nn; (HP + st)[o] = hp;
                }
            }
        }
        publish(cur ^ 1);
        __syncthreads();
    }
}
