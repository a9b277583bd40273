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
        // weight fragments three k-blocks ahead (four register sets: the stream is latency-bound, 768 KiB per step and workgroup from
        // L2); the k-block loop stays rolled: fully unrolled, the compiler hoists all 96 fragment loads to the top of the
        // step and spills them
        uint4 wq[4][6];
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
        wload(wq[0], 0); wload(wq[1], 1); wload(wq[2], 2);
#pragma unroll 1
        for (int kb = 0; kb < H / 16; kb += 4) {                       // four k-blocks per trip, loads three k-blocks ahead
            constexpr int last = H / 16 - 1;
            wload(wq[3], kb + 3);
            kblock(kb, wq[0]);
            wload(wq[0], kb + 4 < last ? kb + 4 : last);               // past the end: a harmless reload
            kblock(kb + 1, wq[1]);
            wload(wq[1], kb + 5 < last ? kb + 5 : last);
            kblock(kb + 2, wq[2]);
            wload(wq[2], kb + 6 < last ? kb + 6 : last);
            kblock(kb + 3, wq[3]);
        }
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
            const float nn = tanhf(gn[r] + rr * hp);
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward.  Per timestep (in the reverse of the forward order) and lane, for its 16 (row, unit) pairs:
//   dh = d out_t + dh_{t+1} z_{t+1} + (dgh_{t+1} W_hh)            (the product of the step before, straight from the accumulator)
//   dn = dh (1 - z)(1 - n^2); dz = dh (h_{t-1} - n) z (1 - z); dr = dn hp r (1 - r)
//   dgi_t = [dr, dz, dn], dgh_t = [dr, dz, dn r] -> global (the weight-gradient products read them), dgh_t also as fp16 hi / lo into
//   LDS = the A operand (32 rows x 768) of this step's product with W_hh (K = 768: 48 k-blocks, wave w owns units [32w, 32w + 32)).
// One accumulator tile per wave leaves room for eleven k-blocks of weight fragments in flight (the forward kernel: three); the next
// step's 96 input values per lane are requested behind the product loop, in the registers the fragments vacate.
// The bias gradients (column sums of dgi / dgh over rows and steps) are accumulated in registers on the way: no separate pass.
//   dO, hout: (T, M, 512) with the direction's column offset applied; hout = the layer's outputs (h_{t-1} of a step);
//   wt: B-operand fragments of W_hh^T: [wave][kb 48][hi | lo][lane], lane (j, g) = split(W_hh[16 kb + 8 g + 0..7][32 wave + j]).
constexpr int kSbRowHalfs = G + 8;                               // 1552 B rows: 16-byte aligned, conflict-free
constexpr int kSbLds = 2 * 32 * kSbRowHalfs * 2;                 // [hi | lo][32 rows][776] halfs = 99328 B
constexpr float kSbScale = 4096.0f;

__global__ void pack_whh_t_kernel(const float* __restrict__ w, uint4* __restrict__ frag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (wave * 48 + kb) * 64 + lane
    if (i >= 8 * (G / 16) * 64) return;
    const int lane = i & 63, kb = (i >> 6) % (G / 16), wave = (i >> 6) / (G / 16);
    const float* src = w + (size_t)(16 * kb + 8 * (lane >> 5)) * H + 32 * wave + (lane & 31);
    sq_half8 hi, lo;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float v = src[(size_t)k * H];
        hi[k] = (_Float16)v;
        lo[k] = (_Float16)(v - (float)hi[k]);
    }
    const size_t o = (size_t)(i >> 6) * 2 * 64 + lane;
    frag[o] = __builtin_bit_cast(uint4, hi);
    frag[o + 64] = __builtin_bit_cast(uint4, lo);
}

__global__ __launch_bounds__(512, 1) void gru_seq_bwd_kernel(const float* __restrict__ dO, const float* __restrict__ hout,
                                                            const float* __restrict__ h0, const uint4* __restrict__ wt,
                                                            const float* __restrict__ R, const float* __restrict__ Z,
                                                            const float* __restrict__ Nn, const float* __restrict__ HP,
                                                            float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ db_part,
                                                            int M, int reverse, int* __restrict__ saturated) {
    extern __shared__ __attribute__((aligned(16))) _Float16 sb_lds[];
    _Float16* t_hi = sb_lds;
    _Float16* t_lo = sb_lds + 32 * kSbRowHalfs;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, hh = lane >> 5;
    const int row0 = blockIdx.x * 32;
    const int u = 32 * wave + j;
    const int lrow = row0 + 4 * hh;                                 // the lane's rows are lrow + c, c = 8 q + e
    const int rows_left = M - lrow;
    auto opaque = [](unsigned v) { asm volatile("" : "+v"(v)); return v; };
    const uint4* wf = wt + (size_t)wave * (G / 16) * 2 * 64 + lane;

    float carry[16], mm[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { carry[r] = 0.f; mm[r] = 0.f; }
    float s_dr = 0.f, s_dz = 0.f, s_dn = 0.f, s_dnr = 0.f;          // bias gradients = column sums of dgi / dgh over rows and steps
    // inputs of one step: requested ahead (behind the previous step's first weight requests), all unguarded (32 rows of slack)
    float xd[16], xr[16], xz[16], xn[16], xp[16], xh[16];
    auto in_load = [&](int s) {
        const int tt = reverse ? T - 1 - s : s;
        const unsigned o2 = opaque((unsigned)lrow * H2 + u), o1 = opaque((unsigned)lrow * H + u);
        const float* dt = dO + (size_t)tt * M * H2;
        const size_t so = (size_t)tt * M * H;
        const float* hp_ = s == 0 ? h0 : hout + (size_t)(reverse ? tt + 1 : tt - 1) * M * H2;
        const unsigned oh = s == 0 ? o1 : o2, sh = s == 0 ? H : H2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned c = (unsigned)(8 * (r >> 2) + (r & 3));
            xd[r] = dt[o2 + c * H2];
            xr[r] = (R + so)[o1 + c * H];
            xz[r] = (Z + so)[o1 + c * H];
            xn[r] = (Nn + so)[o1 + c * H];
            xp[r] = (HP + so)[o1 + c * H];
            xh[r] = hp_[oh + c * sh];
        }
    };
    in_load(T - 1);
    for (int s = T - 1; s >= 0; --s) {
        const int tt = reverse ? T - 1 - s : s;
        // ---- gates of this lane's 16 elements
        {
            float* gi_t = dgi + (size_t)tt * M * G;
            float* gh_t = dgh + (size_t)tt * M * G;
            const unsigned og = opaque((unsigned)lrow * G + u);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 8 * (r >> 2) + (r & 3);
                const float dh = xd[r] + carry[r] + mm[r];
                const float rr = xr[r], zz = xz[r], nn = xn[r];
                const float dn = dh * (1.0f - zz) * (1.0f - nn * nn);
                const float dz = dh * (xh[r] - nn) * zz * (1.0f - zz);
                const float dr = dn * xp[r] * rr * (1.0f - rr);
                const float dnr = dn * rr;
                carry[r] = dh * zz;
                if (c < rows_left) {
                    const unsigned o = og + (unsigned)c * G;
                    gi_t[o] = dr; gi_t[o + H] = dz; gi_t[o + 2 * H] = dn;
                    gh_t[o] = dr; gh_t[o + H] = dz; gh_t[o + 2 * H] = dnr;
                    s_dr += dr; s_dz += dz; s_dn += dn; s_dnr += dnr;
                }
                const int row = 4 * hh + c;
                // gradients are small (a mean over the batch): scaled by 2^12 into fp16's normal range before the split (a value of 1e-7
                // would otherwise be a subnormal hi with no lo), saturated instead of overflowing (and flagged); the product is scaled back
                const float v3[3] = {dr, dz, dnr};
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const float vs = v3[g] * kSbScale;
                    const float v = fminf(fmaxf(vs, -60000.f), 60000.f);
                    // |gate gradient| > 14.6: the recurrent product would silently lose it.  The host repeats the backward pass step by
                    // step in fp32 (rocBLAS) when this flag is up (NaNs raise it too)
                    if (!(fabsf(vs) <= 60000.f) && c < rows_left) *saturated = 1;
                    const _Float16 x = (_Float16)v;
                    t_hi[row * kSbRowHalfs + g * H + u] = x;
                    t_lo[row * kSbRowHalfs + g * H + u] = (_Float16)(v - (float)x);
                }
            }
        }
        if (s == 0) break;                                          // d h0 is not a parameter gradient
        __syncthreads();                                            // the dgh tile is complete
        // ---- mm = dgh_t W_hh for this wave's 32 units: 48 k-blocks, fragments eleven k-blocks ahead (22 KiB per wave in flight: the stream is latency-bound)
        sq_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const _Float16* ph = t_hi + j * kSbRowHalfs + 8 * hh;
        const _Float16* pl = t_lo + j * kSbRowHalfs + 8 * hh;
        uint4 wq[12][2];
        auto wload = [&](uint4 (&w)[2], int kb) { w[0] = wf[(size_t)(kb * 2) * 64]; w[1] = wf[(size_t)(kb * 2 + 1) * 64]; };
        auto kblock = [&](int kb, const uint4 (&w)[2]) {
            const sq_half8 a_hi = *reinterpret_cast<const sq_half8*>(ph + 16 * kb);
            const sq_half8 a_lo = *reinterpret_cast<const sq_half8*>(pl + 16 * kb);
            const sq_half8 w_hi = __builtin_bit_cast(sq_half8, w[0]);
            const sq_half8 w_lo = __builtin_bit_cast(sq_half8, w[1]);
            acc = sq_mfma(a_hi, w_hi, acc);
            acc = sq_mfma(a_lo, w_hi, acc);
            acc = sq_mfma(a_hi, w_lo, acc);
        };
        asm volatile("" ::: "memory");
#pragma unroll
        for (int b = 0; b < 11; ++b) wload(wq[b], b);
        asm volatile("" ::: "memory");
#pragma unroll 1
        for (int kb = 0; kb < G / 16; kb += 12) {
            constexpr int last = G / 16 - 1;
#pragma unroll
            for (int b = 0; b < 12; ++b) {
                const int nx = kb + b + 11;
                wload(wq[(b + 11) % 12], nx < last ? nx : last);    // past the end: a harmless reload
                kblock(kb + b, wq[b]);
            }
        }
        asm volatile("" ::: "memory");
        in_load(s - 1);                                             // the next step's inputs (their registers held weight fragments until here)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) mm[r] = acc[r] * (1.0f / kSbScale);
        __syncthreads();                                            // every wave has read the tile before the next step overwrites it
    }
    // bias gradients: this workgroup's sums of its 32 rows, [workgroup][dr | dz | dn | dn r][unit]; seq_bias_reduce_kernel adds the
    // workgroups in a fixed order (no float atomics: the same gradients every run).  Both half-waves hold the same unit.
    s_dr += __shfl_xor(s_dr, 32, 64); s_dz += __shfl_xor(s_dz, 32, 64); s_dn += __shfl_xor(s_dn, 32, 64); s_dnr += __shfl_xor(s_dnr, 32, 64);
    if (hh == 0) {
        float* p = db_part + (size_t)blockIdx.x * 4 * H + u;
        p[0] = s_dr; p[H] = s_dz; p[2 * H] = s_dn; p[3 * H] = s_dnr;
    }
}
// db_ih = [sum dr | sum dz | sum dn], db_hh = [sum dr | sum dz | sum dn r] over the workgroups of gru_seq_bwd_kernel, in workgroup order
__global__ void seq_bias_reduce_kernel(const float* __restrict__ part, int nwg, float* __restrict__ db_ih, float* __restrict__ db_hh) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;           // 4 H values
    if (i >= 4 * H) return;
    float acc = 0.f;
    for (int w = 0; w < nwg; ++w) acc += part[(size_t)w * 4 * H + i];
    const int k = i / H, u = i - k * H;
    if (k < 3) db_ih[k * H + u] = acc;
    if (k < 2) db_hh[k * H + u] = acc;
    if (k == 3) db_hh[2 * H + u] = acc;
}
