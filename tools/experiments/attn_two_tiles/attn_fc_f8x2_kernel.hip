// Goes into ccsmeth_amd/csrc/ccsm_gru_f8.hip behind attn_fc_f8_kernel (namespace ccsm); launch and LDS size: README.md
// ---------------------------------------------------------------------------------------------------------
// Round 3: the same pool with TWO row tiles per workgroup and 64 attention units per wave.
// attn_fc_f8_kernel above is bound by its LDS operand reads: a wave owns 32 units, so every activation fragment it reads from LDS
// feeds ONE main MFMA (1 KiB per 32 cycles and wave; 330 B per clock and CU asked of a 128 B per clock LDS).  Here waves 0-3 take row
// tile 0 and waves 4-7 row tile 1 of the workgroup, each wave the units [64 ug, 64 ug + 64) of all 256: a fragment read feeds two
// MFMAs, half the LDS bytes per flop.  The price is registers - two accumulator tiles per timestep - so a pass holds TG = 3
// timesteps (96 accumulator registers) and Ua is streamed seven times per workgroup instead of three (from L2, 64 rows per pass
// instead of 32: the same bytes per row).  Staging, the split-f8 products, the fc1 partials (wave ug < 3 takes timestep t0 + ug of its
// row tile) and the fixed-order reductions are as above.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void attn_fc_f8x2_kernel(const uint4* __restrict__ out2, const uint4* __restrict__ wa,
                                                               const uint4* __restrict__ ua, const float* __restrict__ va,
                                                               const float* __restrict__ fcw, float* __restrict__ part,
                                                               SliceTable slices, int sa_wa, int sa_ua, int n_tiles) {
    constexpr int TG = 3;                      // timesteps per Ua pass (21 = 7 * 3)
    constexpr int CK = 2;                      // k-blocks per staged chunk = one pair
    constexpr int RT = 2;                      // row tiles per workgroup
    constexpr int NCHUNK = kKB12 / CK;
    constexpr int CHUNK_FRAGS = RT * CK * TG * 2;   // 24 fragments of 1 KiB
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_stage = smem;                                                      // [2][RT][CK][TG][hi|corr] fragments
    float* s_epart = reinterpret_cast<float*>(smem + 2 * CHUNK_FRAGS * 1024);   // [ug 4][tt][RT][32]: this timestep group's score partials
    float* s_e = s_epart + 4 * TG * RT * 32;                                    // [RT][t][32]
    float* s_pfc = s_e + RT * kSeqLen * 32;                                     // [RT][t][32][2]
    float* s_fcw = s_pfc + RT * kSeqLen * 32 * 2;                               // [2][1024]
    float* s_va = s_fcw + kClasses * 4 * kHidden;                               // [256]
    float* s_q = s_va + kHidden;                                                // [wave][unit tile 2][r 16][64 lanes]: the query, parked (32 registers)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rt = wave >> 2, ug = wave & 3;
    const int tile = blockIdx.x * RT + rt;
    const bool active = tile < n_tiles;        // an odd number of tiles: the last workgroup's second half only keeps the barriers
    const int n = lane & 31, hh = lane >> 5;

    for (int i = threadIdx.x; i < kClasses * 4 * kHidden; i += blockDim.x) s_fcw[i] = fcw[i];
    if (threadIdx.x < kHidden) s_va[threadIdx.x] = va[threadIdx.x];

    const uint4* wap = wa + (size_t)(2 * ug) * kKB12 * 2 * kFragU4 + lane;      // unit tiles 2 ug and 2 ug + 1 are kKB12 * 2 fragments apart
    const uint4* uap = ua + (size_t)(2 * ug) * kKB12 * 2 * kFragU4 + lane;
    constexpr size_t UT = (size_t)kKB12 * 2 * kFragU4;
    const uint4* otile = out2 + (size_t)(active ? tile : 0) * kSeqLen * kKB12 * 2 * kFragU4;   // [t][kb][hi|corr][64]
    const uint4* obase = out2 + (size_t)blockIdx.x * RT * kSeqLen * kKB12 * 2 * kFragU4;
    const bool second = blockIdx.x * RT + 1 < n_tiles;
    const u32x4_t orsrc = dma_rsrc(obase);

    // ---- q = Wa h_n for this wave's two unit tiles
    f32x16 qacc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) qacc[u][r] = 0.f;
    {
        uint4 qa[3][12];
        auto ldq = [&](uint4 (&d)[12], int kb) {
            const int tq = kb < kKBH ? kSeqLen - 1 : 0;
            const uint4* xp = otile + ((size_t)tq * kKB12 + kb) * 2 * kFragU4 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d[i] = wap[(kb * 2 + i) * kFragU4];               // unit tile 0: hi0, corr0, hi1, corr1
                d[4 + i] = wap[UT + (kb * 2 + i) * kFragU4];      // unit tile 1
                d[8 + i] = xp[i * kFragU4];
            }
        };
        ldq(qa[0], 0);
        ldq(qa[1], 2);
#pragma unroll
        for (int p = 0; p < kKB12 / 2; ++p) {
            if (p + 2 < kKB12 / 2) ldq(qa[(p + 2) % 3], 2 * (p + 2));
            asm volatile("" ::: "memory");
            const uint4(&d)[12] = qa[p % 3];
            qacc[0] = mfma16(d[0], d[8], qacc[0]);
            qacc[1] = mfma16(d[4], d[8], qacc[1]);
            qacc[0] = mfma16(d[2], d[10], qacc[0]);
            qacc[1] = mfma16(d[6], d[10], qacc[1]);
            qacc[0] = mfma_corr(d[1], d[3], d[9], d[11], qacc[0], sa_wa);
            qacc[1] = mfma_corr(d[5], d[7], d[9], d[11], qacc[1], sa_wa);
            asm volatile("" ::: "memory");
        }
    }
    // the query is needed again only when a timestep group's scores are taken: it waits in wave-private LDS meanwhile
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_q[((wave * 2 + u) * 16 + r) * 64 + lane] = qacc[u][r];
    int strand = 0;   // which half of fc1.weight this lane's row multiplies: strand 2 rows are the upper half of a slice
    {
        const int row = tile * 32 + n;
        for (int i = 0; i < slices.count; ++i)
            if (row >= slices.row_base[i] && row < slices.row_base[i] + 2 * slices.n_sites[i])
                strand = (row - slices.row_base[i]) >= slices.n_sites[i];
    }

    // stage chunk `c` of timestep group t0 into buffer `buf`: fragment f = (((rtl * CK + kbl) * TG + tt) * 2 + hl), three per wave
    auto stage = [&](int t0, int c, int buf) {
#pragma unroll
        for (int i = 0; i < CHUNK_FRAGS / kWaves; ++i) {
            const int f = wave + kWaves * i;
            const int hl = f & 1, tt = (f >> 1) % TG, kbl = ((f >> 1) / TG) % CK, rtl = (f >> 1) / (TG * CK);
            if (rtl == 0 || second) {
                // buffer-descriptor addressing: wave-uniform byte offset in an SGPR + lane * 16 in one VGPR (per-lane 64-bit source
                // pointers of the three fragments were hoisted out of the loops and spilled)
                const int soff = (int)(((((size_t)rtl * kSeqLen + (t0 + tt)) * kKB12 + (c * CK + kbl)) * 2 + hl) * kFragU4 * sizeof(uint4));
                dma16_buf(orsrc, lane * 16, __builtin_amdgcn_readfirstlane(soff),
                          __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_stage + (buf * CHUNK_FRAGS + f) * 1024)));
            }
        }
    };

    uint4 wu[2][CK][2];       // [unit tile][k-block][hi|corr]: Ua fragments of the next chunk, requested one chunk ahead
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int kbl = 0; kbl < CK; ++kbl) {
            wu[u][kbl][0] = uap[u * UT + (kbl * 2 + 0) * kFragU4];
            wu[u][kbl][1] = uap[u * UT + (kbl * 2 + 1) * kFragU4];
        }

    for (int tg = 0; tg < kSeqLen / TG; ++tg) {
        const int t0 = tg * TG;
        f32x16 kacc[TG][2];
#pragma unroll
        for (int tt = 0; tt < TG; ++tt)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) kacc[tt][u][r] = 0.f;
        float pf0 = 0.f, pf1 = 0.f;           // fc1 partials of timestep t0 + ug (ug < TG) of this wave's row tile, all k-blocks

        stage(t0, 0, 0);
#pragma unroll 1
        for (int c = 0; c < NCHUNK; ++c) {
            uint4 w[2][CK][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int kbl = 0; kbl < CK; ++kbl) { w[u][kbl][0] = wu[u][kbl][0]; w[u][kbl][1] = wu[u][kbl][1]; }
            wait_dma();        // this wave's part of chunk c (and the Ua fragments above) has arrived ...
            __syncthreads();   // ... and so has everybody else's: chunk c is in LDS; buffer (c+1)&1 is free
            if (c + 1 < NCHUNK) stage(t0, c + 1, (c + 1) & 1);
            {
                const int cn = c + 1 < NCHUNK ? c + 1 : 0;       // Ua is re-streamed for every timestep group
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int kbl = 0; kbl < CK; ++kbl) {
                        wu[u][kbl][0] = uap[u * UT + ((cn * CK + kbl) * 2 + 0) * kFragU4];
                        wu[u][kbl][1] = uap[u * UT + ((cn * CK + kbl) * 2 + 1) * kFragU4];
                    }
            }
            const char* sb0 = s_stage + ((c & 1) * CHUNK_FRAGS + rt * CK * TG * 2) * 1024;     // this wave's row tile
            const char* sb = sb0 + lane * 16;
            // ONE operand set: timestep tt + 1's fragments are read right behind the issue of timestep tt's six MFMAs (the matrix pipe has
            // latched its operands by then), their LDS latency runs under those MFMAs and the partner wave's; a second set spilled
            uint4 xo[4];
            auto rdop = [&](uint4 (&d)[4], int tt) {
                d[0] = *reinterpret_cast<const uint4*>(sb + ((0 * TG + tt) * 2 + 0) * 1024);
                d[1] = *reinterpret_cast<const uint4*>(sb + ((0 * TG + tt) * 2 + 1) * 1024);
                d[2] = *reinterpret_cast<const uint4*>(sb + ((1 * TG + tt) * 2 + 0) * 1024);
                d[3] = *reinterpret_cast<const uint4*>(sb + ((1 * TG + tt) * 2 + 1) * 1024);
            };
            rdop(xo, 0);
#pragma unroll
            for (int tt = 0; tt < TG; ++tt) {
                asm volatile("" ::: "memory");
                const uint4 x0h = xo[0], x0c = xo[1], x1h = xo[2], x1c = xo[3];
                kacc[tt][0] = mfma16(w[0][0][0], x0h, kacc[tt][0]);
                kacc[tt][1] = mfma16(w[1][0][0], x0h, kacc[tt][1]);
                kacc[tt][0] = mfma16(w[0][1][0], x1h, kacc[tt][0]);
                kacc[tt][1] = mfma16(w[1][1][0], x1h, kacc[tt][1]);
                kacc[tt][0] = mfma_corr(w[0][0][1], w[0][1][1], x0c, x1c, kacc[tt][0], sa_ua);
                kacc[tt][1] = mfma_corr(w[1][0][1], w[1][1][1], x0c, x1c, kacc[tt][1], sa_ua);
                asm volatile("" ::: "memory");
                if (tt + 1 < TG) rdop(xo, tt + 1);
                // fc1 partial of k-block tt (tt < CK) for timestep t0 + ug of this row tile, in the shadow of the MFMAs
                if (tt < CK && ug < TG) {
                    const int kbl = tt;
                    const int kb = c * CK + kbl;
                    const char* fr = sb0 + ((kbl * TG + ug) * 2) * 1024;
                    const half8 xh = as_half8(*reinterpret_cast<const uint4*>(fr + lane * 16));
                    const int la = *reinterpret_cast<const int*>(fr + 1024 + (n + 32) * 16 + 4 * hh);
                    const int lb = *reinterpret_cast<const int*>(fr + 1024 + (n + 32) * 16 + 8 + 4 * hh);
                    const float xl[8] = {__builtin_amdgcn_cvt_f32_fp8(la, 0), __builtin_amdgcn_cvt_f32_fp8(la, 1),
                                         __builtin_amdgcn_cvt_f32_fp8(la, 2), __builtin_amdgcn_cvt_f32_fp8(la, 3),
                                         __builtin_amdgcn_cvt_f32_fp8(lb, 0), __builtin_amdgcn_cvt_f32_fp8(lb, 1),
                                         __builtin_amdgcn_cvt_f32_fp8(lb, 2), __builtin_amdgcn_cvt_f32_fp8(lb, 3)};
                    const float4* f0 = reinterpret_cast<const float4*>(s_fcw + 0 * 4 * kHidden + strand * 2 * kHidden + kb * 16 + hh * 8);
                    const float4* f1 = reinterpret_cast<const float4*>(s_fcw + 1 * 4 * kHidden + strand * 2 * kHidden + kb * 16 + hh * 8);
                    const float4 a0 = f0[0], a1 = f0[1], b0 = f1[0], b1 = f1[1];
                    const float fa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    const float fb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float xv = (float)xh[j] + xl[j] * (1.0f / kCorrActLo);
                        pf0 += fa[j] * xv;
                        pf1 += fb[j] * xv;
                    }
                }
            }
        }
        float ev[TG] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float qv = s_q[((wave * 2 + u) * 16 + r) * 64 + lane];
                const float vv = s_va[((2 * ug + u) * 2 + hh) * 16 + r];
#pragma unroll
                for (int tt = 0; tt < TG; ++tt) ev[tt] += vv * tanh_f(qv + kacc[tt][u][r]);
            }
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) {
            float e = ev[tt];
            e += __shfl_xor(e, 32);
            if (hh == 0) s_epart[((ug * TG + tt) * RT + rt) * 32 + n] = e;
        }
        if (ug < TG) {
            pf0 += __shfl_xor(pf0, 32);
            pf1 += __shfl_xor(pf1, 32);
            if (hh == 0) {
                s_pfc[((rt * kSeqLen + t0 + ug) * 32 + n) * 2 + 0] = pf0;
                s_pfc[((rt * kSeqLen + t0 + ug) * 32 + n) * 2 + 1] = pf1;
            }
        }
        __syncthreads();   // all waves are done with both staging buffers before the next group restages buffer 0
        if (threadIdx.x < TG * RT * 32) {       // the group's scores: the four unit groups' partials in a fixed order (deterministic)
            const int tt = threadIdx.x / (RT * 32), rtl = (threadIdx.x >> 5) % RT, rl = threadIdx.x & 31;
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) v += s_epart[((g * TG + tt) * RT + rtl) * 32 + rl];
            s_e[(rtl * kSeqLen + t0 + tt) * 32 + rl] = v;
        }                                       // (s_epart is next written at the end of the next group, a chunk loop of barriers away)
    }
    __syncthreads();

    // ---- softmax over t and the strand-half of the logits (fixed summation order: deterministic)
    if (threadIdx.x < RT * 32) {
        const int rtl = threadIdx.x >> 5, rl = threadIdx.x & 31;
        const int tl = blockIdx.x * RT + rtl;
        if (tl < n_tiles) {
            const int row = tl * 32 + rl;
            float e[kSeqLen];
            float m = -3.0e38f;
#pragma unroll
            for (int t = 0; t < kSeqLen; ++t) {
                const float v = s_e[(rtl * kSeqLen + t) * 32 + rl];
                e[t] = v;
                m = fmaxf(m, v);
            }
            float den = 0.f;
#pragma unroll
            for (int t = 0; t < kSeqLen; ++t) { e[t] = __expf(e[t] - m); den += e[t]; }
            const float inv = 1.0f / den;
            float l0 = 0.f, l1 = 0.f;
#pragma unroll
            for (int t = 0; t < kSeqLen; ++t) {
                const float a = e[t] * inv;
                l0 += a * s_pfc[((rtl * kSeqLen + t) * 32 + rl) * 2 + 0];
                l1 += a * s_pfc[((rtl * kSeqLen + t) * 32 + rl) * 2 + 1];
            }
            part[(size_t)row * 2 + 0] = l0;
            part[(size_t)row * 2 + 1] = l1;
        }
    }
}

