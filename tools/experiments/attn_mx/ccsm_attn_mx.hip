// libccsm attention pool + FC on the split-mx activation format.  Included by ccsm_api.hip after ccsm_gru_mx.hip.
#include <hip/hip_runtime.h>

namespace ccsm {

// ---------------------------------------------------------------------------------------------------------
// Attention pool + FC partials in split-mx arithmetic: attn_fc_kernel (ccsm_kernels.hip) on the GRU layers' own output format, the
// activation blobs of ccsm_gru_mx.hip.  One staged chunk (2 k-blocks) is exactly one pair: per timestep two main MFMAs and one K = 64
// correction MFMA (fp6 x fp6) instead of six fp16 MFMAs.  The fc1 partial dot products need the activations themselves: hi from the
// main fragment, the residual from the pair's blob (lane (n, 1): x_lo * 2^14 as fp6 in kMxPerm order, decoded 32 values at a time).
//   wa / ua : [wave][kb 32][hi|corr][64] uint4; the corr fragments of a pair hold the fp6 weight blob: bytes 0-15 of a lane in the first,
//             bytes 16-23 + the lane's scale dword (byte 0: E8M0 of the block) in the second
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void attn_fc_mx_kernel(const uint4* __restrict__ out2, const uint4* __restrict__ wa,
                                                             const uint4* __restrict__ ua, const float* __restrict__ va,
                                                             const float* __restrict__ fcw, float* __restrict__ part,
                                                             SliceTable slices) {
    constexpr int TG = 7;                      // timesteps per Ua pass (21 = 3 * 7)
    constexpr int CK = 2;                      // k-blocks per staged chunk = one pair
    constexpr int NCHUNK = kKB12 / CK;
    constexpr int CHUNK_FRAGS = CK * TG * 2;   // 28 fragments of 1 KiB
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_stage = smem;                                                     // [2][CK][TG][hi|corr] fragments
    float* s_epart = reinterpret_cast<float*>(smem + 2 * CHUNK_FRAGS * 1024);  // [wave][t][32]
    float* s_pfc = s_epart + kWaves * kSeqLen * 32;                            // [wave][t][32][2]
    float* s_fcw = s_pfc + kWaves * kSeqLen * 32 * 2;                          // [2][1024]
    float* s_va = s_fcw + kClasses * 4 * kHidden;                              // [256]: in LDS, the 16 registers go to the operand pipeline

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x;
    const int n = lane & 31, hh = lane >> 5;
    const int act_sb = hh ? kMxScaleLo : kMxScaleHi;      // E8M0 scale of this lane's half of an activation blob

    for (int i = threadIdx.x; i < kClasses * 4 * kHidden; i += blockDim.x) s_fcw[i] = fcw[i];
    if (threadIdx.x < kHidden) s_va[threadIdx.x] = va[threadIdx.x];

    const uint4* wap = wa + (size_t)wave * kKB12 * 2 * kFragU4 + lane;
    const uint4* uap = ua + (size_t)wave * kKB12 * 2 * kFragU4 + lane;
    const uint4* otile = out2 + (size_t)tile * kSeqLen * kKB12 * 2 * kFragU4;   // [t][kb][hi|corr][64]

    // ---- q = Wa h_n, h_n = [fwd final state = out[t=L-1][0:256] | bwd final state = out[t=0][256:512]] (models.py:135-137)
    f32x16 qacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) qacc[r] = 0.f;
    {   // operands of pairs p+1 .. p+3 are in flight while pair p multiplies (four register sets of 8 fragments: the activation
        // fragments come from HBM, one pair of look-ahead left the loop at one round trip per pair)
        uint4 qa[4][8];
        auto ldq = [&](uint4 (&d)[8], int kb) {
            const int tq = kb < kKBH ? kSeqLen - 1 : 0;
            const uint4* xp = otile + ((size_t)tq * kKB12 + kb) * 2 * kFragU4 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d[i] = wap[(kb * 2 + i) * kFragU4];           // hi0, corr0, hi1, corr1
                d[4 + i] = xp[i * kFragU4];
            }
        };
        ldq(qa[0], 0);
        ldq(qa[1], 2);
        ldq(qa[2], 4);
#pragma unroll
        for (int p = 0; p < kKB12 / 2; ++p) {
            if (p + 3 < kKB12 / 2) ldq(qa[(p + 3) & 3], 2 * (p + 3));
            asm volatile("" ::: "memory");
            const uint4(&d)[8] = qa[p & 3];
            qacc = mfma16(d[0], d[4], qacc);
            qacc = mfma16(d[2], d[6], qacc);
            qacc = mfma_corr_mx6<0>(d[1], make_uint2(d[3].x, d[3].y), d[3].z, d[5], make_uint2(d[7].x, d[7].y), qacc, act_sb);
            asm volatile("" ::: "memory");
        }
    }
    int strand = 0;   // which half of fc1.weight this lane's row multiplies: strand 2 rows are the upper half of a slice
    {
        const int row = tile * 32 + n;
        for (int i = 0; i < slices.count; ++i)
            if (row >= slices.row_base[i] && row < slices.row_base[i] + 2 * slices.n_sites[i])
                strand = (row - slices.row_base[i]) >= slices.n_sites[i];
    }

    // stage chunk `c` of timestep group t0 into buffer `buf`: fragment f = (kbl * TG + tt) * 2 + hl
    auto stage = [&](int t0, int c, int buf) {
#pragma unroll
        for (int i = 0; i < (CHUNK_FRAGS + kWaves - 1) / kWaves; ++i) {
            const int f = wave + kWaves * i;
            if (f < CHUNK_FRAGS) {
                const int hl = f & 1, tt = (f >> 1) % TG, kbl = (f >> 1) / TG;
                const uint4* src = otile + (((size_t)(t0 + tt) * kKB12 + (c * CK + kbl)) * 2 + hl) * kFragU4 + lane;
                dma16(src, __builtin_amdgcn_readfirstlane(
                               (unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_stage + (buf * CHUNK_FRAGS + f) * 1024)));
            }
        }
    };

    uint4 wu[CK][2];       // Ua fragments of the next chunk, requested one chunk ahead
#pragma unroll
    for (int kbl = 0; kbl < CK; ++kbl) { wu[kbl][0] = uap[(kbl * 2 + 0) * kFragU4]; wu[kbl][1] = uap[(kbl * 2 + 1) * kFragU4]; }

    for (int tg = 0; tg < kSeqLen / TG; ++tg) {
        const int t0 = tg * TG;
        f32x16 kacc[TG];
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) kacc[tt] = qacc;     // accumulate K_t on top of q
        // fc1 partials: wave w takes timestep t0 + w of the group for ALL k-blocks (w < 7): the work of every chunk is spread
        // over seven waves instead of falling on the two that own its k-blocks (they made the other six wait at the barrier),
        // and each (row, t) sum is complete in one wave's registers, in a fixed order
        float pf0 = 0.f, pf1 = 0.f;

        stage(t0, 0, 0);
#pragma unroll 1
        for (int c = 0; c < NCHUNK; ++c) {
            uint4 w[CK][2];
#pragma unroll
            for (int kbl = 0; kbl < CK; ++kbl) { w[kbl][0] = wu[kbl][0]; w[kbl][1] = wu[kbl][1]; }
            wait_dma();        // this wave's part of chunk c (and the Ua fragments above) has arrived ...
            __syncthreads();   // ... and so has everybody else's: chunk c is in LDS; buffer (c+1)&1 is free
            if (c + 1 < NCHUNK) stage(t0, c + 1, (c + 1) & 1);
            {
                const int cn = c + 1 < NCHUNK ? c + 1 : 0;       // Ua is re-streamed for every timestep group
#pragma unroll
                for (int kbl = 0; kbl < CK; ++kbl) {
                    wu[kbl][0] = uap[((cn * CK + kbl) * 2 + 0) * kFragU4];
                    wu[kbl][1] = uap[((cn * CK + kbl) * 2 + 1) * kFragU4];
                }
            }
            const char* sb0 = s_stage + (c & 1) * CHUNK_FRAGS * 1024;
            const char* sb = sb0 + lane * 16;
            // operands of timestep tt + 1 are read from LDS before the three MFMAs of timestep tt are issued (two register sets,
            // pinned with compiler fences: left alone the compiler reads each operand right in front of its MFMA and every MFMA
            // waits out an LDS round trip — 4.2 k cycles per chunk for 1.85 k of matrix work)
            uint4 xo[2][4];
            auto rdop = [&](uint4 (&d)[4], int tt) {
                d[0] = *reinterpret_cast<const uint4*>(sb + ((0 * TG + tt) * 2 + 0) * 1024);
                d[1] = *reinterpret_cast<const uint4*>(sb + ((0 * TG + tt) * 2 + 1) * 1024);
                d[2] = *reinterpret_cast<const uint4*>(sb + ((1 * TG + tt) * 2 + 0) * 1024);
                d[3] = *reinterpret_cast<const uint4*>(sb + ((1 * TG + tt) * 2 + 1) * 1024);
            };
            rdop(xo[0], 0);
#pragma unroll
            for (int tt = 0; tt < TG; ++tt) {
                asm volatile("" ::: "memory");
                if (tt + 1 < TG) rdop(xo[(tt + 1) & 1], tt + 1);
                asm volatile("" ::: "memory");
                const uint4 x0h = xo[tt & 1][0], x0c = xo[tt & 1][1], x1h = xo[tt & 1][2], x1c = xo[tt & 1][3];
                kacc[tt] = mfma16(w[0][0], x0h, kacc[tt]);
                kacc[tt] = mfma16(w[1][0], x1h, kacc[tt]);
                kacc[tt] = mfma_corr_mx6<0>(w[0][1], make_uint2(w[1][1].x, w[1][1].y), w[1][1].z, x0c, make_uint2(x1c.x, x1c.y), kacc[tt], act_sb);
                // fc1 partials of the pair for timestep t0 + wave, issued behind the second and third timestep's MFMAs: vector ALU and LDS work in the
                // shadow of the matrix pipe.  hi = the two hi fragments; residuals = the blob of row n's lo lane (n + 32): 16 bytes in the
                // pair's first corr fragment, 8 in the second, decoded 32 values at a time to fp16.  Position j' of the blob is
                // k = kMxPerm[j'], so k = 16 kbl + 8 hh + j sits at j' = 8 kbl + 4 hh + j (j < 4) and 16 + 8 kbl + 4 hh + (j - 4): as
                // fp16 pairs, registers 4 kbl + 2 hh + {0, 1} and 8 + 4 kbl + 2 hh + {0, 1}.
                if ((tt == 1 || tt == 2) && wave < TG) {
                    typedef _Float16 half32v __attribute__((ext_vector_type(32)));
                    typedef uint32_t u32x16v __attribute__((ext_vector_type(16)));
                    const char* cr = sb0 + ((0 * TG + wave) * 2 + 1) * 1024 + (n + 32) * 16;
                    const uint4 q0 = *reinterpret_cast<const uint4*>(cr);
                    const uint2 q1 = *reinterpret_cast<const uint2*>(cr + TG * 2 * 1024);
                    const i32x6 blob = {(int)q0.x, (int)q0.y, (int)q0.z, (int)q0.w, (int)q1.x, (int)q1.y};
                    const u32x16v lo = __builtin_bit_cast(u32x16v, __builtin_amdgcn_cvt_scalef32_pk32_f16_fp6(blob, 1.0f));
                    {
                        const int kbl = tt - 1;                      // one k-block per timestep slot: both at once spill an accumulator tile
                        const int kb = c * CK + kbl;
                        const half8 xh = as_half8(*reinterpret_cast<const uint4*>(sb0 + ((kbl * TG + wave) * 2) * 1024 + lane * 16));
                        const uint32_t p[4] = {hh ? lo[4 * kbl + 2] : lo[4 * kbl + 0], hh ? lo[4 * kbl + 3] : lo[4 * kbl + 1],
                                               hh ? lo[8 + 4 * kbl + 2] : lo[8 + 4 * kbl + 0], hh ? lo[8 + 4 * kbl + 3] : lo[8 + 4 * kbl + 1]};
                        const float4* f0 = reinterpret_cast<const float4*>(s_fcw + 0 * 4 * kHidden + strand * 2 * kHidden + kb * 16 + hh * 8);
                        const float4* f1 = reinterpret_cast<const float4*>(s_fcw + 1 * 4 * kHidden + strand * 2 * kHidden + kb * 16 + hh * 8);
                        const float4 a0 = f0[0], a1 = f0[1], b0 = f1[0], b1 = f1[1];
                        const float fa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                        const float fb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            typedef _Float16 half2v __attribute__((ext_vector_type(2)));
                            const half2v l2 = __builtin_bit_cast(half2v, p[j >> 1]);
                            const float xv = (float)xh[j] + (float)l2[j & 1] * (1.0f / 16384.0f);
                            pf0 += fa[j] * xv;
                            pf1 += fb[j] * xv;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) {
            float e = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) e += s_va[(wave * 2 + hh) * 16 + r] * tanh_f(kacc[tt][r]);
            e += __shfl_xor(e, 32);
            if (hh == 0) s_epart[(wave * kSeqLen + t0 + tt) * 32 + n] = e;
        }
        if (wave < TG) {
            pf0 += __shfl_xor(pf0, 32);
            pf1 += __shfl_xor(pf1, 32);
            if (hh == 0) {
                s_pfc[((t0 + wave) * 32 + n) * 2 + 0] = pf0;
                s_pfc[((t0 + wave) * 32 + n) * 2 + 1] = pf1;
            }
        }
        __syncthreads();   // all waves are done with both staging buffers before the next group restages buffer 0
    }

    // ---- softmax over t and the strand-half of the logits (fixed summation order: deterministic)
    if (threadIdx.x < 32) {
        const int rl = threadIdx.x;
        const int row = tile * 32 + rl;
        float e[kSeqLen];
        float m = -3.0e38f;
#pragma unroll
        for (int t = 0; t < kSeqLen; ++t) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) v += s_epart[(w * kSeqLen + t) * 32 + rl];
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
            l0 += a * s_pfc[(t * 32 + rl) * 2 + 0];
            l1 += a * s_pfc[(t * 32 + rl) * 2 + 1];
        }
        part[(size_t)row * 2 + 0] = l0;
        part[(size_t)row * 2 + 1] = l1;
    }
}

}  // namespace ccsm
