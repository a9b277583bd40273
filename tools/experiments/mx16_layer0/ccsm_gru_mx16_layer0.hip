// libccsm GRU layer 0 in split-mx arithmetic on the 16-wide matrix instructions: v_mfma_f32_16x16x32_f16 (main product) +
// v_mfma_scale_f32_16x16x128_f8f6f4 (both correction terms of TWO pairs of k-blocks).  Included by ccsm_api.hip after ccsm_gru_f3s.hip.
//
// Why: at the package power cap the 16-wide instruction mix sustains 15 % more than the 32-wide one (tools/ubench/mfma_power_mix_shapes.hip),
// and the recurrent phase of the split-mx kernels in isolation runs 9.7 % faster on it (tools/ubench/phase_h_shapes.hip, profiles/r05_b2).
// Layer 0 is that phase plus a tail (its input part is one k-block), so it is the first kernel of the family to move; its input, its output
// (what gru_layer12_mx_kernel reads) and its LDS layout are gru_layer0_mx_kernel's, byte for byte.
//
// Arithmetic (= ccsm_gru_mx.hip's): W x = W_hi x_hi (fp16, one 16x16x32 per pair) + [W_lo x_hi | W_hi x_lo] (fp4 x fp6 blobs under E8M0 scales).
// The scaled instruction contracts K = 128 = four blocks of 32: lane (m, q) of A holds 32 values of unit row m, q & 1 = term (0: W_lo 2^11,
// 1: W_hi), q >> 1 = pair of the DOUBLE pair; lane (n', q) of B reads the existing activation blob fragment of pair q >> 1 (lane (n, g) of a
// 32-row blob fragment = 32 values of row n: g = 0 x_hi, g = 1 x_lo, kMxPerm order) at lane position 16 h + n' + 32 (q & 1): bytes 0-15 from
// the corr fragment of the pair's first k-block, 16-23 from the second's.  Unit tiles as in ccsm_gru_f3s.hip: tile T, row 4 q + i <-> hidden
// unit 32 wave + 8 q + 4 T + i, so the step tail's hi fragments need no lane exchange; the blobs (32 units of ONE row per lane) do: the four
// lane groups exchange 4-dword chunks (eight v_permlane32_swap + eight v_permlane16_swap per row tile; the 32-wide tail: eight swaps).
//   xin : [tile][t][hi|lo][64] uint4                  out : [tile][t][32 kb][hi | corr][64] uint4 (blobs in the corr fragments)
//   wst : per (direction, wave): x-part A1 / A2 fragments of (T, g in r, z) (ccsm_gru_f3s.hip's layer 0)                          8 KiB
//         per double pair D: hi of pair 2D (T, g) at (3 T + g) KiB | hi of pair 2D + 1 at 6 + ... | fp4 blobs (T, g) at 12 + ... |
//                            scale dwords (lane * 4; byte g = gate g) of T = 0 at 18 KiB, of T = 1 at 18 KiB + 256                18.5 KiB x 4
//         x-part of n: A1 / A2 of (T)                                                                                            4 KiB
//   LDS : gru_layer0_mx_kernel's (h fragments [kb][bt][hi | corr] | x double buffer | fp8 residuals | biases); the fp8 residuals (x 2^16) of
//         a lane's own 8 units: row half 0 in bytes 8-15 of its slot in the corr fragment of the wave's second k-block, half 1 at LO_OFF
#include <hip/hip_runtime.h>

namespace ccsm {

typedef unsigned u32x6_t __attribute__((ext_vector_type(6)));

constexpr int kMx16DW = 18 * 1024 + 512;                                     // weight bytes of one double pair
constexpr int kMx16OffB = 8 * 1024, kMx16OffC = kMx16OffB + (kKBH / 4) * kMx16DW, kMx16WBytes = kMx16OffC + 4 * 1024;

#define CCSM_FENCE asm volatile("" ::: "memory")

// scaled product with the accumulator tied; BYTE = which byte of the A scale dword
template <int BYTE>
__device__ __forceinline__ f32x4 mfma_corr16(uint4 a, uint32_t sa, uint4 b0, uint2 b1, uint32_t sb, f32x4 c) {
    const u32x4_t av = __builtin_bit_cast(u32x4_t, a);
    const u32x6_t bv = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y};
    // (s_nop 1: hipcc assembles the six-register B operand with v_mov right in front of the statement - tools/isa_hazard_scan.py finds 24 of
    //  them per step without it - and pads no hazard whose consumer is inside an asm string)
    if constexpr (BYTE == 0) asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:2" : "+v"(c) : "v"(av), "v"(bv), "v"(sa), "v"(sb));
    else if constexpr (BYTE == 1) asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[1,0,0] op_sel_hi:[0,0,0] cbsz:4 blgp:2" : "+v"(c) : "v"(av), "v"(bv), "v"(sa), "v"(sb));
    else asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[1,0,0] cbsz:4 blgp:2" : "+v"(c) : "v"(av), "v"(bv), "v"(sa), "v"(sb));
    return c;
}

__device__ __forceinline__ void swap16(uint32_t& x, uint32_t& y) {            // odd 16-lane rows of x <-> even rows of y
    const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}

// One row tile of new values in the 16x16 C layout: v[h][j] = unit 8 q + j of row 16 h + n' (lane (n', q)).  Produces
//   hi[h]      : this lane's 16 bytes of the hi fragments (k-block q >> 1 of the wave's pair, lane position n' + 32 (q & 1) + 16 h)
//   lo8[h]     : fp8 residuals (x 2^16) of the lane's own 8 values (private copy for the blend)
//   c0, c1     : THIS PHYSICAL LANE's 24 bytes of the row tile's activation blob fragment (lane L = n + 32 g: x_hi (g = 0) / x_lo (g = 1) of the
//                wave's 32 units of row n, kMxPerm order), gathered from the four lanes that hold a row
// CLAMP / scale: pack_pair_mx's (initial states: clamped, divided by kMxH0Div; GRU outputs: 0.25)
template <bool CLAMP>
__device__ __forceinline__ void mx16_pack(const float (&v)[2][8], float scale, uint4 (&hi)[2], uint2 (&lo8)[2], uint4& c0, uint2& c1) {
    typedef _Float16 half2p __attribute__((ext_vector_type(2)));
    uint32_t ch[4][4];                                              // chunks by target lane group q_t = h + 2 g: [q_t][dword]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t hp[4];
        float lf[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const half2p hh = {(_Float16)v[h][2 * j], (_Float16)v[h][2 * j + 1]};
            hp[j] = __builtin_bit_cast(uint32_t, hh);
            lf[2 * j] = v[h][2 * j] - (float)hh[0];
            lf[2 * j + 1] = v[h][2 * j + 1] - (float)hh[1];
            float l0 = lf[2 * j] * 4096.0f, l1 = lf[2 * j + 1] * 4096.0f;
            if constexpr (CLAMP) {
                const float top = 7.5f * scale;
                const half2p hc = {(_Float16)fminf(fmaxf(v[h][2 * j], -top), top), (_Float16)fminf(fmaxf(v[h][2 * j + 1], -top), top)};
                ch[h][j] = __builtin_bit_cast(uint32_t, hc);
                l0 = fminf(fmaxf(l0, -top), top);
                l1 = fminf(fmaxf(l1, -top), top);
            } else {
                ch[h][j] = hp[j];
            }
            ch[2 + h][j] = pack2((_Float16)l0, (_Float16)l1);
        }
        hi[h] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
        uint32_t r[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float c[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = __builtin_amdgcn_fmed3f(lf[4 * q + e], -kF8Clamp / kMxLoScale, kF8Clamp / kMxLoScale);
            short2v t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_bit_cast(short2v, hp[2 * q]), c[0], c[1], 1.0f / kMxLoScale, false);
            t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, c[2], c[3], 1.0f / kMxLoScale, true);
            r[q] = __builtin_bit_cast(uint32_t, t);
        }
        lo8[h] = make_uint2(r[0], r[1]);
    }
    // 4 x 4 exchange of the chunks between the lane groups: afterwards group q holds chunk q of all four source groups; the register a value
    // ends up in names its SOURCE group - stage 1 (bit 1: lanes 0-31 <-> 32-63) leaves sources 0, 1 in ch[0 / 1] and 2, 3 in ch[2 / 3], stage 2
    // (bit 0: odd <-> even rows of 16) sources 0, 2 in ch[0 / 2] and 1, 3 in ch[1 / 3]
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        swap32(ch[0][d], ch[2][d]);
        swap32(ch[1][d], ch[3][d]);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        swap16(ch[0][d], ch[1][d]);
        swap16(ch[2][d], ch[3][d]);
    }
    // natural dword 4 q_s + d = units 8 q_s + 2 d, + 1; blob dword p takes natural dword 4 ((p & 7) >> 1) + (p & 1) + 2 (p >> 3)  (kMxPerm)
    uint32_t p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int nd = 4 * ((i & 7) >> 1) + (i & 1) + 2 * (i >> 3);
        p[i] = ch[nd >> 2][nd & 3];
    }
    blob_of(p, scale, c0, c1);
}

template <int NB_ = kMxNB>
__global__ __launch_bounds__(512, 2) void gru_layer0_mx16_kernel(const uint4* __restrict__ xin, uint4* __restrict__ out, const uint4* __restrict__ wst,
                                                                  const float* __restrict__ bias, const float* __restrict__ h0, int rows_p) {
    constexpr int NB = NB_;
    constexpr int X_OFF = mx0_xoff(NB), LO_OFF = mx0_looff(NB), BIAS_OFF = mx0_biasoff(NB);
    constexpr int DW = kMx16DW, OFF_B = kMx16OffB, OFF_C = kMx16OffC, ND = kKBH / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dir = blockIdx.x & 1;
    const int tile0 = (blockIdx.x >> 1) * NB;
    const int lane16 = lane * 16;

    if (threadIdx.x < kWaves * 4 * 32 / 4)
        reinterpret_cast<float4*>(smem + BIAS_OFF)[threadIdx.x] = reinterpret_cast<const float4*>(bias + (size_t)dir * kWaves * 4 * 32)[threadIdx.x];

    // per-lane offsets (bytes): lane position n' + 32 (q & 1) inside a fragment; + the k-block (q >> 1) of a pair in a [kb][bt][hl] array;
    // + the PAIR (q >> 1) of a double pair
    const int lpos = (((lane >> 4) & 1) * 32 + (lane & 15)) << 4;
    const int kbo = (lane & 32) << 6;
    const int lxs = lpos + (NB == 1 ? kbo : NB == 2 ? (kbo << 1) : kbo + (kbo << 1));
    const int lbs = lpos + (lane >> 5) * (2 * NB * 2048);
    const int own = wave * (2 * NB * 2 * 1024);                      // mx_hfrag(2 wave, 0, 0)

    // ---- h0 -> LDS: hi fragments, blobs (coarse scale) and residuals of this wave's own units, every row tile
    {
        const float* h0d = h0 + (size_t)dir * rows_p * kHidden;
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            float v[2][8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float* src = h0d + ((size_t)(tile0 + bt) * 32 + 16 * h + (lane & 15)) * kHidden + 32 * wave + 8 * (lane >> 4);
                const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
                v[h][0] = a.x; v[h][1] = a.y; v[h][2] = a.z; v[h][3] = a.w; v[h][4] = b.x; v[h][5] = b.y; v[h][6] = b.z; v[h][7] = b.w;
            }
            uint4 hi[2], c0;
            uint2 lo8[2], c1;
            mx16_pack<true>(v, kMxH0Div, hi, lo8, c0, c1);
            *reinterpret_cast<uint4*>(smem + own + lxs + bt * 2048) = hi[0];
            *reinterpret_cast<uint4*>(smem + own + lxs + bt * 2048 + 256) = hi[1];
            *reinterpret_cast<uint4*>(smem + own + ((bt * 2 + 1) << 10) + lane16) = c0;
            *reinterpret_cast<uint4*>(smem + own + (((NB + bt) * 2 + 1) << 10) + lane16) = make_uint4(c1.x, c1.y, lo8[0].x, lo8[0].y);
            *reinterpret_cast<uint2*>(smem + LO_OFF + ((wave * NB + bt) * 64 + lane) * 8) = lo8[1];
        }
    }

    const u32x4_t xrs = dma_rsrc(xin + (size_t)tile0 * kSeqLen * 2 * kFragU4);
    const unsigned sx_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)X_OFF;    // (cast first: see gru_layer0_mx_kernel)
    auto stage_load = [&](int t, int buf) {
        const int f = wave < 2 * NB ? wave : 2 * NB - 1;
        const int hl = f & 1, bt = f >> 1;
        const int soff = (((bt * kSeqLen + t) * 2 + hl) << 10);
        dma16_buf(xrs, lane16, __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane((int)(sx_base + ((buf * 2 * NB + f) << 10))));
    };
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const char*>(wst) + (size_t)(dir * kWaves + wave) * kMx16WBytes);
    const int bias_off = BIAS_OFF + wave * 4 * 32 * 4;
    auto w_at = [&](int off) -> uint4 { return buf_load(wrs, lane16, off); };

    uint4 wxa[2][2][2];                                             // phase A: [unit tile][gate r, z][A1, A2]
    uint4 wxc[2][2];                                                // phase C: [unit tile][A1, A2]
    uint4 w[2][6];                                                  // phase B: two slots of six fragments in use order: hi of pair 2D | hi of pair 2D + 1 | blobs of D
    uint32_t wsc[2] = {0, 0};                                       // scale dwords of the blobs in flight (T = 0, 1)
    auto ld_set = [&](auto UC) {                                    // use U of a step (12): D = U / 3, kind = U % 3, slot = U & 1
        constexpr int U = decltype(UC)::value;
        constexpr int D = U / 3, KIND = U % 3, SL = U & 1;
#pragma unroll
        for (int i = 0; i < 6; ++i) w[SL][i] = w_at(OFF_B + D * DW + ((KIND * 6 + i) << 10));
        if constexpr (KIND == 2) {
            wsc[0] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, lane * 4, OFF_B + D * DW + (18 << 10), 0);
            wsc[1] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, lane * 4, OFF_B + D * DW + (18 << 10) + 256, 0);
        }
    };
    auto ld_first = [&]() {                                         // everything a step needs before its third use of phase B: 8 + 12 requests
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int g = 0; g < 2; ++g) { wxa[T][g][0] = w_at((4 * T + 2 * g) << 10); wxa[T][g][1] = w_at((4 * T + 2 * g + 1) << 10); }
        ld_set(std::integral_constant<int, 0>{});
        ld_set(std::integral_constant<int, 1>{});
    };
    stage_load(dir ? kSeqLen - 1 : 0, 0);
    ld_first();
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");               // the first transfer (older than the 20 weight requests)

    for (int s = 0; s < kSeqLen; ++s) {
        const int t = dir ? (kSeqLen - 1 - s) : s;
        const int tn = s + 1 < kSeqLen ? (dir ? t - 1 : t + 1) : t;
        f32x4 acc[3][2][NB][2];                                     // [gate R, Z, N][unit tile][row tile][16-row half]
        auto opq = [&](int v) -> int { asm volatile("" : "+v"(v)); return v; };     // opaque copies of the per-lane offsets
        auto bias_set = [&](int set, f32x4 (&b)[2]) {               // b[T][i] = bias of unit 8 q + 4 T + i
            const char* bp = smem + (bias_off + set * 128 + ((opq(lane) >> 4) << 5));
            const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 16);
            b[0] = f32x4{b0.x, b0.y, b0.z, b0.w};
            b[1] = f32x4{b1.x, b1.y, b1.z, b1.w};
        };
        {
            f32x4 b0[2], b1[2];
            bias_set(0, b0);
            bias_set(1, b1);
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int h = 0; h < 2; ++h) { acc[0][T][bt][h] = b0[T]; acc[1][T][bt][h] = b1[T]; }
        }
        // the transfer of this step's x (issued one step ago) is older than the 20 weight requests and 4 NB output stores of the tail
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(20 + 4 * NB) : "memory");
        __syncthreads();                                            // x_t in LDS; everybody's h_{t-1} fragments and blobs written
        stage_load(tn, (s + 1) & 1);
        uint4 x0[2][NB];
        auto rd_x0 = [&]() {                                        // B operand [x_hi | x_lo]: fragment hl = q >> 1, lane position n' + 32 (q & 1) + 16 h
            const int lx0 = opq(lpos) + ((opq(lane) & 32) << 5);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
                    x0[h][bt] = *reinterpret_cast<const uint4*>(smem + X_OFF + ((((s & 1) * NB + bt) * 2) << 10) + h * 256 + lx0);
        };
        // ---------------- phase A: R, Z += W_i{r,z} x_t (three fp16 passes in two instructions: ccsm_gru_f3s.hip) -------------------
        rd_x0();
        CCSM_FENCE;
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[g][T][bt][h] = mfma32k_first(wxa[T][g][0], x0[h][bt], acc[g][T][bt][h]);
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[g][T][bt][h] = mfma32k(wxa[T][g][1], x0[h][bt], acc[g][T][bt][h]);
        CCSM_FENCE;
        // ---------------- phase B: R, Z, N += W_h{r,z,n} h_{t-1}  (N starts at b_hn): twelve uses of six fragments per step ---------
        {
            f32x4 b3[2];
            bias_set(3, b3);
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int h = 0; h < 2; ++h) acc[2][T][bt][h] = b3[T];
        }
        // E8M0 scale of the activation blobs: x_hi * 4 | x_lo * 2^14; the initial states were packed kMxH0Div coarser (first step)
        const uint32_t sbv = (uint32_t)(((opq(lane) & 16) ? kMxScaleLo : kMxScaleHi) + (s == 0 ? kMxScaleHi0 - kMxScaleHi : 0));
        static_for<0, 12>([&](auto UC) {
            constexpr int U = decltype(UC)::value;
            constexpr int D = U / 3, KIND = U % 3, SL = U & 1;
            if constexpr (KIND < 2) {                               // main products of pair 2 D + KIND, row halves in turn
                constexpr int P = 2 * D + KIND;
                uint4 xh[2][NB];                                    // both row halves requested up front: the second is in flight under the first's 18 MFMAs
                {
                    const int a = opq(lxs);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int bt = 0; bt < NB; ++bt) xh[h][bt] = *reinterpret_cast<const uint4*>(smem + a + ((((2 * P) * NB + bt) * 2) << 10) + h * 256);
                }
                CCSM_FENCE;
                static_for<0, 2>([&](auto HC) {
                    constexpr int H = decltype(HC)::value;
#pragma unroll
                    for (int T = 0; T < 2; ++T)
#pragma unroll
                        for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                            for (int g = 0; g < 3; ++g)
                                acc[g][T][bt][H] = (U == 0 && g == 2) ? mfma32k_first(w[SL][3 * T + g], xh[H][bt], acc[g][T][bt][H]) : mfma32k(w[SL][3 * T + g], xh[H][bt], acc[g][T][bt][H]);
                    CCSM_FENCE;
                });
            } else {                                                // correction products of the double pair D
                uint4 xc0[2][NB];
                uint2 xc1[2][NB];
                {
                    const int a = opq(lbs);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int bt = 0; bt < NB; ++bt) {
                            xc0[h][bt] = *reinterpret_cast<const uint4*>(smem + a + ((((4 * D) * NB + bt) * 2 + 1) << 10) + h * 256);
                            xc1[h][bt] = *reinterpret_cast<const uint2*>(smem + a + ((((4 * D + 1) * NB + bt) * 2 + 1) << 10) + h * 256);
                        }
                }
                CCSM_FENCE;
                static_for<0, 2>([&](auto HC) {
                    constexpr int H = decltype(HC)::value;
#pragma unroll
                    for (int T = 0; T < 2; ++T)
#pragma unroll
                        for (int bt = 0; bt < NB; ++bt) {
                            acc[0][T][bt][H] = mfma_corr16<0>(w[SL][3 * T + 0], wsc[T], xc0[H][bt], xc1[H][bt], sbv, acc[0][T][bt][H]);
                            acc[1][T][bt][H] = mfma_corr16<1>(w[SL][3 * T + 1], wsc[T], xc0[H][bt], xc1[H][bt], sbv, acc[1][T][bt][H]);
                            acc[2][T][bt][H] = mfma_corr16<2>(w[SL][3 * T + 2], wsc[T], xc0[H][bt], xc1[H][bt], sbv, acc[2][T][bt][H]);
                        }
                    CCSM_FENCE;
                });
            }
            if constexpr (U + 2 < 12) ld_set(std::integral_constant<int, U + 2>{});      // the slot just used takes the set of two uses ahead
            else if constexpr (U == 10) { wxc[0][0] = w_at(OFF_C); wxc[0][1] = w_at(OFF_C + 1024); wxc[1][0] = w_at(OFF_C + 2048); wxc[1][1] = w_at(OFF_C + 3072); }
            CCSM_FENCE;
        });
        // r = sigmoid(R) ; N = b_in + r * N
        mfma_drain();
        {
            f32x4 b2[2];
            bias_set(2, b2);
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[2][T][bt][h][i] = b2[T][i] + sigmoid_f(acc[0][T][bt][h][i]) * acc[2][T][bt][h][i];
        }
        // ---------------- phase C: N += W_in x_t (x_t is still in its buffer) ------------------------------------------------------
        rd_x0();
        CCSM_FENCE;
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) acc[2][T][bt][h] = mfma32k_first(wxc[T][0], x0[h][bt], acc[2][T][bt][h]);
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) acc[2][T][bt][h] = mfma32k(wxc[T][1], x0[h][bt], acc[2][T][bt][h]);
        CCSM_FENCE;
        mfma_drain();
        ld_first();                                                 // the next step's first weight fragments: in flight during the tail
        CCSM_FENCE;
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[1][T][bt][h][i] = sigmoid_f(acc[1][T][bt][h][i]);
        __syncthreads();                                            // every wave has read h_{t-1} (phase B) before anybody overwrites its fragments
        // ---------------- tail: n = tanh(N); h' = n + z (h_{t-1} - n); hi fragments, blobs and residuals for the next step and layer ----
        {
            const int a = opq(lxs), l16 = opq(lane16);
            const int lo_ = opq(lpos) + ((opq(lane) & 32) << 6);    // the same inside the output's [kb][hl] fragments
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                char* p_hi = smem + own + a + bt * 2048;                                     // own 16 bytes of the hi fragments (half 0; + 256: half 1)
                char* p_c0 = smem + own + ((bt * 2 + 1) << 10) + l16;                         // blob bytes 0-15: corr fragment of the wave's first k-block
                char* p_c1 = smem + own + (((NB + bt) * 2 + 1) << 10) + l16;                  // bytes 16-23 | this lane's residuals of half 0
                char* p_lo = smem + LO_OFF + ((wave * NB + bt) * 64) * 8 + (l16 >> 1);        // residuals of half 1
                const uint2 l8[2] = {make_uint2(reinterpret_cast<const uint4*>(p_c1)->z, reinterpret_cast<const uint4*>(p_c1)->w),
                                     *reinterpret_cast<const uint2*>(p_lo)};
                float hn[2][8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const half8 hv = as_half8(*reinterpret_cast<const uint4*>(p_hi + h * 256));
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int lo4 = (int)(j < 4 ? l8[h].x : l8[h].y);
                        const float lo = (j & 3) == 0 ? __builtin_amdgcn_cvt_f32_fp8(lo4, 0) : (j & 3) == 1 ? __builtin_amdgcn_cvt_f32_fp8(lo4, 1)
                                       : (j & 3) == 2 ? __builtin_amdgcn_cvt_f32_fp8(lo4, 2) : __builtin_amdgcn_cvt_f32_fp8(lo4, 3);
                        const float hp = (float)hv[j] + lo * (1.0f / kMxLoScale);
                        const float nn = tanh_fold(acc[2][j >> 2][bt][h][j & 3]);
                        hn[h][j] = (hp - nn) * acc[1][j >> 2][bt][h][j & 3] + nn;
                    }
                }
                uint4 hi[2], c0;
                uint2 lo8[2], c1;
                mx16_pack<false>(hn, 0.25f, hi, lo8, c0, c1);
                const uint4 c1w = make_uint4(c1.x, c1.y, lo8[0].x, lo8[0].y);
                *reinterpret_cast<uint4*>(p_hi) = hi[0];
                *reinterpret_cast<uint4*>(p_hi + 256) = hi[1];
                *reinterpret_cast<uint4*>(p_c0) = c0;
                *reinterpret_cast<uint4*>(p_c1) = c1w;
                *reinterpret_cast<uint2*>(p_lo) = lo8[1];
                char* o = reinterpret_cast<char*>(out + (((size_t)(tile0 + bt) * kSeqLen + t) * kKB12 + (dir * kKBH + 2 * wave)) * 2 * kFragU4);
                nt_store(hi[0], reinterpret_cast<uint4*>(o + (uint32_t)lo_));
                nt_store(hi[1], reinterpret_cast<uint4*>(o + (uint32_t)lo_ + 256));
                nt_store(c0, reinterpret_cast<uint4*>(o + 1024 + (uint32_t)l16));
                nt_store(c1w, reinterpret_cast<uint4*>(o + 3072 + (uint32_t)l16));
            }
        }
        CCSM_FENCE;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // no transfer may still be writing LDS when the workgroup retires
}
#undef CCSM_FENCE

}  // namespace ccsm
