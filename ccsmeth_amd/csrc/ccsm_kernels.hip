// libccsm device code: hand-written HIP for gfx950 (CDNA4 / MI355X) of the ccsmeth attbigru2s forward.
//
// Reference behaviour being reproduced (paths relative to /root/reference/):
//   ccsmeth/models.py:89-150        ModelAttRNN.forward (embedding+kinetics concat, 2-strand shared 3-layer BiGRU,
//                                   additive attention pool, FC, softmax)
//   ccsmeth/utils/attention.py:48-70  Bahdanau attention
//   torch.nn.GRU (ATen gru_cell)    r,z,n gate equations, h' = (h - n) * z + n
//
// Design: the contraction work (244 MFLOP / CpG site) runs on v_mfma_f32_32x32x16_f16 with
// split-fp16 operands (v = hi + lo; products hi*hi + lo*hi + hi*lo accumulate in fp32, i.e. ~22-bit mantissa
// products, fp32-class accuracy) in the transposed form  G^T[unit][row] = W[unit][k] * X^T[k][row]  so that the MFMA
// result layout (a lane holds 16 units of ONE batch row) converts to the next step's B operand with a single
// half-wave exchange.  One workgroup = 8 waves = 32*NB batch rows of one direction; wave w owns hidden units
// [32w, 32w+32) of all three gates; weights stream from L2 straight into registers as pre-packed fragments;
// the hidden state lives in LDS as fragments; layer outputs go to HBM as fragments for the next layer.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "ccsm_layout.h"

namespace ccsm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// non-temporal 16-byte accesses for streaming data (keeps the re-used weight fragments resident in the XCD's L2)
__device__ __forceinline__ uint4 nt_load(const uint4* p) {
    return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)));
}
__device__ __forceinline__ void nt_store(uint4 v, uint4* p) {
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(p));
}
// Buffer-descriptor addressing: a wave-uniform base (4 SGPRs) + a wave-uniform byte offset (1 SGPR) + lane*16 (ONE VGPR
// shared by every access) instead of a 64-bit per-lane pointer per stream.  In the GRU kernel this removes the address
// VGPR pairs whose spills forced `s_waitcnt vmcnt(0)` (a full drain of the weight prefetch) in front of every reload.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store(uint4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
// 16 bytes per lane from global memory straight into LDS (lane l lands at lds_base + 16 l), issued as inline assembly ON PURPOSE:
// for the builtin the compiler cannot tell which LDS bytes the transfer writes and puts an s_waitcnt vmcnt(0) in front of every
// later ds_read, i.e. the chunk being prefetched had to land before the chunk already in LDS could be multiplied.  The caller
// orders the transfer itself: s_waitcnt vmcnt(0) (wait_dma) before the barrier that publishes the buffer.
// (m0 carries the LDS address; nothing else in these kernels uses m0, and the compiler does not accept it as a clobber.)
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_base) : "memory");
}
// the same through a buffer descriptor: wave-uniform byte offset `soff` in an SGPR, per-lane offset `voff` in ONE VGPR (no 64-bit
// per-lane address pair)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t dma_rsrc(const void* base) {
    const unsigned long long b = (unsigned long long)base;
    return u32x4_t{(unsigned)__builtin_amdgcn_readfirstlane((int)b), (unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) & 0xffffu,
                   0x7fffffffu, 0x00020000u};
}
__device__ __forceinline__ void dma16_buf(u32x4_t rsrc, int voff, int soff, unsigned lds_base) {
    // cache policy of the transfers (A/B builds: -DCCSM_DMA_POLICY="\" nt\""): measured on MI355X, `nt` costs 3.4 % (phase C's second read of
    // x_t finds less of it in L2), `sc1` nothing: profiles/r03_u_dma_policy.log
#ifndef CCSM_DMA_POLICY
#define CCSM_DMA_POLICY ""
#endif
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen" CCSM_DMA_POLICY " lds" : : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base)
                 : "memory");
}
// the same with the non-temporal policy: the LAST read of a line (experiment -DCCSM_DMA_C_NT: phase C's second pass over x_t)
__device__ __forceinline__ void dma16_buf_nt(u32x4_t rsrc, int voff, int soff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen nt lds" : : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base)
                 : "memory");
}
// Start-time stagger of the GRU workgroups (diagnostic, only in a -DCCSM_STAGGER_DIAG build; CCSM_STAGGER=<n> at ccsm_create): workgroup
// pair p waits (p & 3) * n sleeps of 127 x 64 cycles before its first instruction, so that the compute units are not all in the same phase
// of the step at the same time.  Measured (profiles/r05_p_stagger.log): nothing to gain - the kernels get slower by the delay itself.
#ifdef CCSM_STAGGER_DIAG
__device__ int g_ccsm_stagger = 0;
__device__ __forceinline__ void start_stagger(int group) {
    const int n = __builtin_amdgcn_readfirstlane(g_ccsm_stagger) * group;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
}
#else
__device__ __forceinline__ void start_stagger(int) {}
#endif
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ half8 as_half8(uint4 v) { return __builtin_bit_cast(half8, v); }
__device__ __forceinline__ half4 as_half4(uint2 v) { return __builtin_bit_cast(half4, v); }

__device__ __forceinline__ f32x16 mfma16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(a), as_half8(b), c, 0, 0, 0);
}

// v = hi + lo split.  hi = RNE fp16(v); lo = fp16(v - hi) (exact difference in fp32).
__device__ __forceinline__ void split16(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}
__device__ __forceinline__ uint32_t pack2(_Float16 a, _Float16 b) {
    half2v t = {a, b};
    return __builtin_bit_cast(uint32_t, t);
}

#ifdef CCSM_SIGMOID_EXP2   // experiment: exp2 on the pre-scaled argument, no range fix-up (exp2(+big) = inf -> rcp = 0, exp2(-big) = 0 -> rcp(1) = 1)
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }
#else
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
#endif
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

// Split-fp16 product of one k-block (CCSM_PRECISION_SPLIT3): hi*hi + hi*lo + lo*hi, fp32 accumulate.  Callers that hold several
// accumulators issue pass-major (mma_pass) so that consecutive MFMAs hit different accumulators.
template <int P>
__device__ __forceinline__ f32x16 mma_pass(const uint4 (&w)[2], const uint4 (&x)[2], f32x16 c) {
    return mfma16(w[P == 2 ? 1 : 0], x[P == 1 ? 1 : 0], c);
}
__device__ __forceinline__ f32x16 mma_split3(const uint4 (&w)[2], const uint4 (&x)[2], f32x16 c) {
    c = mma_pass<0>(w, x, c);
    c = mma_pass<1>(w, x, c);
    return mma_pass<2>(w, x, c);
}

// ---------------------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller: device-side N(0,1) initial states (production mode; the reference draws
// torch.randn on the CPU, unseeded in its workers — models.py:77-87 — so only the distribution is defined).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// A "slice" is one caller batch of n_sites CpG sites occupying strand rows [row_base, row_base + 2*n_sites) of a
// workspace: rows [row_base, row_base + n_sites) are strand 1, the next n_sites are strand 2.  Several slices packed
// back to back are run by ONE launch of each heavy kernel (micro-batch coalescing, ccsm_group_*).
//
// h0buf[ld][row][256] fp32, ld = 2*layer + dir, row stride = rows_cap (the workspace's padded row capacity).
// mode 0: explicit (src1/src2 = (6, n_sites, 256) per strand, reference init_hidden layout); 1: zeros; 2: Philox N(0,1).
// Philox counter of a drawn value: (key, sub | strand*6+ld, u4) with key = offset + site index and sub = 0 by default; with
// site_key (and optionally site_sub) the caller names every site's stream itself - call_mods uses (hash of the read name, position
// of the C in the read), so that a site's initial states do not depend on where its read stands in the file, on the batching or
// on which GPU processes it.
__global__ void prep_h0_kernel(float* __restrict__ h0buf, const float* __restrict__ src1, const float* __restrict__ src2,
                               int n_sites, int row_base, int rows_cap, int mode, uint64_t seed, uint64_t offset,
                               const unsigned long long* __restrict__ site_key, const unsigned int* __restrict__ site_sub) {
    const size_t total4 = (size_t)2 * kLayers * 2 * n_sites * (kHidden / 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int u4 = (int)(i % (kHidden / 4));
        const size_t rr = i / (kHidden / 4);
        const int rl = (int)(rr % (2 * n_sites));
        const int ld = (int)(rr / (2 * n_sites));
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mode != 1) {
            const int strand = rl >= n_sites;
            const int site = rl - strand * n_sites;
            if (mode == 0) {
                const float* src = strand ? src2 : src1;
                v = *reinterpret_cast<const float4*>(src + ((size_t)ld * n_sites + site) * kHidden + u4 * 4);
            } else {
                // counter = (key, sub << 4 | strand*6+ld, u4) -> 4 uniforms -> 4 normals (2 Box-Muller pairs)
                const uint64_t gs = site_key ? (uint64_t)site_key[site] : offset + (uint64_t)site;
                const uint32_t sub = site_key && site_sub ? site_sub[site] << 4 : 0u;
                uint32_t r[4];
                philox4x32_10((uint32_t)gs, (uint32_t)(gs >> 32), sub | (uint32_t)(strand * 6 + ld), (uint32_t)u4,
                              (uint32_t)seed, (uint32_t)(seed >> 32), r);
                const float k2pi = 6.283185307179586f, inv = 2.3283064365386963e-10f;  // 2^-32
                const float u0 = ((float)r[0] + 1.0f) * inv, u1 = (float)r[1] * inv;
                const float u2 = ((float)r[2] + 1.0f) * inv, u3 = (float)r[3] * inv;
                const float ra = sqrtf(-2.0f * __logf(fminf(u0, 1.0f))), rb = sqrtf(-2.0f * __logf(fminf(u2, 1.0f)));
                v = make_float4(ra * __cosf(k2pi * u1), ra * __sinf(k2pi * u1), rb * __cosf(k2pi * u3), rb * __sinf(k2pi * u3));
            }
        }
        *reinterpret_cast<float4*>(h0buf + ((size_t)ld * rows_cap + row_base + rl) * kHidden + u4 * 4) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Layer-0 input fragments: x = cat(embed[kmer.int()], ipd, pw, npass) (models.py:91-106), 11 features padded to
// one k-block of 16.  x0[tile][t][0][hl][lane] with lane (n,g): g=0 -> embedding dims 0..7, g=1 -> ipd,pw,npass,0...
// ---------------------------------------------------------------------------------------------------------
constexpr int kMaxSlices = 16;
struct SliceTable {          // strand rows of slice i: [row_base[i], row_base[i] + 2 * n_sites[i])
    int count;
    int row_base[kMaxSlices];
    int n_sites[kMaxSlices];
};

struct StrandDev {
    const void* kmer;   // (N,21) u8 codes, or f32 if kmer_is_f32
    const float* ipd;   // (N,21)
    const float* pw;    // (N,21)
    const float* npass; // (N) or (N,21) if npass_per_base
    const float* ipd_std = nullptr;  // (N,21)  is_stds
    const float* pw_std = nullptr;   // (N,21)  is_stds
    const float* sn = nullptr;       // (N,4)   is_sn
    const float* map = nullptr;      // (N,21)  is_map
};

// feature flags of the model variant (ccsm_config.is_npass / is_stds / is_sn / is_map): which of the optional planes follow
// [embedding(8) | ipd | pw] in a row of the layer-0 input, in the reference's concatenation order (models.py:100-123)
constexpr int kFeatNpass = 1, kFeatStds = 2, kFeatSn = 4, kFeatMap = 8;

// fold: the 17- and 18-column variants (is_npass + is_stds + is_sn [+ is_map]) do not fit the 16-wide k-block as [embedding(8) | features];
// for them ccsm_create folds the embedding table into the layer-0 matrix (W'[:, c] = W[:, 0:8] E[c] for the N_VOCAB = 5 codes) and the
// row becomes [one-hot(5) | features] = 14 or 15 columns: the same product, W[:, 0:8] E[code] = W'[:, code].
__global__ void pack_x0_kernel(uint4* __restrict__ x0, StrandDev s1, StrandDev s2, const float* __restrict__ embed,
                               int n_sites, int row_base, int kmer_is_f32, int npass_per_base, int feat, int fold) {
    const int total = 2 * n_sites * kSeqLen * 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int g = i & 1;
        const int t = (i >> 1) % kSeqLen;
        const int rl = (i >> 1) / kSeqLen;
        const int row = row_base + rl;
        const int tile = row >> 5, lane = (row & 31) + 32 * g;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float f[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // the columns behind the embedding
        const int strand = rl >= n_sites;
        const int site = rl - strand * n_sites;
        const StrandDev& s = strand ? s2 : s1;
        const size_t e = (size_t)site * kSeqLen + t;
        if (g == 1 || fold) {
            f[0] = s.ipd[e];
            f[1] = s.pw[e];
            int k = 2;
            if (feat & kFeatNpass) f[k++] = npass_per_base ? s.npass[e] : s.npass[site];
            if (feat & kFeatStds) { f[k++] = s.ipd_std[e]; f[k++] = s.pw_std[e]; }
            if (feat & kFeatSn)
                for (int j = 0; j < 4; ++j) f[k++] = s.sn[(size_t)site * 4 + j];    // one signal-to-noise quadruple per site, every position
            if (feat & kFeatMap) f[k++] = s.map[e];
        }
        if (g == 0) {
            int code = kmer_is_f32 ? (int)reinterpret_cast<const float*>(s.kmer)[e]
                                   : (int)reinterpret_cast<const uint8_t*>(s.kmer)[e];
            code = code < 0 ? 0 : (code >= kVocab ? kVocab - 1 : code);
            if (fold) {
#pragma unroll
                for (int j = 0; j < kVocab; ++j) v[j] = j == code ? 1.f : 0.f;
                v[5] = f[0]; v[6] = f[1]; v[7] = f[2];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = embed[code * kEmbed + j];
            }
        } else if (fold) {
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = f[3 + j];                   // columns 8..14
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = f[j];                       // at most 6 optional features without the fold (checked by ccsm_create)
        }
        _Float16 hi[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) split16(v[j], hi[j], lo[j]);
        uint4* dst = x0 + ((size_t)(tile * kSeqLen + t) * 2) * kFragU4 + lane;
        dst[0] = make_uint4(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]), pack2(hi[4], hi[5]), pack2(hi[6], hi[7]));
        dst[kFragU4] = make_uint4(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]), pack2(lo[4], lo[5]), pack2(lo[6], lo[7]));
    }
}

// ---------------------------------------------------------------------------------------------------------
// GRU layer in split-fp16 arithmetic (CCSM_PRECISION_SPLIT3): 96 batch rows per workgroup (NB = 3 tiles), three accumulator
// sets, x staged through LDS.  grid = 2 * (rows_p / 96); block = 512 (8 waves, 2 per SIMD, <= 256 VGPR); blockIdx.x & 1 =
// direction, so even XCDs (blockIdx % 8) hold the forward weights in L2 and odd XCDs the backward.
//   bias : [dir][wave][4 sets: r=b_ir+b_hr, z=b_iz+b_hz, nx=b_in, nh=b_hn][hh][16] in MFMA C-row order
//   h0   : [dir][rows_p][256] fp32 (this layer's two slabs)
// The whole (layer, direction) weight set (2.36 MB of fragments) streams from L2 every timestep; 96 rows amortise it over
// 1.5x the MFMA work of 64.  The accumulators for 96 rows x 4 quantities (r, z, n_x, n_h) do not fit in
// 256 VGPRs next to double-buffered weight fragments, so each step runs in three phases over three sets:
//   A  x-part of r and z            (K = 16*KX)   sets R, Z
//   B  h-part of r, z and n         (K = 256)     sets R, Z, N(= W_hn h + b_hn)
//      r = sigmoid(R);  N = b_in + r * N
//   C  x-part of n                  (K = 16*KX)   set N
//   epilogue: z = sigmoid(Z), n = tanh(N), h' = (h - n) * z + n
// x_t is read twice (A and C).  It is staged once per pass into LDS in chunks of 4 k-blocks (24 KiB, two buffers) by
// plain global loads issued one chunk ahead and written to LDS at the end of the previous chunk, so every wave reads
// each x fragment from LDS instead of 8 waves pulling it through L1/L2.  The 16 chunk barriers of a step also order
// the hidden-state fragments (no extra barriers).
//   wst : [dir][wave][ A: KX x (r,z) x hl | B: 16 x (r,z,n) x hl | C: KX x (n) x hl ][64] uint4
// ---------------------------------------------------------------------------------------------------------
template <int KX>
__global__ __launch_bounds__(512, 2) void gru_layer_v2_kernel(const uint4* __restrict__ xin, uint4* __restrict__ out,
                                                               const uint4* __restrict__ wst, const float* __restrict__ bias,
                                                               const float* __restrict__ h0, int rows_p) {
    constexpr int NB = 3;
    constexpr int CK = KX >= 4 ? 4 : KX;       // k-blocks per staged chunk
    constexpr int NCH = KX / CK;               // chunks per pass over x_t
    constexpr int CHF = CK * NB * 2;           // fragments per chunk
    constexpr int SPW = (CHF + kWaves - 1) / kWaves;  // fragments staged per wave per chunk
    constexpr int FA = 4, FB = 6, FC = 2;      // fragments per k-block in phases A, B, C
    constexpr int OFF_B = KX * FA, OFF_C = OFF_B + kKBH * FB, WFRAGS = OFF_C + KX * FC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_h = smem;                               // h fragments  [kb 16][bt 3][hl 2] x 1 KiB = 96 KiB
    char* s_x = smem + kKBH * NB * 2 * 1024;        // x chunk ring [buf 2][kbl CK][bt 3][hl 2] x 1 KiB
    float* s_bias = reinterpret_cast<float*>(s_x + 2 * CHF * 1024);   // [wave][set 4][hh 2][16] fp32 = 4 KiB (this direction's biases)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dir = blockIdx.x & 1;
    const int tile0 = (blockIdx.x >> 1) * NB;
    const int n = lane & 31, hh = lane >> 5;
    // biases -> LDS once; the accumulator sets are initialised from there every step (no global loads on the step path)
    if (threadIdx.x < kWaves * 4 * 32 / 4)
        reinterpret_cast<float4*>(s_bias)[threadIdx.x] = reinterpret_cast<const float4*>(bias + (size_t)dir * kWaves * 4 * 32)[threadIdx.x];

    auto hfrag = [&](int kb, int bt, int hl) -> char* { return s_h + (((kb * NB + bt) * 2 + hl) << 10); };
    auto xfrag = [&](int buf, int kbl, int bt, int hl) -> char* { return s_x + ((((buf * CK + kbl) * NB + bt) * 2 + hl) << 10); };

    // ---- h0 -> LDS fragments (this wave's own two k-blocks, every batch tile)
    {
        const float* h0d = h0 + (size_t)dir * rows_p * kHidden;
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            const float* src = h0d + ((size_t)(tile0 + bt) * 32 + n) * kHidden;
#pragma unroll
            for (int kbl = 0; kbl < 2; ++kbl) {
                const int kb = 2 * wave + kbl;
                const float4 a = *reinterpret_cast<const float4*>(src + kb * 16 + hh * 8);
                const float4 b = *reinterpret_cast<const float4*>(src + kb * 16 + hh * 8 + 4);
                const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                _Float16 hi[8], lo[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) split16(v[j], hi[j], lo[j]);
                *reinterpret_cast<uint4*>(hfrag(kb, bt, 0) + lane * 16) =
                    make_uint4(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]), pack2(hi[4], hi[5]), pack2(hi[6], hi[7]));
                *reinterpret_cast<uint4*>(hfrag(kb, bt, 1) + lane * 16) =
                    make_uint4(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]), pack2(lo[4], lo[5]), pack2(lo[6], lo[7]));
            }
        }
    }

    // ---- x staging: fragment f = (kbl*NB + bt)*2 + hl of chunk c of timestep t; this wave moves f = wave + 8*i
    // (three named registers rather than an array: the array form is demoted to scratch memory by hipcc)
    static_assert(SPW <= 3, "staging registers");
    uint4 sreg0, sreg1, sreg2;
    const int lane16 = lane * 16;
    // the descriptor starts at THIS workgroup's first tile: its 2 GiB range and the 32-bit offsets never see more than NB tiles
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(xin + (size_t)tile0 * kSeqLen * KX * 2 * kFragU4);
    auto stage_off = [&](int t, int c, int i) -> int {      // wave-uniform byte offset of the fragment this wave stages
        // branch-free: a wave with nothing left to move re-stages the last fragment (same bytes, same place)
        const int f = (CHF % kWaves == 0) ? wave + kWaves * i : min(wave + kWaves * i, CHF - 1);
        const int hl = f & 1, bt = (f >> 1) % NB, kbl = (f >> 1) / NB;
        return ((((bt * kSeqLen + t) * KX + (c * CK + kbl)) * 2 + hl) << 10);
    };
    auto stage_dst = [&](int buf, int i) -> uint4* {
        const int f = (CHF % kWaves == 0) ? wave + kWaves * i : min(wave + kWaves * i, CHF - 1);
        return reinterpret_cast<uint4*>(s_x + ((buf * CHF + f) << 10) + lane * 16);
    };
    auto stage_load = [&](int t, int c) {
        sreg0 = buf_load(xrs, lane16, stage_off(t, c, 0));
        if constexpr (SPW > 1) sreg1 = buf_load(xrs, lane16, stage_off(t, c, 1));
        if constexpr (SPW > 2) sreg2 = buf_load(xrs, lane16, stage_off(t, c, 2));
    };
    auto stage_store = [&](int buf) {
        *stage_dst(buf, 0) = sreg0;
        if constexpr (SPW > 1) *stage_dst(buf, 1) = sreg1;
        if constexpr (SPW > 2) *stage_dst(buf, 2) = sreg2;
    };

    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(wst + (size_t)(dir * kWaves + wave) * WFRAGS * kFragU4);   // wave-uniform
    const float* bp = s_bias + wave * 4 * 32 + hh * 16;

    stage_load(dir ? kSeqLen - 1 : 0, 0);
    stage_store(0);

    for (int s = 0; s < kSeqLen; ++s) {
        const int t = dir ? (kSeqLen - 1 - s) : s;
        const int tn = s + 1 < kSeqLen ? (dir ? t - 1 : t + 1) : t;   // next step's timestep (last step: harmless reload)
        f32x16 acc[3][NB];                            // R, Z, N
        auto bias_set = [&](int set) {                // from LDS
            f32x16 b;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(bp + set * 32 + q * 4);
                b[4 * q + 0] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
            }
            return b;
        };
        {
            const f32x16 b0 = bias_set(0), b1 = bias_set(1);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) { acc[0][bt] = b0; acc[1][bt] = b1; }
        }

        // Operand movers and one-k-block MFMA groups.  The schedule is explicit (compiler-level memory fences around every
        // MFMA group) because, left alone under this register pressure, hipcc sinks the weight loads next to their uses
        // and the L2 latency is exposed on every k-block.  Measured with s_memtime (tools/gpu_phases.py): a weight
        // prefetch needs ~1700 cycles of MFMA cover (both waves of the SIMD) to be free, i.e. one k-block ahead in
        // phase B (27 MFMAs per k-block) but two ahead in phase A (18) and four ahead in phase C (9).  Register plan:
        //   phase A : accumulators R,Z (96)   + weight ring of 4 k-blocks (64)            + x (24) + staging (12)
        //   phase B : accumulators R,Z,N (144) + weights double-buffered (48)              + h (24)
        //   phase C : accumulators Z,N (96)   + weights of two whole chunks (2 x 32 = 64) + x (24) + staging (12)
        // x / h fragments are read from LDS at the start of their k-block (short latency, covered by the other wave).
        uint4 wa[4][2][2];        // phase A ring: [slot][gate r,z][hl]
        uint4 wq[2][3][2];        // phase B double buffer: [buf][gate r,z,n][hl]
        uint4 wc[2][4][2];        // phase C: [chunk parity][k-block in chunk][hl]   (KX == 1: wc[0][0] only)
        uint4 xq[2][NB][2];       // x / h fragments; phases A and C read the next k-block's while the current one multiplies
        auto w_at = [&](int frag) -> uint4 { return buf_load(wrs, lane16, frag << 10); };
        auto ldA = [&](uint4 (&dst)[2][2], int kb) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) dst[g][hl] = w_at(kb * FA + g * 2 + hl);
        };
        auto ldB = [&](uint4 (&dst)[3][2], int kb) {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) dst[g][hl] = w_at(OFF_B + kb * FB + g * 2 + hl);
        };
        auto ldC = [&](uint4 (&dst)[4][2], int c) {     // the n-gate fragments of a whole chunk
#pragma unroll
            for (int j = 0; j < CK; ++j)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) dst[j][hl] = w_at(OFF_C + (c * CK + j) * FC + hl);
        };
        auto rdx = [&](uint4 (&x)[NB][2], int buf, int kbl) {
#pragma unroll
            for (int hl = 0; hl < 2; ++hl)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) x[bt][hl] = *reinterpret_cast<const uint4*>(xfrag(buf, kbl, bt, hl) + lane * 16);
        };
        auto rdh = [&](uint4 (&x)[NB][2], int kb) {
#pragma unroll
            for (int hl = 0; hl < 2; ++hl)
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) x[bt][hl] = *reinterpret_cast<const uint4*>(hfrag(kb, bt, hl) + lane * 16);
        };
        // A compiler-level memory barrier: the operand loads issued before it may not sink below it (MFMAs are free to
        // move).  (__builtin_amdgcn_sched_barrier(0) here produced NaNs on ROCm 7.2 / gfx950 — do not use it.)
#define CCSM_FENCE asm volatile("" ::: "memory")
        // one k-block: gate g of `w` (g < G) into accumulator set S0 + g; pass-major issue order
#define CCSM_MM(W, X, G, S0)                                                                                        \
    do {                                                                                                            \
        CCSM_FENCE;                                                                                                 \
        _Pragma("unroll") for (int bt = 0; bt < NB; ++bt) _Pragma("unroll") for (int g = 0; g < G; ++g)             \
            acc[S0 + g][bt] = mfma16(W[g][0], X[bt][0], acc[S0 + g][bt]);                                           \
        _Pragma("unroll") for (int bt = 0; bt < NB; ++bt) _Pragma("unroll") for (int g = 0; g < G; ++g)             \
            acc[S0 + g][bt] = mfma16(W[g][0], X[bt][1], acc[S0 + g][bt]);                                           \
        _Pragma("unroll") for (int bt = 0; bt < NB; ++bt) _Pragma("unroll") for (int g = 0; g < G; ++g)             \
            acc[S0 + g][bt] = mfma16(W[g][1], X[bt][0], acc[S0 + g][bt]);                                           \
        CCSM_FENCE;                                                                                                 \
    } while (0)

        // ---------------- phase A: R, Z += W_i{r,z} x_t  (fragment index of k-block kb: kb*FA + g*2 + hl) ----------
        // Every register slot is (re)loaded unconditionally — clamped indices, redundant reloads at the ends — because a
        // conditional load would keep the slot's old value live across the whole step loop and spill the accumulators.
        ldA(wa[0], 0);
        if constexpr (CK == 4) ldA(wa[1], 1);
        if constexpr (CK == 4) {
#pragma unroll 1
            for (int c = 0; c < NCH; ++c) {
                __syncthreads();                       // chunk c (buffer c&1) is in LDS; the other buffer is free
                const int buf = c & 1;
                const bool more = c + 1 < NCH;
                rdx(xq[0], buf, 0);
#define CCSM_KA(J)                                                                                 \
    ldA(wa[(J + 2) & 3], min(c * 4 + J + 2, KX - 1));   /* two k-blocks ahead */                   \
    if (J == 0) stage_load(t, more ? c + 1 : 0); /* after the weight prefetch: younger in vmcnt */ \
    if (J < 3) rdx(xq[(J + 1) & 1], buf, J + 1);        /* next k-block's x fragments */           \
    CCSM_MM(wa[J], xq[J & 1], 2, 0);
                CCSM_KA(0);
                CCSM_KA(1);
                CCSM_KA(2);
                CCSM_KA(3);
#undef CCSM_KA
                stage_store((c + 1) & 1);              // next A chunk, or C chunk 0 into buffer NCH & 1 == 0
            }
            ldB(wq[0], 0);                             // first recurrent k-block
        } else {
            __syncthreads();
            ldB(wq[0], 0);
            stage_load(tn, 0);                         // KX == 1: the next step's only chunk
            rdx(xq[0], s & 1, 0);
            CCSM_MM(wa[0], xq[0], 2, 0);
            stage_store((s + 1) & 1);
        }

        // ---------------- phase B: R, Z, N += W_h{r,z,n} h_{t-1}  (N starts at b_hn) ---------------------------
        // entering: recurrent k-block 0 is in wq[0]
        {
            const f32x16 b3 = bias_set(3);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) acc[2][bt] = b3;
        }
#pragma unroll 1
        for (int kb = 0; kb < kKBH; kb += 2) {
            ldB(wq[1], kb + 1);
            rdh(xq[0], kb);
            CCSM_MM(wq[0], xq[0], 3, 0);
            ldB(wq[0], min(kb + 2, kKBH - 1));
            rdh(xq[0], kb + 1);
            CCSM_MM(wq[1], xq[0], 3, 0);
        }
        ldC(wc[0], 0);                                 // n-gate weights of phase C's first chunk: covered by the VALU work below
        // r = sigmoid(R) ; N = b_in + r * N
        {
            const f32x16 b2 = bias_set(2);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[2][bt][r] = b2[r] + sigmoid_f(acc[0][bt][r]) * acc[2][bt][r];
        }

        // ---------------- phase C: N += W_in x_t.  Only 9 MFMAs per k-block: the n-gate weights of a whole chunk are
        // fetched one chunk (four k-blocks) ahead.
#define CCSM_MMC(WKB, X)                                                                                             \
    do {                                                                                                            \
        CCSM_FENCE;                                                                                                 \
        _Pragma("unroll") for (int bt = 0; bt < NB; ++bt) acc[2][bt] = mfma16(WKB[0], X[bt][0], acc[2][bt]);        \
        _Pragma("unroll") for (int bt = 0; bt < NB; ++bt) acc[2][bt] = mfma16(WKB[0], X[bt][1], acc[2][bt]);        \
        _Pragma("unroll") for (int bt = 0; bt < NB; ++bt) acc[2][bt] = mfma16(WKB[1], X[bt][0], acc[2][bt]);        \
        CCSM_FENCE;                                                                                                 \
    } while (0)
        if constexpr (CK == 4) {
#pragma unroll 1
            for (int c2 = 0; c2 < NCH; c2 += 2) {
#define CCSM_CHUNK_C(C, CUR, NXT)                                                                  \
    {                                                                                              \
        __syncthreads();                                                                           \
        const int buf = (C) & 1;                                                                   \
        const bool more = (C) + 1 < NCH;                                                           \
        ldC(wc[NXT], min((C) + 1, NCH - 1));             /* a whole chunk ahead */                \
        stage_load(more ? t : tn, more ? (C) + 1 : 0); /* next C chunk / next step's first A chunk */ \
        rdx(xq[0], buf, 0);                                                                        \
        rdx(xq[1], buf, 1); CCSM_MMC(wc[CUR][0], xq[0]);               \
        rdx(xq[0], buf, 2); CCSM_MMC(wc[CUR][1], xq[1]);               \
        rdx(xq[1], buf, 3); CCSM_MMC(wc[CUR][2], xq[0]);               \
        CCSM_MMC(wc[CUR][3], xq[1]);                                   \
        stage_store(((C) + 1) & 1);                                             \
    }
                CCSM_CHUNK_C(c2, 0, 1)
                CCSM_CHUNK_C(c2 + 1, 1, 0)
#undef CCSM_CHUNK_C
            }
        } else {
            rdx(xq[0], s & 1, 0);                 // x chunk of this step is still in its buffer
            CCSM_MMC(wc[0][0], xq[0]);
        }
#undef CCSM_MMC
#undef CCSM_MM
#undef CCSM_FENCE

        // ---------------- h_{t-1} of this wave's own units (C layout), then the gate epilogue -------------------
        float hprev[NB][16];
#pragma unroll
        for (int bt = 0; bt < NB; ++bt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kb = 2 * wave + (q >> 1);
                const int src_lane = n + 32 * (q & 1);
                const half4 hi = as_half4(*reinterpret_cast<const uint2*>(hfrag(kb, bt, 0) + src_lane * 16 + hh * 8));
                const half4 lo = as_half4(*reinterpret_cast<const uint2*>(hfrag(kb, bt, 1) + src_lane * 16 + hh * 8));
#pragma unroll
                for (int e = 0; e < 4; ++e) hprev[bt][4 * q + e] = (float)hi[e] + (float)lo[e];
            }
        if constexpr (CK != 4) __syncthreads();   // KX == 1: no phase-C barriers, so order the h_{t-1} reads explicitly

#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            uint32_t phi[8], plo[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float hn2[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float zz = sigmoid_f(acc[1][bt][r + e]);
                    const float nn = tanh_f(acc[2][bt][r + e]);
                    hn2[e] = (hprev[bt][r + e] - nn) * zz + nn;
                }
                _Float16 h0a, l0a, h1a, l1a;
                split16(hn2[0], h0a, l0a);
                split16(hn2[1], h1a, l1a);
                phi[r >> 1] = pack2(h0a, h1a);
                plo[r >> 1] = pack2(l0a, l1a);
            }
#pragma unroll
            for (int kbl = 0; kbl < 2; ++kbl) {
                uint4 v[2];
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    const uint32_t* p = hl ? plo : phi;
                    const uint32_t a0 = p[4 * kbl + 0], a1 = p[4 * kbl + 1], b0 = p[4 * kbl + 2], b1 = p[4 * kbl + 3];
                    const uint32_t own0 = hh ? b0 : a0, own1 = hh ? b1 : a1;
                    const uint32_t snd0 = hh ? a0 : b0, snd1 = hh ? a1 : b1;
                    const uint32_t rcv0 = __shfl_xor(snd0, 32), rcv1 = __shfl_xor(snd1, 32);
                    v[hl] = hh ? make_uint4(rcv0, rcv1, own0, own1) : make_uint4(own0, own1, rcv0, rcv1);
                }
                const int kb = 2 * wave + kbl;
                *reinterpret_cast<uint4*>(hfrag(kb, bt, 0) + lane * 16) = v[0];
                *reinterpret_cast<uint4*>(hfrag(kb, bt, 1) + lane * 16) = v[1];
                // plain 64-bit-address stores: the raw_buffer_store_b128 form of these two stores corrupted a few output
                // elements at random on ROCm 7.2 / gfx950 (data registers reused too early); loads are unaffected.
                uint4* o = out + (((size_t)(tile0 + bt) * kSeqLen + t) * kKB12 + (dir * kKBH + kb)) * 2 * kFragU4;
                o[lane] = v[0];
                o[kFragU4 + lane] = v[1];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Attention pool + FC partials (utils/attention.py:48-70, models.py:135-148), one strand-row at a time.
//   q = Wa h_n ; K_t = Ua out_t ; e_t = va . tanh(q + K_t) ; a = softmax_t(e) ; c = sum_t a_t out_t
//   logits = fc1 [c_strand1 | c_strand2] + b.  Since fc1 is linear, fc1_s . c = sum_t a_t (fc1_s . out_t): the
//   2x512 dot products p[t][row][class] are taken while the out_t fragments pass through LDS for the Ua GEMM, so
//   out is read once and c is never materialised.  Output: part[row][2] = strand-half of the logits.
// grid = rows_p / 32 (one batch tile per workgroup); block 512; wave w owns attention units [32w, 32w+32) and keeps
// q + K_t for 7 timesteps in 7 accumulators (init = q), so Ua streams from L2 three times per workgroup.  The
// activation fragments (shared by all 8 waves) are staged once per workgroup into LDS with global_load_lds
// (2 k-blocks x 7 timesteps x hi/lo = 28 KiB per chunk, double-buffered).
//   wa / ua : [wave][kb 32][hl][64] uint4 ; va : [wave][hh][16] floats (C-row order) ; fcw : fc1.weight (2,1024) fp32
// ---------------------------------------------------------------------------------------------------------
constexpr int kAttFc3 = kKB12 * 2 * 8 + 1;        // uint4 items of the compact fc1 fragments + the zero line (attn_fc_kernel, attn_fc_f8_kernel)
__global__ __launch_bounds__(512, 2) void attn_fc_kernel(const uint4* __restrict__ out2, const uint4* __restrict__ wa,
                                                          const uint4* __restrict__ ua, const float* __restrict__ va,
                                                          const uint4* __restrict__ fc3, float* __restrict__ part,
                                                          SliceTable slices) {
    constexpr int TG = 7;                      // timesteps per Ua pass (21 = 3 * 7)
    constexpr int CK = 2;                      // k-blocks per staged chunk
    constexpr int NCHUNK = kKB12 / CK;
    constexpr int CHUNK_FRAGS = CK * TG * 2;   // 28 fragments of 1 KiB
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_stage = smem;                                                     // [2][CK][TG][hl] fragments
    float* s_epart = reinterpret_cast<float*>(smem + 2 * CHUNK_FRAGS * 1024);  // [wave][t][32]
    float* s_pfc = s_epart + kWaves * kSeqLen * 32;                            // [t][32][2]: wave w < 7 owns timestep t0 + w of every group
    uint4* s_fc3 = reinterpret_cast<uint4*>(s_pfc + kSeqLen * 32 * 2);         // fc1.weight as compact A fragments [kb 32][hi|lo][q 2][i 4] + one zero line (pack_fc_s3)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x;
    const int n = lane & 31, hh = lane >> 5;

    for (int i = threadIdx.x; i < kAttFc3; i += blockDim.x) s_fc3[i] = fc3[i];

    const uint4* wap = wa + (size_t)wave * kKB12 * 2 * kFragU4 + lane;
    const uint4* uap = ua + (size_t)wave * kKB12 * 2 * kFragU4 + lane;
    const uint4* otile = out2 + (size_t)tile * kSeqLen * kKB12 * 2 * kFragU4;   // [t][kb][hl][64]

    // ---- q = Wa h_n, h_n = [fwd final state = out[t=L-1][0:256] | bwd final state = out[t=0][256:512]] (models.py:135-137)
    f32x16 qacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) qacc[r] = 0.f;
#pragma unroll 4
    for (int kb = 0; kb < kKB12; ++kb) {
        const uint4 w[2] = {wap[(kb * 2 + 0) * kFragU4], wap[(kb * 2 + 1) * kFragU4]};
        const int tq = kb < kKBH ? kSeqLen - 1 : 0;
        const uint4* xp = otile + ((size_t)tq * kKB12 + kb) * 2 * kFragU4 + lane;
        const uint4 x[2] = {xp[0], xp[kFragU4]};
        qacc = mma_split3(w, x, qacc);
    }
    float vav[16], vsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { vav[r] = va[(wave * 2 + hh) * 16 + r]; vsum += vav[r]; }
    int strand = 0;   // which half of fc1.weight this lane's row multiplies: strand 2 rows are the upper half of a slice
    {
        const int row = tile * 32 + n;
        for (int i = 0; i < slices.count; ++i)
            if (row >= slices.row_base[i] && row < slices.row_base[i] + 2 * slices.n_sites[i])
                strand = (row - slices.row_base[i]) >= slices.n_sites[i];
    }

    // stage fragment f = wave + 8 i of chunk `c` of timestep group t0 into buffer `buf`: f = (kbl * TG + tt) * 2 + hl
    auto stage_one = [&](int t0, int c, int buf, int i) {
        const int f = wave + kWaves * i;
        if (f < CHUNK_FRAGS) {
            const int hl = f & 1, tt = (f >> 1) % TG, kbl = (f >> 1) / TG;
            const uint4* src = otile + (((size_t)(t0 + tt) * kKB12 + (c * CK + kbl)) * 2 + hl) * kFragU4 + lane;
            dma16(src, __builtin_amdgcn_readfirstlane(
                           (unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_stage + (buf * CHUNK_FRAGS + f) * 1024)));
        }
    };
    constexpr int NST = (CHUNK_FRAGS + kWaves - 1) / kWaves;      // 4 transfers per wave and chunk (waves 4-7: 3)
    auto stage = [&](int t0, int c, int buf) {
#pragma unroll
        for (int i = 0; i < NST; ++i) stage_one(t0, c, buf, i);
    };
    // fc1 partials: wave w < 7 takes timestep t0 + w of the group for ALL k-blocks, as three more MFMAs per k-block on the operands it holds
    // for that timestep anyway: fc1.weight is the A operand of a 33rd unit tile whose rows 0-3 are (class, strand half) and whose other rows
    // are zero (lanes n >= 4 read the zero line).  Rounds 1-4 did this sum on the vector ALU of ONE wave per k-block (280 instructions that
    // every other wave waited for at the chunk's barrier).
    f32x16 facc;
    const int fc_lane = n < 4 ? hh * 4 + n : -1;
    uint4 wu[CK][2];       // Ua fragments of the next chunk, requested one chunk ahead
#pragma unroll
    for (int kbl = 0; kbl < CK; ++kbl) { wu[kbl][0] = uap[(kbl * 2 + 0) * kFragU4]; wu[kbl][1] = uap[(kbl * 2 + 1) * kFragU4]; }

    for (int tg = 0; tg < kSeqLen / TG; ++tg) {
        const int t0 = tg * TG;
        f32x16 kacc[TG];
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) kacc[tt] = qacc;     // accumulate K_t on top of q
#pragma unroll
        for (int r = 0; r < 16; ++r) facc[r] = 0.f;

        stage(t0, 0, 0);
#pragma unroll 1
        for (int c = 0; c < NCHUNK; ++c) {
            // Ua fragments of this chunk were requested a whole chunk ago (wu).  The next chunk's are requested right AFTER the
            // barrier, next to the LDS-DMA staging of the next activation chunk: the compiler drains vmcnt(0) in front of every
            // barrier (the DMA must have landed), so anything requested before it would be waited for on the spot.
            uint4 w[CK][2];
#pragma unroll
            for (int kbl = 0; kbl < CK; ++kbl) { w[kbl][0] = wu[kbl][0]; w[kbl][1] = wu[kbl][1]; }
            wait_dma();        // this wave's part of chunk c has arrived (see dma16)
            __syncthreads();   // chunk c has landed; buffer (c+1)&1 is free
            // The next chunk's requests - the wave's transfers and its four Ua fragments (Ua is re-streamed for every timestep group) - are
            // issued one per timestep behind the MFMAs of this chunk's first k-block and the start of its second: in a block behind the barrier
            // all eight waves queued on the CU's vector-memory path at once while no MFMA ran (profiles/r04_j_attn_stamps_before.log).
            const int cn = c + 1 < NCHUNK ? c + 1 : 0;
            const char* sb = s_stage + (c & 1) * CHUNK_FRAGS * 1024 + lane * 16;
#pragma unroll
            for (int kbl = 0; kbl < CK; ++kbl) {
                const int kb = c * CK + kbl;
#pragma unroll
                for (int tt = 0; tt < TG; ++tt) {
                    const uint4 x[2] = {*reinterpret_cast<const uint4*>(sb + ((kbl * TG + tt) * 2 + 0) * 1024),
                                        *reinterpret_cast<const uint4*>(sb + ((kbl * TG + tt) * 2 + 1) * 1024)};
                    kacc[tt] = mma_split3(w[kbl], x, kacc[tt]);
                    if (wave == tt) {
                        const uint4 fw[2] = {s_fc3[fc_lane < 0 ? kAttFc3 - 1 : (kb * 2 + 0) * 8 + fc_lane],
                                             s_fc3[fc_lane < 0 ? kAttFc3 - 1 : (kb * 2 + 1) * 8 + fc_lane]};
                        facc = mma_split3(fw, x, facc);
                    }
                    const int slot = kbl * TG + tt;          // 0..13: transfers in slots 0..3, Ua fragments in slots 4..7
                    if (slot < NST) { if (c + 1 < NCHUNK) stage_one(t0, c + 1, (c + 1) & 1, slot); }
                    else if (slot < NST + CK * 2) {
                        const int j = slot - NST;
                        wu[j >> 1][j & 1] = uap[((cn * CK + (j >> 1)) * 2 + (j & 1)) * kFragU4];
                    }
                    asm volatile("" ::: "memory");
                }
            }
        }
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) {
            // sum_r va_r tanh(k_r) = sum_r va_r - 2 sum_r va_r / (e^{2 k_r} + 1)
            float e = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) e += vav[r] * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(kacc[tt][r] * 2.8853900817779268f) + 1.0f);
            e = vsum - 2.0f * e;
            e += __shfl_xor(e, 32);
            if (hh == 0) s_epart[(wave * kSeqLen + t0 + tt) * 32 + n] = e;
        }
        if (wave < TG && hh == 0) {       // C rows 0-3 of the fc1 tile = registers 0-3 of the lower lanes: (class 0, strand), (class 1, strand)
            s_pfc[((t0 + wave) * 32 + n) * 2 + 0] = strand ? facc[1] : facc[0];
            s_pfc[((t0 + wave) * 32 + n) * 2 + 1] = strand ? facc[3] : facc[2];
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

// logits = strand-1 half + strand-2 half + fc1.bias ; probs = softmax(logits)   (models.py:145-150)
__global__ void finalize_kernel(const float* __restrict__ part, const float* __restrict__ fcb, float* __restrict__ logits,
                                float* __restrict__ probs, int n_sites, int row_base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sites) return;
    const float* p1 = part + ((size_t)row_base + i) * 2;
    const float* p2 = part + ((size_t)row_base + n_sites + i) * 2;
    const float l0 = p1[0] + p2[0] + fcb[0];
    const float l1 = p1[1] + p2[1] + fcb[1];
    const float m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m);
    const float inv = 1.0f / (e0 + e1);
    logits[(size_t)i * 2 + 0] = l0;
    logits[(size_t)i * 2 + 1] = l1;
    probs[(size_t)i * 2 + 0] = e0 * inv;
    probs[(size_t)i * 2 + 1] = e1 * inv;
}

// ---------------------------------------------------------------------------------------------------------
// MFMA layout self-test: C = A * B for one 32x32x16 tile with the fragment conventions used above
// (A[i][k] at lane i+32*(k>>3), elem k&7; B[k][j] likewise; C[i][j] at lane j+32*((i>>2)&1), reg (i&3)+4*(i>>3)).
// ---------------------------------------------------------------------------------------------------------
__global__ void mfma_selftest_kernel(const _Float16* __restrict__ a /*32x16*/, const _Float16* __restrict__ b /*16x32*/,
                                     float* __restrict__ c /*32x32*/) {
    const int lane = threadIdx.x & 63;
    const int n = lane & 31, g = lane >> 5;
    half8 af, bf;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        af[j] = a[n * 16 + 8 * g + j];
        bf[j] = b[(8 * g + j) * 32 + n];
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * g;
        c[i * 32 + n] = acc[r];
    }
}

}  // namespace ccsm
