// libccsm, read -> feature extraction on the GPU ("next" row 1).
//
// Reference behaviour reproduced (paths into /root/reference/ccsmeth/):
//   extract_features.py:327-334   CodecV1 LUT decode of fi/ri/fp/rp (utils/process_utils.py:426-449)
//   extract_features.py:181-199   per-read z-score in float64 (mean, std ddof=0, all-zero when std == 0), np.around(., 6)
//   extract_features.py:339-405   CG scan (loc = index of C), keep iff 10 <= loc < n-10 and 10 <= n-2-loc < n-10,
//                                 forward window seq/ipd_f/pw_f [loc-10, loc+10], reverse window = reverse-complement
//                                 k-mer and ri/rp (unflipped) at [n-12-loc, n+8-loc]
//   call_modifications.py:95-121  base codes A0 C1 G2 T3 other4, npass repeated over the 21 positions
//   models.py:91-106              x = cat(embed[kmer], ipd, pw, npass)
// Instead of building Python lists, the raw per-read byte arrays are uploaded once (5 B per base) and two kernels
// write the layer-0 input fragments of the BiGRU directly (what pack_x0_kernel builds from host features):
//   extract_stats_kernel : one workgroup per read: float64 mean / std of the four decoded arrays + number of kept sites
//   extract_pack_kernel  : one workgroup per read: ordered compaction of the kept CG sites, per-site rows written at
//                          row_base + first_site[read] + k (strand 1) and + n_sites_total (strand 2), plus loc list
// float64 sums of the integer frame counts are exact; the variance sum is a tree reduction (NumPy: pairwise), which can
// differ in the last ulp of std and therefore — after the 6-decimal rounding — essentially never (tests compare).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ccsm_layout.h"

namespace ccsm_extract {

using namespace ccsm;

constexpr int kPackParts = 8;      // workgroups per read of extract_pack_kernel

__device__ __forceinline__ int codec_v1(int c) {   // utils/process_utils.py:426-449
    return c < 64 ? c : (c < 128 ? 64 + 2 * (c - 64) : (c < 192 ? 192 + 4 * (c - 128) : 448 + 8 * (c - 192)));
}
__device__ __forceinline__ int base_code(uint8_t b) {   // process_utils.py:26-29 (upper-case alphabet)
    return b == 'A' ? 0 : (b == 'C' ? 1 : (b == 'G' ? 2 : (b == 'T' ? 3 : 4)));
}
__device__ __forceinline__ uint8_t comp_base(uint8_t b) {   // process_utils.py:12-15, unknown -> 'N'
    switch (b) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'M': return 'K'; case 'K': return 'M'; case 'R': return 'Y'; case 'Y': return 'R';
        case 'B': return 'V'; case 'V': return 'B'; case 'D': return 'H'; case 'H': return 'D';
        case 'W': return 'W'; case 'S': return 'S'; case 'Z': return 'Z';
        default: return 'N';
    }
}
__device__ __forceinline__ bool keep_site(int loc, int n) {   // extract_features.py:343-350, seq_len 21
    const int rl = n - 2 - loc;
    return loc >= 10 && loc < n - 10 && rl >= 10 && rl < n - 10;
}

struct ReadTable {            // device arrays, one entry per read
    const long long* offset;  // start of the read's bases in the concatenated byte arrays
    const int* length;
    const float* fn;          // subread passes, forward / reverse
    const float* rn;
};

template <typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    T s = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += scratch[w];
    return s;
}

// stats[read][8] = mean_fi, std_fi, mean_ri, std_ri, mean_fp, std_fp, mean_rp, std_rp ; nsites[read] = kept CG sites
__global__ __launch_bounds__(256) void extract_stats_kernel(ReadTable rt, const uint8_t* __restrict__ seq,
                                                            const uint8_t* __restrict__ fi, const uint8_t* __restrict__ ri,
                                                            const uint8_t* __restrict__ fp, const uint8_t* __restrict__ rp,
                                                            double* __restrict__ stats, int* __restrict__ nsites) {
    __shared__ double s_d[4];
    __shared__ long long s_l[4];
    const int r = blockIdx.x;
    const long long off = rt.offset[r];
    const int n = rt.length[r];
    const uint8_t* arr[4] = {fi + off, ri + off, fp + off, rp + off};
    int cnt = 0;
    for (int i = threadIdx.x; i + 1 < n; i += blockDim.x)
        cnt += (seq[off + i] == 'C' && seq[off + i + 1] == 'G' && keep_site(i, n)) ? 1 : 0;
    const long long total_sites = block_sum<long long>(cnt, s_l);
    if (threadIdx.x == 0) nsites[r] = (int)total_sites;
    for (int a = 0; a < 4; ++a) {
        long long isum = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) isum += codec_v1(arr[a][i]);
        const long long tot = block_sum<long long>(isum, s_l);
        const double mean = (double)tot / (double)n;          // exact integer sum, one rounding (== np.mean)
        double v = 0.0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const double d = (double)codec_v1(arr[a][i]) - mean;
            v += d * d;
        }
        const double var = block_sum<double>(v, s_d) / (double)n;
        if (threadIdx.x == 0) {
            stats[(size_t)r * 8 + 2 * a] = mean;
            stats[(size_t)r * 8 + 2 * a + 1] = sqrt(var);
        }
    }
}

__device__ __forceinline__ float zscore6(int code, double mean, double sd) {
    if (sd == 0.0) return 0.0f;                               // extract_features.py:195-196
    const double z = ((double)codec_v1(code) - mean) / sd;
    return (float)(rint(z * 1.0e6) / 1.0e6);                  // np.around(., 6), then the float32 cast of FloatTensor
}
__device__ __forceinline__ void split16f(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}
__device__ __forceinline__ uint32_t pack2h(_Float16 a, _Float16 b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 t = {a, b};
    return __builtin_bit_cast(uint32_t, t);
}

// x0: layer-0 input fragments [tile][t][hl][64 lanes] uint4 ; locs[first_site[r] + k] = position of the C in the read
__global__ __launch_bounds__(256) void extract_pack_kernel(ReadTable rt, const uint8_t* __restrict__ seq,
                                                           const uint8_t* __restrict__ fi, const uint8_t* __restrict__ ri,
                                                           const uint8_t* __restrict__ fp, const uint8_t* __restrict__ rp,
                                                           const double* __restrict__ stats, const int* __restrict__ first_site /* n_reads + 1 */,
                                                           const float* __restrict__ embed, uint4* __restrict__ x0,
                                                           int* __restrict__ locs, int n_sites_total, int row_base,
                                                           const unsigned long long* __restrict__ read_key /* n_reads or NULL */,
                                                           unsigned long long* __restrict__ site_key /* per site, when read_key */) {
    // grid = (reads, kPackParts): every part redoes the (cheap) ordered scan of the read's CG sites and writes the rows of
    // its own share of them, so that a chunk of a few long reads still spreads over the chip
    __shared__ int s_wtot[4];
    const int r = blockIdx.x;
    const long long off = rt.offset[r];
    const int n = rt.length[r];
    const double* st = stats + (size_t)r * 8;
    const float fn = rt.fn[r], rn = rt.rn[r];
    const int first = first_site[r];
    const int cap = first_site[r + 1] - first;     // rows reserved for this read (the caller's count when it supplied one)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ordered compaction of the kept sites, 256 candidate positions per round: wave ballots + four wave totals
    int base = 0;
    for (int start = 0; start < n; start += 256) {
        const int i = start + threadIdx.x;
        const bool hit = i + 1 < n && seq[off + i] == 'C' && seq[off + i + 1] == 'G' && keep_site(i, n);
        const unsigned long long mask = __ballot(hit);
        if (lane == 0) s_wtot[wave] = __popcll(mask);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int c = s_wtot[w];
            woff += w < wave ? c : 0;
            tot += c;
        }
        const int idx = base + woff + __popcll(mask & ((1ull << lane) - 1ull));
        if (hit && idx < cap) {                          // every part writes the same values
            locs[first + idx] = i;
            if (read_key) site_key[first + idx] = read_key[r];
        }
        base += tot;
        __syncthreads();
    }
    const int nk = min(base, cap);
    __threadfence_block();
    __syncthreads();
    const int items = nk * 2 * kSeqLen * 2;
    const int per_part = (items + (int)gridDim.y - 1) / (int)gridDim.y;
    const int w_begin = blockIdx.y * per_part, w_end = min(items, w_begin + per_part);
    // one thread per (site, strand, t, g): a 16-byte hi and a 16-byte lo fragment piece
    for (int w = w_begin + threadIdx.x; w < w_end; w += blockDim.x) {
        const int g = w & 1;
        const int t = (w >> 1) % kSeqLen;
        const int strand = ((w >> 1) / kSeqLen) & 1;
        const int k = (w >> 1) / (2 * kSeqLen);
        const int loc = locs[first + k];
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (strand == 0) {
            const int p = loc - 10 + t;
            if (g == 0) {
                const int code = base_code(seq[off + p]);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = embed[code * kEmbed + j];
            } else {
                v[0] = zscore6(fi[off + p], st[0], st[1]);
                v[1] = zscore6(fp[off + p], st[4], st[5]);
                v[2] = fn;
            }
        } else {
            if (g == 0) {
                const int code = base_code(comp_base(seq[off + loc + 11 - t]));   // reverse-complement k-mer
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = embed[code * kEmbed + j];
            } else {
                const int p = n - 12 - loc + t;                                    // ri / rp are not flipped
                v[0] = zscore6(ri[off + p], st[2], st[3]);
                v[1] = zscore6(rp[off + p], st[6], st[7]);
                v[2] = rn;
            }
        }
        const int row = row_base + first + k + strand * n_sites_total;
        const int tile = row >> 5, lane = (row & 31) + 32 * g;
        _Float16 hi[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) split16f(v[j], hi[j], lo[j]);
        uint4* dst = x0 + ((size_t)(tile * kSeqLen + t) * 2) * kFragU4 + lane;
        dst[0] = make_uint4(pack2h(hi[0], hi[1]), pack2h(hi[2], hi[3]), pack2h(hi[4], hi[5]), pack2h(hi[6], hi[7]));
        dst[kFragU4] = make_uint4(pack2h(lo[0], lo[1]), pack2h(lo[2], lo[3]), pack2h(lo[4], lo[5]), pack2h(lo[6], lo[7]));
    }
}

}  // namespace ccsm_extract
