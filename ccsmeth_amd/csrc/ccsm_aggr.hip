// libccsm, aggregate mode (BASELINE config 5): per-site methylation frequency from pile-up histograms.
//
// Reference behaviour reproduced (paths into /root/reference/ccsmeth/):
//   models.py:625-694                 AggrAttRNN: x = cat(histogram(20), |pos - centre|) over an 11-site window,
//                                     1-layer BiGRU(H=32), attention (utils/attention.py:48-70), fc1 (64 -> 1), no softmax
//   call_mods_freq_bam.py:265-305     _cal_modfreq_in_aggregate_mode: zero-padded windows, pad positions first-1000 /
//                                     last+1000, batches of 1024, h0 = torch.randn(2, B, 32) from the stream seeded per
//                                     region (call_mods_freq_bam.py:313)
// 275 KFLOP and 88 B per site: far below MFMA tile sizes (K = 21 / 32 / 64), so this is an fp32 VALU kernel:
// one wavefront per site, lane = direction * 32 + hidden unit, GRU weights resident in VGPRs (168 per lane) across the
// sites a wave walks, windows built on the fly from the (M,20) histogram table (never materialising the reference's
// (M,11,21) tensor), hidden state and the 11x64 layer output in LDS, wave-level reductions for the attention scores.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ccsm_aggr {

constexpr int L = 11, H = 32, NB = 20, F = 21, WAVES = 4;

__device__ __forceinline__ float sigmoid_a(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_a(float x) { return 1.0f - 2.0f / (__expf(2.0f * x) + 1.0f); }

struct Weights {
    const float* w_ih;   // [2][96][21]
    const float* w_hh;   // [2][96][32]
    const float* b_ih;   // [2][96]
    const float* b_hh;   // [2][96]
    const float* wa_t;   // [64][32]  Wa transposed: wa_t[k][a] = Wa[a][k]
    const float* ua_t;   // [64][32]
    const float* va;     // [32]
    const float* fcw;    // [64]
    const float* fcb;    // [1]
};

// pos: (M) int64 sorted reference positions; hist: (M,20) fp32 normalised histograms; normals: the seeded randn stream;
// stream_pos: index of the first value this call consumes; out: (M) fp32 raw fc1 output.  only_close (--only_close,
// call_mods_freq_bam.py:285-290): the position feature is 1 where the window's site lies exactly 2 bases after its predecessor
// (the padded sequence first-1000, ..., positions, ..., last+1000), else 0, instead of the distance to the centre site.
__global__ __launch_bounds__(256, 2) void aggr_kernel(Weights w, const long long* __restrict__ pos, const float* __restrict__ hist,
                                                       const float* __restrict__ normals, long long stream_pos, int m,
                                                       float* __restrict__ out, int only_close) {
    __shared__ float s_wa[64 * H], s_ua[64 * H];
    __shared__ float s_h[WAVES][2 * H];         // current hidden state, [dir*32 + unit]
    __shared__ float s_o[WAVES][L][2 * H];      // layer output
    __shared__ float s_e[WAVES][16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dir = lane >> 5, unit = lane & 31;
    for (int i = threadIdx.x; i < 64 * H; i += blockDim.x) { s_wa[i] = w.wa_t[i]; s_ua[i] = w.ua_t[i]; }
    __syncthreads();

    // this lane's rows of the GRU weights (gate g: row g*32 + unit)
    float wih[3][F], whh[3][H], bi[3], bh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int row = g * H + unit;
#pragma unroll
        for (int k = 0; k < F; ++k) wih[g][k] = w.w_ih[(dir * 96 + row) * F + k];
#pragma unroll
        for (int k = 0; k < H; ++k) whh[g][k] = w.w_hh[(dir * 96 + row) * H + k];
        bi[g] = w.b_ih[dir * 96 + row];
        bh[g] = w.b_hh[dir * 96 + row];
    }
    const float va = w.va[unit];
    const float fcw = w.fcw[lane];
    const float fcb = w.fcb[0];
    const long long pos_lo = pos[0] - 1000, pos_hi = pos[m - 1] + 1000;

    const int gwave = blockIdx.x * WAVES + wave;
    const int nwaves = gridDim.x * WAVES;
    for (int site = gwave; site < m; site += nwaves) {
        // h0 = randn(2, B, 32)[dir][site - batch0][unit] with B = size of this site's reference batch of 1024
        const int batch0 = (site >> 10) << 10;
        const int bsz = min(1024, m - batch0);
        float h = normals[stream_pos + (long long)batch0 * 64 + ((long long)dir * bsz + (site - batch0)) * H + unit];
        const long long pc = pos[site];
        for (int s = 0; s < L; ++s) {
            const int t = dir ? L - 1 - s : s;
            const int nb = site + t - L / 2;
            const bool in = nb >= 0 && nb < m;
            float x[F];
            if (in) {
                const float4* hp = reinterpret_cast<const float4*>(hist + (size_t)nb * NB);
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const float4 v = hp[q];
                    x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < NB; ++k) x[k] = 0.f;
            }
            const long long pn = in ? pos[nb] : (nb < 0 ? pos_lo : pos_hi);
            if (only_close) {
                const int pb = nb - 1;
                const long long pp = (pb >= 0 && pb < m) ? pos[pb] : (pb < 0 ? pos_lo : pos_hi);
                x[NB] = pn - pp == 2 ? 1.f : 0.f;
            } else {
                const long long d = pn - pc;
                x[NB] = (float)(d < 0 ? -d : d);
            }
            s_h[wave][lane] = h;                       // wave-private LDS: in-order within the wave, no barrier needed
            float gi[3], gh[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float a = bi[g];
#pragma unroll
                for (int k = 0; k < F; ++k) a = fmaf(wih[g][k], x[k], a);
                gi[g] = a;
                gh[g] = bh[g];
            }
            const float4* hv = reinterpret_cast<const float4*>(&s_h[wave][dir * H]);
#pragma unroll
            for (int q = 0; q < H / 4; ++q) {
                const float4 v = hv[q];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    gh[g] = fmaf(whh[g][4 * q], v.x, gh[g]);
                    gh[g] = fmaf(whh[g][4 * q + 1], v.y, gh[g]);
                    gh[g] = fmaf(whh[g][4 * q + 2], v.z, gh[g]);
                    gh[g] = fmaf(whh[g][4 * q + 3], v.w, gh[g]);
                }
            }
            const float r = sigmoid_a(gi[0] + gh[0]);
            const float z = sigmoid_a(gi[1] + gh[1]);
            const float n = tanh_a(gi[2] + r * gh[2]);
            h = (h - n) * z + n;
            s_o[wave][t][lane] = h;
        }
        s_h[wave][lane] = h;                            // final states = attention query [fwd | bwd]
        // q[a] = sum_k Wa[a][k] hn[k]: this half sums k in [32*dir, 32*dir+32), then the halves are added
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < H; ++k) q = fmaf(s_wa[(dir * H + k) * H + unit], s_h[wave][dir * H + k], q);
        q += __shfl_xor(q, 32);
        // e[t] = sum_a va[a] tanh(q[a] + Ua[a] . out[t]); half 0 takes t = 0..5, half 1 takes t = 6..10
        const int t0 = dir ? 6 : 0, nt = dir ? 5 : 6;
        for (int tt = 0; tt < nt; ++tt) {
            const int t = t0 + tt;
            float kq = q;
#pragma unroll 8
            for (int k = 0; k < 2 * H; ++k) kq = fmaf(s_ua[k * H + unit], s_o[wave][t][k], kq);
            float e = va * tanh_a(kq);
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) e += __shfl_xor(e, o);     // reduce over the 32 units of this half
            if (unit == 0) s_e[wave][t] = e;
        }
        float ev[L], mx = -3.0e38f;
#pragma unroll
        for (int t = 0; t < L; ++t) { ev[t] = s_e[wave][t]; mx = fmaxf(mx, ev[t]); }
        float den = 0.f;
#pragma unroll
        for (int t = 0; t < L; ++t) { ev[t] = __expf(ev[t] - mx); den += ev[t]; }
        float c = 0.f;
#pragma unroll
        for (int t = 0; t < L; ++t) c = fmaf(ev[t] / den, s_o[wave][t][lane], c);
        float y = fcw * c;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) y += __shfl_xor(y, o);
        if (lane == 0) out[site] = y + fcb;
    }
}

}  // namespace ccsm_aggr
