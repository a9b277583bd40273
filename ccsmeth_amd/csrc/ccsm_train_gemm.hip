// libccsm_train: the training step's plain matrix products on the matrix cores, fp32 in / fp32 out, in the library's three-pass split-fp16
// arithmetic (every fp32 operand v = fp16 hi + fp16 lo; hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_f16, fp32 accumulation: 2^-22
// relative, what the inference kernels' SPLIT3 computes).  Replaces the rocBLAS sgemm / sgemm_strided_batched calls of rounds 1-3
// (reference: train_multigpu.py:283-312 runs these products inside torch autograd on cuBLAS / rocBLAS):
//   forward input projections    gi  = X W_ih^T          (21 M x in) (768 x in)^T          A row-major, B stored N x K      <false, true>
//   backward through them        dX  = dgi W_ih          (21 M x 768) (768 x in)           A row-major, B stored K x N      <false, false>
//   weight gradients             dW  = sum_t dg_t^T X_t  batched A^T B over blocks of rows  A stored K x M, B stored K x N    <true, false>
//   attention / fc1 products and the stepwise recurrent products of small batches: the same three forms.
// One kernel: C (M x N, ldc) = alpha op(A) op(B) + beta C, batched over blockIdx.z with element strides.
//   workgroup = 256 threads = 4 waves (2 x 2), tile 128 x 128, k-step 32; wave (wm, wn) owns 64 x 64 = 2 x 2 MFMA tiles (64 accumulator
//   registers).  A k-step's operands are read from global memory as fp32 (coalesced along whichever index is contiguous in memory),
//   split into hi / lo halfs ONCE per workgroup and written to LDS in MFMA fragment order (lane (n, g) of a fragment = 8 halfs of row n,
//   k = 8 g .. 8 g + 7: one conflict-free ds_read_b128 per operand); the next k-step's global loads are in flight while the current one
//   multiplies (two LDS buffers, one barrier per k-step).  Out-of-range rows / columns / k are zero-filled, so any shape goes (K = 11).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ccsm_train {

typedef _Float16 g_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 g_half4 __attribute__((ext_vector_type(4)));
typedef float g_f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    long long sA, sB, sC;          // batch strides in elements (blockIdx.z)
    float alpha, beta;
    const unsigned* amax;          // NULL, or the bits of max |A| over the whole operand (gemm_absmax_kernel): A is a GRADIENT - values far below
                                   // fp16's normal range (6e-5) - and is multiplied by the power of two that puts its maximum at 2^13 before the split
};

// max |x| over a (rows x cols) view of `batch` operands, as float bits (positive floats order like unsigned integers): atomicMax
// is exact and order-independent, so the scale it gives is the same every run
// (the view as it is STORED: `srows` rows of `scols` contiguous floats at row stride ld; one workgroup per row and batch entry at a time)
__global__ void gemm_absmax_kernel(const float* __restrict__ src, int ld, long long stride, int srows, int scols, int batch, unsigned* __restrict__ out) {
    float m = 0.f;
    const bool v4 = (ld & 3) == 0 && (stride & 3) == 0 && ((size_t)src & 15) == 0;
    for (long long rb = blockIdx.x; rb < (long long)srows * batch; rb += gridDim.x) {
        const float* row = src + (rb / srows) * stride + (rb % srows) * ld;
        if (v4) {
            const int n4 = scols >> 2;
            for (int c = threadIdx.x; c < n4; c += blockDim.x) {
                const float4 t = reinterpret_cast<const float4*>(row)[c];
                m = fmaxf(fmaxf(m, fmaxf(fabsf(t.x), fabsf(t.y))), fmaxf(fabsf(t.z), fabsf(t.w)));
            }
            for (int c = 4 * n4 + threadIdx.x; c < scols; c += blockDim.x) m = fmaxf(m, fabsf(row[c]));
        } else {
            for (int c = threadIdx.x; c < scols; c += blockDim.x) m = fmaxf(m, fabsf(row[c]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}
// the power of two that brings max |A| to [2^13, 2^14): as a float factor, and its inverse for alpha
__device__ __forceinline__ float gemm_scale_of(const unsigned* amax, float* inv) {
    if (amax == nullptr) { *inv = 1.f; return 1.f; }
    const int e = (int)((*amax >> 23) & 0xffu) - 127;               // floor(log2(max)); 0 bits (all-zero operand): e = -127 -> clamped
    int sh = 13 - e;
    sh = sh > 60 ? 60 : sh < -60 ? -60 : sh;
    *inv = __uint_as_float((unsigned)(127 - sh) << 23);
    return __uint_as_float((unsigned)(127 + sh) << 23);
}

constexpr int kGBM = 128, kGBN = 128, kGBK = 32;
constexpr int kGOperandHalfs = 4 * 2 * 64 * 8;          // one operand tile (128 x 32) in fragment order: [tile 4][kb 2][lane 64][8 halfs] = 8 KiB
constexpr int kGemmLds = 2 * 4 * kGOperandHalfs * 2;    // two buffers x (A hi, A lo, B hi, B lo) = 64 KiB

// One 128 x 32 operand tile: element (r, k) of the tile = src(r0 + r, k0 + k).  KCONTIG: k is the contiguous index in memory
// (src[(r0 + r) * ld + k]), else r is (src[k * ld + r]).  Each thread takes 4 items of 4 consecutive k of one row.
// Thread -> items of a tile: 4 items of 4 consecutive k of one row each.  KCONTIG (k contiguous in memory): lanes along k (8 lanes = 32 k
// = 128 B of one row; one 16-byte load per item when the view is 16-byte aligned).  Otherwise (the row index is contiguous): lanes along
// the rows, four 4-byte loads per item, each coalesced over 64 rows.  (A 4 x 4 block per thread with 16-byte loads along the rows was
// measured slower: its LDS stores land 64 bytes apart, a 16-way bank conflict.)
template <bool KCONTIG>
__device__ __forceinline__ void gemm_load(const float* __restrict__ src, int ld, int rows, int K, int r0, int k0, float (&v)[4][4], bool vec4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = threadIdx.x + 256 * i;                               // 1024 items
        const int r = KCONTIG ? q >> 3 : q & 127, kq = KCONTIG ? q & 7 : q >> 7;
        const int gr = r0 + r, gk = k0 + 4 * kq;
        if (KCONTIG && vec4 && gr < rows && gk + 3 < K) {                   // (ld, the base and k0 are multiples of 4 floats: one 16-byte load)
            const float4 t = *reinterpret_cast<const float4*>(src + (long long)gr * ld + gk);
            v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
            continue;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = gr < rows && gk + j < K;
            const long long off = KCONTIG ? (long long)gr * ld + gk + j : (long long)(gk + j) * ld + gr;
            v[i][j] = ok ? src[off] : 0.f;
        }
    }
}
template <bool KCONTIG>
__device__ __forceinline__ void gemm_store_lds(const float (&v)[4][4], _Float16* hi, _Float16* lo, float scale = 1.f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = threadIdx.x + 256 * i;
        const int r = KCONTIG ? q >> 3 : q & 127, kq = KCONTIG ? q & 7 : q >> 7;
        const int k = 4 * kq;
        const int idx = ((((r >> 5) * 2 + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + (r & 31)) * 8) + (k & 7);
        g_half4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = v[i][j] * scale;
            h[j] = (_Float16)x;
            l[j] = (_Float16)(x - (float)h[j]);
        }
        *reinterpret_cast<g_half4*>(hi + idx) = h;
        *reinterpret_cast<g_half4*>(lo + idx) = l;
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_s3_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char g_smem[];
    _Float16* sm = reinterpret_cast<_Float16*>(g_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * kGBM, n0 = blockIdx.x * kGBN;
    const float* A = a.A + (long long)blockIdx.z * a.sA;
    const float* B = a.B + (long long)blockIdx.z * a.sB;
    float* C = a.C + (long long)blockIdx.z * a.sC;
    g_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto buf = [&](int b, int which) -> _Float16* { return sm + (b * 4 + which) * kGOperandHalfs; };     // which: 0 A hi, 1 A lo, 2 B hi, 3 B lo
    float va[4][4], vb[4][4];
    float inv_scale;
    const float scale_a = gemm_scale_of(a.amax, &inv_scale);
    const bool va4 = (a.lda & 3) == 0 && (a.sA & 3) == 0 && ((size_t)a.A & 15) == 0;      // 16-byte loads along the contiguous index
    const bool vb4 = (a.ldb & 3) == 0 && (a.sB & 3) == 0 && ((size_t)a.B & 15) == 0;
    gemm_load<!TA>(A, a.lda, a.M, a.K, m0, 0, va, va4);
    gemm_load<TB>(B, a.ldb, a.N, a.K, n0, 0, vb, vb4);
    gemm_store_lds<!TA>(va, buf(0, 0), buf(0, 1), scale_a);
    gemm_store_lds<TB>(vb, buf(0, 2), buf(0, 3));
    __syncthreads();
    const int nk = (a.K + kGBK - 1) / kGBK;
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) {                                          // the next k-step's operands: in flight while this one multiplies
            gemm_load<!TA>(A, a.lda, a.M, a.K, m0, (ks + 1) * kGBK, va, va4);
            gemm_load<TB>(B, a.ldb, a.N, a.K, n0, (ks + 1) * kGBK, vb, vb4);
        }
        const _Float16 *ah = buf(cur, 0), *al = buf(cur, 1), *bh = buf(cur, 2), *bl = buf(cur, 3);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            g_half8 fah[2], fal[2], fbh[2], fbl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ia = (((2 * wm + i) * 2 + kb) * 64 + lane) * 8, ib = (((2 * wn + i) * 2 + kb) * 64 + lane) * 8;
                fah[i] = *reinterpret_cast<const g_half8*>(ah + ia);
                fal[i] = *reinterpret_cast<const g_half8*>(al + ia);
                fbh[i] = *reinterpret_cast<const g_half8*>(bh + ib);
                fbl[i] = *reinterpret_cast<const g_half8*>(bl + ib);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
        }
        if (ks + 1 < nk) {
            gemm_store_lds<!TA>(va, buf(cur ^ 1, 0), buf(cur ^ 1, 1), scale_a);
            gemm_store_lds<TB>(vb, buf(cur ^ 1, 2), buf(cur ^ 1, 3));
        }
        __syncthreads();                                            // the other buffer is written, this one is read by everybody
    }
    // C layout of a 32 x 32 tile: lane (n, hh) holds rows 8 q + 4 hh + e (accumulator 4 q + e) of column n
    const int n = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wn + 32 * j + n;
            if (col >= a.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 64 * wm + 32 * i + 8 * (r >> 2) + 4 * hh + (r & 3);
                if (row < a.M) {
                    float* p = C + (long long)row * a.ldc + col;
                    const float v = a.alpha * inv_scale * acc[i][j][r];
                    *p = a.beta != 0.f ? v + a.beta * *p : v;
                }
            }
        }
}

// max |x| of a stored (rows x cols, ld) view into *slot (one pass; several products on the same gradient tensor share it)
inline hipError_t gemm_scan_amax(hipStream_t st, const float* x, int ld, int rows, int cols, unsigned* slot) {
    hipError_t e = hipMemsetAsync(slot, 0, sizeof(unsigned), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gemm_absmax_kernel, dim3((unsigned)(rows < 2048 ? rows : 2048)), dim3(256), 0, st, x, ld, 0, rows, cols, 1, slot);
    return hipGetLastError();
}
// launch: C (M x N) = alpha op(A) op(B) + beta C over `batch` problems.  amax: NULL for operands of ordinary magnitude (activations,
// weights); a device word when A is a gradient: the kernel scales A by the power of two that word asks for.  scan: fill the word here
// with max |A| first; false: it already holds a bound of max |A| (gemm_scan_amax over the tensor A is a view of)
inline hipError_t gemm_s3(hipStream_t st, bool tA, bool tB, int M, int N, int K, float alpha, const float* A, int lda, long long sA, const float* B, int ldb,
                          long long sB, float beta, float* C, int ldc, long long sC, int batch, unsigned* amax = nullptr, bool scan = true) {
    if (amax != nullptr && scan) {
        hipError_t e = hipMemsetAsync(amax, 0, sizeof(unsigned), st);
        if (e != hipSuccess) return e;
        const int srows = tA ? K : M, scols = tA ? M : K;
        const long long rb = (long long)srows * batch;
        hipLaunchKernelGGL(gemm_absmax_kernel, dim3((unsigned)(rb < 2048 ? rb : 2048)), dim3(256), 0, st, A, lda, sA, srows, scols, batch, amax);
    }
    GemmArgs a{A, B, C, M, N, K, lda, ldb, ldc, sA, sB, sC, alpha, beta, amax};
    const dim3 grid((N + kGBN - 1) / kGBN, (M + kGBM - 1) / kGBM, batch);
    if (!tA && tB) hipLaunchKernelGGL((gemm_s3_kernel<false, true>), grid, dim3(256), kGemmLds, st, a);
    else if (!tA && !tB) hipLaunchKernelGGL((gemm_s3_kernel<false, false>), grid, dim3(256), kGemmLds, st, a);
    else if (tA && !tB) hipLaunchKernelGGL((gemm_s3_kernel<true, false>), grid, dim3(256), kGemmLds, st, a);
    else hipLaunchKernelGGL((gemm_s3_kernel<true, true>), grid, dim3(256), kGemmLds, st, a);
    return hipGetLastError();
}
inline hipError_t gemm_s3_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_s3_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_s3_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_s3_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_s3_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds);
    return e;
}

// sum of squares of x[0 .. n): per-block partials in a fixed order, then one block adds them in a fixed order (deterministic)
__global__ void sumsq_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
    __shared__ float s[256];
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) acc += x[i] * x[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}
__global__ void sumsq_final_kernel(const float* __restrict__ part, int parts, float* __restrict__ out) {
    __shared__ double s[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < parts; i += 256) acc += (double)part[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)sqrt(s[0]);
}

}  // namespace ccsm_train
