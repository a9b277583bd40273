/* CPU oracle (plain C restatement) of the ccsmeth attbigru2s forward — TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built from this file
 * (oracle/_build/liboracle.so); nothing under ccsmeth_amd/ links, loads or calls it.
 *
 * Parity status: PINNED — checked in tests/test_oracle_c.py against tests/golden/forward_golden.npz, which holds
 * outputs of the reference itself (tests/golden/make_golden.py).
 *
 * Follows (paths relative to /root/reference/):
 *   ccsmeth/models.py:89-150        ModelAttRNN.forward                 -> oracle_forward()
 *   ccsmeth/models.py:91-106        input concat: embed(kmer.int()), ipd, pw, npass -> build_input()
 *   torch.nn.GRU (ATen gru_cell; reference pins torch<=2.1.0, requirements.txt:4; call sites models.py:54-55,
 *   125-130): r = s(gi_r+gh_r), z = s(gi_z+gh_z), n = tanh(gi_n + r*gh_n), h' = (h-n)*z+n   -> gru_dir()
 *   ccsmeth/utils/attention.py:48-70   e = va.tanh(Wa q + Ua k), softmax over L, weighted sum -> attention()
 * fp32 storage and arithmetic (like the reference's CPU PyTorch path), 8-wide partial sums in the dot products.
 * Sites are independent; OpenMP parallelises over blocks of sites.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define L 21
#define H 256
#define NL 3
#define E 8
#define F0 11
#define BS 8 /* sites per block */

typedef struct {
    const float* embed;            /* (5,8) */
    const float* w_ih[NL][2];      /* (768, 11|512) */
    const float* w_hh[NL][2];      /* (768, 256) */
    const float* b_ih[NL][2];
    const float* b_hh[NL][2];
    const float* wa;               /* (256,512) */
    const float* ua;               /* (256,512) */
    const float* va;               /* (256) */
    const float* fcw;              /* (2,1024) */
    const float* fcb;              /* (2) */
} oracle_weights;

/* out[m][n] = sum_k a[m][k] * w[n][k] (+ bias[n]);  a: M x K (lda), w: N x K row-major.
 * 4 (rows of a) x 2 (rows of w) register block with 8-wide partial sums; the AVX2+FMA clone is used when the CPU has it. */
typedef float v8f __attribute__((vector_size(32), aligned(4), may_alias));

static inline float hsum8(v8f v) { return ((v[0] + v[4]) + (v[1] + v[5])) + ((v[2] + v[6]) + (v[3] + v[7])); }

#define GEMM_BODY                                                                                                      \
    const int K8 = K & ~7;                                                                                             \
    int n = 0;                                                                                                         \
    for (; n + 2 <= N; n += 2) {                                                                                       \
        const float *w0 = w + (size_t)n * K, *w1 = w0 + K;                                                             \
        const float b0 = bias ? bias[n] : 0.0f, b1 = bias ? bias[n + 1] : 0.0f;                                        \
        int m = 0;                                                                                                     \
        for (; m + 4 <= M; m += 4) {                                                                                   \
            const float* a0 = a + (size_t)m * lda;                                                                     \
            const float *a1 = a0 + lda, *a2 = a1 + lda, *a3 = a2 + lda;                                                \
            v8f c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0}, c20 = {0}, c21 = {0}, c30 = {0}, c31 = {0};                \
            for (int k = 0; k < K8; k += 8) {                                                                          \
                const v8f x0 = *(const v8f*)(w0 + k), x1 = *(const v8f*)(w1 + k);                                      \
                const v8f y0 = *(const v8f*)(a0 + k), y1 = *(const v8f*)(a1 + k);                                      \
                const v8f y2 = *(const v8f*)(a2 + k), y3 = *(const v8f*)(a3 + k);                                      \
                c00 += y0 * x0; c01 += y0 * x1; c10 += y1 * x0; c11 += y1 * x1;                                        \
                c20 += y2 * x0; c21 += y2 * x1; c30 += y3 * x0; c31 += y3 * x1;                                        \
            }                                                                                                          \
            float r[4][2] = {{hsum8(c00), hsum8(c01)}, {hsum8(c10), hsum8(c11)}, {hsum8(c20), hsum8(c21)},             \
                             {hsum8(c30), hsum8(c31)}};                                                                \
            const float* ar[4] = {a0, a1, a2, a3};                                                                     \
            for (int i = 0; i < 4; ++i) {                                                                              \
                for (int k = K8; k < K; ++k) { r[i][0] += ar[i][k] * w0[k]; r[i][1] += ar[i][k] * w1[k]; }             \
                out[(size_t)(m + i) * ldo + n] = r[i][0] + b0;                                                         \
                out[(size_t)(m + i) * ldo + n + 1] = r[i][1] + b1;                                                     \
            }                                                                                                          \
        }                                                                                                              \
        for (; m < M; ++m) {                                                                                           \
            const float* a0 = a + (size_t)m * lda;                                                                     \
            v8f c0 = {0}, c1 = {0};                                                                                    \
            for (int k = 0; k < K8; k += 8) {                                                                          \
                const v8f y = *(const v8f*)(a0 + k);                                                                   \
                c0 += y * *(const v8f*)(w0 + k);                                                                       \
                c1 += y * *(const v8f*)(w1 + k);                                                                       \
            }                                                                                                          \
            float r0 = hsum8(c0), r1 = hsum8(c1);                                                                      \
            for (int k = K8; k < K; ++k) { r0 += a0[k] * w0[k]; r1 += a0[k] * w1[k]; }                                 \
            out[(size_t)m * ldo + n] = r0 + b0;                                                                        \
            out[(size_t)m * ldo + n + 1] = r1 + b1;                                                                    \
        }                                                                                                              \
    }                                                                                                                  \
    for (; n < N; ++n) { /* odd N tail (unused for this model: N is 768, 256 or 2) */                                  \
        const float* w0 = w + (size_t)n * K;                                                                           \
        for (int m = 0; m < M; ++m) {                                                                                  \
            const float* a0 = a + (size_t)m * lda;                                                                     \
            v8f c0 = {0};                                                                                              \
            for (int k = 0; k < K8; k += 8) c0 += *(const v8f*)(a0 + k) * *(const v8f*)(w0 + k);                        \
            float r0 = hsum8(c0);                                                                                      \
            for (int k = K8; k < K; ++k) r0 += a0[k] * w0[k];                                                          \
            out[(size_t)m * ldo + n] = r0 + (bias ? bias[n] : 0.0f);                                                   \
        }                                                                                                              \
    }

__attribute__((target("avx2,fma"))) static void gemm_nt_avx2(const float* a, int lda, const float* w, const float* bias,
                                                             float* out, int ldo, int M, int N, int K) {
    GEMM_BODY
}
static void gemm_nt_generic(const float* a, int lda, const float* w, const float* bias, float* out, int ldo, int M, int N,
                            int K) {
    GEMM_BODY
}
static void gemm_nt(const float* a, int lda, const float* w, const float* bias, float* out, int ldo, int M, int N, int K) {
    static int has_avx2 = -1;
    if (has_avx2 < 0) has_avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    if (has_avx2) gemm_nt_avx2(a, lda, w, bias, out, ldo, M, N, K);
    else gemm_nt_generic(a, lda, w, bias, out, ldo, M, N, K);
}

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* one direction of one layer for B sites.  x: (B, L, K); h0: B rows of H (stride h0_stride); out: (B, L, 2H) at column offset */
static void gru_dir(const float* x, int K, const float* h0, size_t h0_stride, const float* w_ih, const float* w_hh,
                    const float* b_ih, const float* b_hh, int reverse, float* out, int col0, int B, float* gi, float* gh,
                    float* h) {
    gemm_nt(x, K, w_ih, b_ih, gi, 3 * H, B * L, 3 * H, K);
    for (int b = 0; b < B; ++b) memcpy(h + (size_t)b * H, h0 + (size_t)b * h0_stride, sizeof(float) * H);
    for (int s = 0; s < L; ++s) {
        const int t = reverse ? L - 1 - s : s;
        gemm_nt(h, H, w_hh, b_hh, gh, 3 * H, B, 3 * H, H);
        for (int b = 0; b < B; ++b) {
            const float* gib = gi + ((size_t)b * L + t) * 3 * H;
            const float* ghb = gh + (size_t)b * 3 * H;
            float* hb = h + (size_t)b * H;
            float* ob = out + ((size_t)b * L + t) * 2 * H + col0;
            for (int u = 0; u < H; ++u) {
                const float r = sigmoidf_(gib[u] + ghb[u]);
                const float z = sigmoidf_(gib[H + u] + ghb[H + u]);
                const float n = tanhf(gib[2 * H + u] + r * ghb[2 * H + u]);
                const float hn = (hb[u] - n) * z + n;
                hb[u] = hn;
                ob[u] = hn;
            }
        }
    }
}

typedef struct {
    float *x0, *bufa, *bufb, *gi, *gh, *h, *q, *k, *ctx;
} scratch;

static int scratch_alloc(scratch* s) {
    s->x0 = malloc(sizeof(float) * BS * L * F0);
    s->bufa = malloc(sizeof(float) * BS * L * 2 * H);
    s->bufb = malloc(sizeof(float) * BS * L * 2 * H);
    s->gi = malloc(sizeof(float) * BS * L * 3 * H);
    s->gh = malloc(sizeof(float) * BS * 3 * H);
    s->h = malloc(sizeof(float) * BS * H);
    s->q = malloc(sizeof(float) * BS * H);
    s->k = malloc(sizeof(float) * BS * L * H);
    s->ctx = malloc(sizeof(float) * BS * 4 * H);
    return s->x0 && s->bufa && s->bufb && s->gi && s->gh && s->h && s->q && s->k && s->ctx;
}
static void scratch_free(scratch* s) {
    free(s->x0); free(s->bufa); free(s->bufb); free(s->gi); free(s->gh); free(s->h); free(s->q); free(s->k); free(s->ctx);
}

/* models.py:91-106 */
static void build_input(const oracle_weights* w, const uint8_t* kmer, const float* ipd, const float* pw, const float* npass,
                        int site0, int B, float* x0) {
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < L; ++t) {
            const size_t e = (size_t)(site0 + b) * L + t;
            float* xr = x0 + ((size_t)b * L + t) * F0;
            int code = kmer[e];
            if (code > 4) code = 4;
            for (int j = 0; j < E; ++j) xr[j] = w->embed[code * E + j];
            xr[8] = ipd[e];
            xr[9] = pw[e];
            xr[10] = npass[site0 + b];
        }
}

/* one strand of B sites -> ctx (B, 2H) written at ctx[b*4H + strand*2H] */
static void strand_forward(const oracle_weights* w, const uint8_t* kmer, const float* ipd, const float* pw, const float* npass,
                           const float* h0 /* (6,N,256) */, int n_sites, int site0, int B, int strand, scratch* s) {
    build_input(w, kmer, ipd, pw, npass, site0, B, s->x0);
    const float* in = s->x0;
    int K = F0;
    float* bufs[2] = {s->bufa, s->bufb};
    for (int l = 0; l < NL; ++l) {
        float* out = bufs[l & 1];
        for (int d = 0; d < 2; ++d)
            gru_dir(in, K, h0 + ((size_t)(2 * l + d) * n_sites + site0) * H, H, w->w_ih[l][d], w->w_hh[l][d], w->b_ih[l][d],
                    w->b_hh[l][d], d, out, d * H, B, s->gi, s->gh, s->h);
        in = out;
        K = 2 * H;
    }
    const float* enc = in; /* (B, L, 2H) */
    /* h_n of the last layer = [fwd final = enc[:, L-1, :H] | bwd final = enc[:, 0, H:]]  (models.py:135-137) */
    for (int b = 0; b < B; ++b) {
        float hn[2 * H];
        memcpy(hn, enc + ((size_t)b * L + (L - 1)) * 2 * H, sizeof(float) * H);
        memcpy(hn + H, enc + ((size_t)b * L) * 2 * H + H, sizeof(float) * H);
        gemm_nt(hn, 2 * H, w->wa, NULL, s->q + (size_t)b * H, H, 1, H, 2 * H);
    }
    gemm_nt(enc, 2 * H, w->ua, NULL, s->k, H, B * L, H, 2 * H);
    for (int b = 0; b < B; ++b) {
        float e[L], m = -3.0e38f, den = 0.f;
        for (int t = 0; t < L; ++t) {
            const float* kr = s->k + ((size_t)b * L + t) * H;
            const float* qr = s->q + (size_t)b * H;
            float acc = 0.f;
            for (int u = 0; u < H; ++u) acc += w->va[u] * tanhf(qr[u] + kr[u]);
            e[t] = acc;
            if (acc > m) m = acc;
        }
        for (int t = 0; t < L; ++t) { e[t] = expf(e[t] - m); den += e[t]; }
        float* c = s->ctx + (size_t)b * 4 * H + strand * 2 * H;
        for (int u = 0; u < 2 * H; ++u) c[u] = 0.f;
        for (int t = 0; t < L; ++t) {
            const float a = e[t] / den;
            const float* er = enc + ((size_t)b * L + t) * 2 * H;
            for (int u = 0; u < 2 * H; ++u) c[u] += a * er[u];
        }
    }
}

/* Returns 0 on success.  kmer: (N,21) u8; ipd/pw: (N,21) f32; npass: (N) f32; h0_x: (6,N,256) f32. threads<=0: OpenMP default. */
int oracle_forward(const oracle_weights* w, int n_sites, const uint8_t* kmer1, const float* ipd1, const float* pw1,
                   const float* npass1, const uint8_t* kmer2, const float* ipd2, const float* pw2, const float* npass2,
                   const float* h0_1, const float* h0_2, float* logits, float* probs, int threads) {
    int err = 0;
    const int nblk = (n_sites + BS - 1) / BS;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        scratch s;
        const int ok = scratch_alloc(&s);
        if (!ok) {
#pragma omp atomic write
            err = 1;
        }
#pragma omp for schedule(dynamic, 1)
        for (int blk = 0; blk < nblk; ++blk) {
            if (!ok) continue;
            const int site0 = blk * BS;
            const int B = n_sites - site0 < BS ? n_sites - site0 : BS;
            strand_forward(w, kmer1, ipd1, pw1, npass1, h0_1, n_sites, site0, B, 0, &s);
            strand_forward(w, kmer2, ipd2, pw2, npass2, h0_2, n_sites, site0, B, 1, &s);
            for (int b = 0; b < B; ++b) {
                float lg[2];
                gemm_nt(s.ctx + (size_t)b * 4 * H, 4 * H, w->fcw, w->fcb, lg, 2, 1, 2, 4 * H);   /* models.py:145-148 */
                const float m = lg[0] > lg[1] ? lg[0] : lg[1];
                const float e0 = expf(lg[0] - m), e1 = expf(lg[1] - m);
                logits[(size_t)(site0 + b) * 2] = lg[0];
                logits[(size_t)(site0 + b) * 2 + 1] = lg[1];
                probs[(size_t)(site0 + b) * 2] = e0 / (e0 + e1);                                  /* models.py:150 */
                probs[(size_t)(site0 + b) * 2 + 1] = e1 / (e0 + e1);
            }
        }
        scratch_free(&s);
    }
    return err;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
