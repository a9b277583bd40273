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
 * fp32 storage and arithmetic (like the reference's CPU PyTorch path).  Sites are independent; OpenMP parallelises over blocks
 * of BS sites.  So that bench.py's cpu_baseline is a fair CPU number and not a strawman, the matrix products run as a blocked
 * GEMM (weights packed once per call into K x 2-vector panels, MR x 2-vector outer-product register blocks, AVX-512 / AVX2+FMA /
 * generic clones picked at run time — oracle_kernels.inc) and the gate non-linearities use a vectorised exp.
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
#define BS 48  /* sites per block: a multiple of every MR below */

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

typedef struct {
    int vw, mr;
    void (*gemm)(const float* a, int lda, const float* wp, const float* bias, float* out, int ldo, int M, int N, int K);
    void (*pack)(const float* w, int N, int K, float* wp);
    void (*gates)(const float* gi, const float* gh, float* h, float* o);
    float (*score)(const float* va, const float* q, const float* k);
} oracle_isa;

#pragma GCC push_options
#pragma GCC target("avx512f,avx512dq,avx512vl,avx2,fma")
#define VW 16
#define MR 12
#define SFX _avx512
#include "oracle_kernels.inc"
#undef VW
#undef MR
#undef SFX
#pragma GCC pop_options

#pragma GCC push_options
#pragma GCC target("avx2,fma")
#define VW 8
#define MR 6
#define SFX _avx2
#include "oracle_kernels.inc"
#undef VW
#undef MR
#undef SFX
#pragma GCC pop_options

#define VW 4
#define MR 4
#define SFX _generic
#include "oracle_kernels.inc"
#undef VW
#undef MR
#undef SFX

static const oracle_isa* pick_isa(void) {
    const char* force = getenv("ORACLE_ISA");      /* tests: "generic", "avx2", "avx512" */
    if (force && !strcmp(force, "generic")) return &isa_generic;
    if (force && !strcmp(force, "avx2")) return &isa_avx2;
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl")) return &isa_avx512;
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) return &isa_avx2;
    return &isa_generic;
}

/* weights packed for the chosen clone (shared by all threads, read-only) */
typedef struct {
    float *ih[NL][2], *hh[NL][2], *wa, *ua;
} packed_weights;

static void packed_free(packed_weights* p) {
    for (int l = 0; l < NL; ++l)
        for (int d = 0; d < 2; ++d) { free(p->ih[l][d]); free(p->hh[l][d]); }
    free(p->wa); free(p->ua);
}
static float* pack_one(const oracle_isa* isa, const float* w, int N, int K) {
    float* wp = aligned_alloc(64, sizeof(float) * (size_t)N * K);
    if (wp) isa->pack(w, N, K, wp);
    return wp;
}
static int packed_build(const oracle_isa* isa, const oracle_weights* w, packed_weights* p) {
    int ok = 1;
    memset(p, 0, sizeof(*p));
    for (int l = 0; l < NL; ++l)
        for (int d = 0; d < 2; ++d) {
            ok &= (p->ih[l][d] = pack_one(isa, w->w_ih[l][d], 3 * H, l ? 2 * H : F0)) != NULL;
            ok &= (p->hh[l][d] = pack_one(isa, w->w_hh[l][d], 3 * H, H)) != NULL;
        }
    ok &= (p->wa = pack_one(isa, w->wa, H, 2 * H)) != NULL;
    ok &= (p->ua = pack_one(isa, w->ua, H, 2 * H)) != NULL;
    return ok;
}

/* one direction of one layer for B sites.  x: (B, L, K); h0: B rows of H (stride h0_stride); out: (B, L, 2H) at column offset.
 * Scratch rows beyond B hold finite leftovers of earlier blocks: the GEMM computes whole register blocks, the gates only B rows. */
static void gru_dir(const oracle_isa* isa, const float* x, int K, const float* h0, size_t h0_stride, const float* wp_ih,
                    const float* wp_hh, const float* b_ih, const float* b_hh, int reverse, float* out, int col0, int B, float* gi,
                    float* gh, float* h) {
    for (int b = 0; b < B; ++b) memcpy(h + (size_t)b * H, h0 + (size_t)b * h0_stride, sizeof(float) * H);
    for (int s = 0; s < L; ++s) {
        const int t = reverse ? L - 1 - s : s;
        /* the input projection of THIS step only (rows b*L + t of x): the per-thread working set stays at the two layer buffers, which
         * is what lets all hardware threads of a many-core host run at once without streaming a (B*L, 3H) buffer through DRAM */
        isa->gemm(x + (size_t)t * K, L * K, wp_ih, b_ih, gi, 3 * H, B, 3 * H, K);
        isa->gemm(h, H, wp_hh, b_hh, gh, 3 * H, B, 3 * H, H);
        for (int b = 0; b < B; ++b)
            isa->gates(gi + (size_t)b * 3 * H, gh + (size_t)b * 3 * H, h + (size_t)b * H, out + ((size_t)b * L + t) * 2 * H + col0);
    }
}

typedef struct {
    float *x0, *bufa, *bufb, *gi, *gh, *h, *hn, *q, *k, *ctx;
} scratch;

static float* zalloc(size_t n) {
    float* p = aligned_alloc(64, (sizeof(float) * n + 63) / 64 * 64);
    if (p) memset(p, 0, sizeof(float) * n);
    return p;
}
static int scratch_alloc(scratch* s) {
    s->x0 = zalloc((size_t)BS * L * F0);
    s->bufa = zalloc((size_t)BS * L * 2 * H);
    s->bufb = zalloc((size_t)BS * L * 2 * H);
    s->gi = zalloc((size_t)BS * 3 * H);
    s->gh = zalloc((size_t)BS * 3 * H);
    s->h = zalloc((size_t)BS * H);
    s->hn = zalloc((size_t)BS * 2 * H);
    s->q = zalloc((size_t)BS * H);
    s->k = zalloc((size_t)BS * L * H);
    s->ctx = zalloc((size_t)BS * 4 * H);
    return s->x0 && s->bufa && s->bufb && s->gi && s->gh && s->h && s->hn && s->q && s->k && s->ctx;
}
static void scratch_free(scratch* s) {
    free(s->x0); free(s->bufa); free(s->bufb); free(s->gi); free(s->gh); free(s->h); free(s->hn); free(s->q); free(s->k); free(s->ctx);
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
static void strand_forward(const oracle_isa* isa, const oracle_weights* w, const packed_weights* pw_, const uint8_t* kmer,
                           const float* ipd, const float* pw, const float* npass, const float* h0 /* (6,N,256) */, int n_sites,
                           int site0, int B, int strand, scratch* s) {
    build_input(w, kmer, ipd, pw, npass, site0, B, s->x0);
    const float* in = s->x0;
    int K = F0;
    float* bufs[2] = {s->bufa, s->bufb};
    for (int l = 0; l < NL; ++l) {
        float* out = bufs[l & 1];
        for (int d = 0; d < 2; ++d)
            gru_dir(isa, in, K, h0 + ((size_t)(2 * l + d) * n_sites + site0) * H, H, pw_->ih[l][d], pw_->hh[l][d], w->b_ih[l][d],
                    w->b_hh[l][d], d, out, d * H, B, s->gi, s->gh, s->h);
        in = out;
        K = 2 * H;
    }
    const float* enc = in; /* (B, L, 2H) */
    /* h_n of the last layer = [fwd final = enc[:, L-1, :H] | bwd final = enc[:, 0, H:]]  (models.py:135-137) */
    for (int b = 0; b < B; ++b) {
        memcpy(s->hn + (size_t)b * 2 * H, enc + ((size_t)b * L + (L - 1)) * 2 * H, sizeof(float) * H);
        memcpy(s->hn + (size_t)b * 2 * H + H, enc + ((size_t)b * L) * 2 * H + H, sizeof(float) * H);
    }
    isa->gemm(s->hn, 2 * H, pw_->wa, NULL, s->q, H, B, H, 2 * H);
    isa->gemm(enc, 2 * H, pw_->ua, NULL, s->k, H, B * L, H, 2 * H);
    for (int b = 0; b < B; ++b) {
        float e[L], m = -3.0e38f, den = 0.f;
        for (int t = 0; t < L; ++t) {
            e[t] = isa->score(w->va, s->q + (size_t)b * H, s->k + ((size_t)b * L + t) * H);
            if (e[t] > m) m = e[t];
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
    const oracle_isa* isa = pick_isa();
    packed_weights pk;
    if (!packed_build(isa, w, &pk)) { packed_free(&pk); return 1; }
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
            strand_forward(isa, w, &pk, kmer1, ipd1, pw1, npass1, h0_1, n_sites, site0, B, 0, &s);
            strand_forward(isa, w, &pk, kmer2, ipd2, pw2, npass2, h0_2, n_sites, site0, B, 1, &s);
            for (int b = 0; b < B; ++b) {
                float lg[2] = {w->fcb[0], w->fcb[1]};                                             /* models.py:145-148 */
                for (int u = 0; u < 4 * H; ++u) {
                    lg[0] += s.ctx[(size_t)b * 4 * H + u] * w->fcw[u];
                    lg[1] += s.ctx[(size_t)b * 4 * H + u] * w->fcw[4 * H + u];
                }
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
    packed_free(&pk);
    return err;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* sites per block (callers that size a timing sample use a multiple of this times the thread count) and the clone in use */
int oracle_block_sites(void) { return BS; }
const char* oracle_isa_name(void) {
    const oracle_isa* isa = pick_isa();
    return isa == &isa_avx512 ? "avx512" : isa == &isa_avx2 ? "avx2+fma" : "generic";
}
