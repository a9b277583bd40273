/* A plain C caller of the drop-in boundary (include/ccsm.h): no Python, no torch types.
 *
 *   cabi_client --version                 load the library, print ccsm_version(), exercise the error path (no GPU needed)
 *   cabi_client <case.bin> [max_err]      forward one batch on device 0 and compare with the expected probabilities
 *
 * case.bin (little endian, written by tests/test_gpu_cabi_client.py): int32 n; then float32 arrays in this order:
 * the 30 parameter tensors in state_dict order; per strand kmer as uint8 (n*21), ipd, pw (n*21), npass (n);
 * h0 strand 1, h0 strand 2 (6*n*256 each); expected probs (n*2).
 * Build: gcc -O2 -Iinclude tests/cabi/cabi_client.c -Lccsmeth_amd/lib -lccsm -Wl,-rpath,<abs lib dir> -lm -o cabi_client */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ccsm.h"

static void* slurp(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read (%zu bytes)\n", bytes); exit(2); }
    return p;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s --version | case.bin [max_err]\n", argv[0]); return 2; }
    if (strcmp(argv[1], "--version") == 0) {
        ccsm_model* m = NULL;
        printf("%s\n", ccsm_version());
        if (ccsm_create(NULL, NULL, 0, &m) != CCSM_ERR_INVALID_ARG) { fprintf(stderr, "NULL config was accepted\n"); return 1; }
        printf("error path ok: %s\n", ccsm_last_error());
        return 0;
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    const double max_err = argc > 2 ? atof(argv[2]) : 1e-4;
    int32_t n = 0;
    if (fread(&n, 4, 1, f) != 1 || n <= 0) { fprintf(stderr, "bad header\n"); return 2; }
    const size_t H = 256, G = 768, T = 21;
    ccsm_weights w;
    memset(&w, 0, sizeof w);
    w.embed_weight = slurp(f, 4 * 5 * 8);
    for (int l = 0; l < CCSM_LAYERS; ++l)
        for (int d = 0; d < 2; ++d) {
            w.weight_ih[l][d] = slurp(f, 4 * G * (l == 0 ? 11 : 2 * H));
            w.weight_hh[l][d] = slurp(f, 4 * G * H);
            w.bias_ih[l][d] = slurp(f, 4 * G);
            w.bias_hh[l][d] = slurp(f, 4 * G);
        }
    w.att_wa = slurp(f, 4 * H * 2 * H);
    w.att_ua = slurp(f, 4 * H * 2 * H);
    w.att_va = slurp(f, 4 * H);
    w.fc1_weight = slurp(f, 4 * 2 * 4 * H);
    w.fc1_bias = slurp(f, 4 * 2);
    ccsm_batch b;
    memset(&b, 0, sizeof b);
    for (int s = 0; s < 2; ++s) {
        b.strand[s].kmer = slurp(f, (size_t)n * T);
        b.strand[s].ipd = slurp(f, 4 * (size_t)n * T);
        b.strand[s].pw = slurp(f, 4 * (size_t)n * T);
        b.strand[s].npass = slurp(f, 4 * (size_t)n);
    }
    ccsm_h0 h0;
    memset(&h0, 0, sizeof h0);
    h0.mode = CCSM_H0_EXPLICIT;
    h0.h0[0] = slurp(f, 4 * 6 * (size_t)n * H);
    h0.h0[1] = slurp(f, 4 * 6 * (size_t)n * H);
    float* want = slurp(f, 4 * (size_t)n * 2);
    fclose(f);

    ccsm_config cfg = {21, 3, 2, 256, 1, 0, 0, 0, "attbigru2s", 0};
    ccsm_model* m = NULL;
    ccsm_workspace* ws = NULL;
    if (ccsm_create(&cfg, &w, 0, &m) != CCSM_OK || ccsm_workspace_create(m, n, &ws) != CCSM_OK) {
        fprintf(stderr, "create failed: %s\n", ccsm_last_error());
        return 1;
    }
    float* logits = malloc(4 * (size_t)n * 2);
    float* probs = malloc(4 * (size_t)n * 2);
    if (ccsm_forward_host(m, ws, n, &b, &h0, logits, probs, NULL) != CCSM_OK) {
        fprintf(stderr, "forward failed: %s\n", ccsm_last_error());
        return 1;
    }
    double err = 0.0;
    for (size_t i = 0; i < (size_t)n * 2; ++i) err = fmax(err, fabs((double)probs[i] - (double)want[i]));
    /* the same batch again through the pipelined entry points */
    if (ccsm_submit_host(m, ws, n, &b, &h0, NULL) != CCSM_OK || ccsm_wait_host(ws, logits, probs) != CCSM_OK) {
        fprintf(stderr, "submit/wait failed: %s\n", ccsm_last_error());
        return 1;
    }
    double err2 = 0.0;
    for (size_t i = 0; i < (size_t)n * 2; ++i) err2 = fmax(err2, fabs((double)probs[i] - (double)want[i]));
    /* capacity error is reported, not a crash */
    const ccsm_status cap = ccsm_forward_host(m, ws, n + 1, &b, &h0, logits, probs, NULL);
    printf("n=%d max|dprob| forward_host %.3e submit/wait %.3e capacity_status %d precision %d\n", n, err, err2, (int)cap, ccsm_model_precision(m));
    ccsm_workspace_destroy(ws);
    ccsm_destroy(m);
    return (err <= max_err && err2 <= max_err && cap == CCSM_ERR_CAPACITY) ? 0 : 1;
}
