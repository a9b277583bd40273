/* libccsm — C-ABI of the MI355X-native (gfx950) ccsmeth `call_mods` attbigru2s hot path.
 *
 * The reference (PengNi/ccsmeth v0.5.0, pure Python) has no FFI; its seam for this path is the Python call
 *   model(*16 tensors)                      ccsmeth/call_modifications.py:201-208   (_call_mods2s)
 * plus the constructor / load_state_dict contract at ccsmeth/call_modifications.py:315-358 and the model
 * definition ccsmeth/models.py:17-150 (ModelAttRNN, model_type="attbigru2s"), ccsmeth/utils/attention.py:30-70.
 * Each entry point below names the reference interface it replaces.  Plain pointers and sizes only; no
 * exceptions cross the ABI; every function returns a ccsm_status and ccsm_last_error() gives the text.
 *
 * Threading: a ccsm_model is immutable after creation and may be shared by threads; a ccsm_workspace is
 * single-owner (one forward in flight per workspace).  Use several workspaces on several HIP streams to keep
 * more than one batch in flight on a GPU (the reference runs one model instance per worker process,
 * call_modifications.py:561-578).
 */
#ifndef CCSM_H_
#define CCSM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCSM_SEQ_LEN 21
#define CCSM_HIDDEN 256
#define CCSM_LAYERS 3
#define CCSM_CLASSES 2

typedef enum {
    CCSM_OK = 0,
    CCSM_ERR_INVALID_ARG = 1, /* NULL pointer, n_sites out of range, ... */
    CCSM_ERR_UNSUPPORTED = 2, /* configuration outside what this build implements; mirrors the reference's
                                 ValueError("--model_type not right!") call_modifications.py:340 */
    CCSM_ERR_HIP = 3,         /* a HIP runtime call failed; text in ccsm_last_error() */
    CCSM_ERR_NOMEM = 4,
    CCSM_ERR_CAPACITY = 5,    /* n_sites exceeds the workspace's max_sites */
    CCSM_ERR_BUSY = 6         /* the model has slices of a group bound to a workspace and not yet run (ccsm_group_add_device .. ccsm_group_run) */
} ccsm_status;

/* Arithmetic of the contraction (matmul) work; everything else is fp32.  The reference computes in fp32 (models.py:125-130,
 * utils/constants_torch.py:9-12); every operand v here is carried as fp16 hi = fp16(v) plus the residual lo = v - hi, fp32 accumulation.
 * SPLIT3: hi*hi + lo*hi + hi*lo, three passes on v_mfma_f32_32x32x16_f16: fp32-class (max |dprob| <= 6e-7 against the oracle on every
 *         checkpoint tried, trained ones included).  What `precision 0` serves to every checkpoint whose probe is not clean (below):
 *         in practice every TRAINED checkpoint.
 * SPLIT_F8 ("split-mx"; the enumerator's name is historical - round 1 used fp8 operands): hi*hi on the fp16 MFMA, both correction
 *         products in ONE block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 per pair of k-blocks, weights fp4 e2m1 (fp6 for the n gate's
 *         input part), activations fp6 e2m3, per-32-block E8M0 scales; 1.55 MFMA passes per flop.  4-9e-6 max |dprob| on the
 *         synthetic initialisation (light-tailed: 8.5e-6 over 2^20 sites); on trained checkpoints a HEAVY tail (sites beyond 1e-4).
 * SPLIT_MXD: SPLIT_F8 with fp6 recurrent weight blobs and activation blobs scaled per (row, 32-value block).
 * HYBRID: SPLIT_MXD's input part; recurrent part in three fp16 passes on an exact fp16 hi + lo state.
 *         Both are lighter-tailed than SPLIT_F8 and still leave single sites beyond 1e-4 on trained checkpoints
 *         (profiles/r04_a_tail_study.log): explicit choices only, never selected by `precision 0`. */
typedef enum { CCSM_PRECISION_SPLIT_F8 = 4, CCSM_PRECISION_SPLIT3 = 3, CCSM_PRECISION_HYBRID = 5, CCSM_PRECISION_SPLIT_MXD = 6 } ccsm_precision;

/* Mirrors ModelAttRNN.__init__ (models.py:18-22) as called from call_modifications.py:315-323. */
typedef struct {
    int32_t seq_len;     /* 21 (odd; call_modifications.py:500-501 rejects even values) */
    int32_t num_layers;  /* 3 */
    int32_t num_classes; /* 2 */
    int32_t hidden_size; /* 256 */
    int32_t is_npass;    /* 1 */
    int32_t is_sn;       /* 0   the optional input features of models.py:39-47, 100-123: any combination (8 + 2 + npass 1 + stds 2 + */
    int32_t is_map;      /* 0   sn 4 + map 1 columns; the 17- and 18-column ones run as [one-hot(5) | features] against a layer-0 matrix */
    int32_t is_stds;     /* 0   with the embedding table folded in: the same product in 14 or 15 columns) */
    const char* model_type; /* "attbigru2s" */
    int32_t precision;   /* ccsm_precision; 0 = default: SPLIT_F8 if the probe of ccsm_create is clean, else SPLIT3 (see ccsm_model_probe_*) */
} ccsm_config;

/* Host fp32 parameter tensors in the reference state_dict layout (models.py:32-61).
 * Index [layer][dir]: dir 0 = "", dir 1 = "_reverse".  Row order of the 768 = [r; z; n]. */
typedef struct {
    const float* embed_weight;                 /* embed.weight (5, 8) */
    const float* weight_ih[CCSM_LAYERS][2];    /* rnn.weight_ih_l{l}{sfx} (768, 11 | 512) */
    const float* weight_hh[CCSM_LAYERS][2];    /* rnn.weight_hh_l{l}{sfx} (768, 256) */
    const float* bias_ih[CCSM_LAYERS][2];      /* rnn.bias_ih_l{l}{sfx} (768) */
    const float* bias_hh[CCSM_LAYERS][2];      /* rnn.bias_hh_l{l}{sfx} (768) */
    const float* att_wa;                       /* _att3.Wa.weight (256, 512) */
    const float* att_ua;                       /* _att3.Ua.weight (256, 512) */
    const float* att_va;                       /* _att3.va.weight (1, 256) */
    const float* fc1_weight;                   /* fc1.weight (2, 1024) */
    const float* fc1_bias;                     /* fc1.bias (2) */
} ccsm_weights;

/* One strand's per-site 21-mer features = the tensors `kmer, kpass, ipd_means, ipd_stds, pw_means, pw_stds, sns, maps` of
 * ModelAttRNN.forward (models.py:89-90).  The last four are read only by a model created with is_stds / is_sn / is_map (NULL
 * otherwise: the reference passes placeholders there), npass only with is_npass. */
typedef struct {
    const void* kmer;   /* (N,21) base codes A0 C1 G2 T3 other4 (process_utils.py:26-29): uint8, or float32 when
                           ccsm_batch.kmer_is_f32 (the reference hands FloatTensors, truncated by .int(), models.py:91) */
    const float* ipd;   /* (N,21) normalised IPD */
    const float* pw;    /* (N,21) normalised PW */
    const float* npass; /* (N) subread passes, or (N,21) when ccsm_batch.npass_per_base (call_modifications.py:96) */
    const float* ipd_std; /* (N,21) is_stds (models.py:105-111) */
    const float* pw_std;  /* (N,21) is_stds */
    const float* sn;      /* (N,4)  is_sn: one signal-to-noise quadruple per site, expanded over the 21 positions (models.py:112-116) */
    const float* map;     /* (N,21) is_map (models.py:117-121) */
} ccsm_strand;

typedef struct {
    ccsm_strand strand[2]; /* [0] = forward-strand features, [1] = reverse-strand ("2"-suffixed arguments) */
    int32_t kmer_is_f32;
    int32_t npass_per_base;
} ccsm_batch;

/* Initial hidden state.  The reference draws torch.randn(6, N, 256) twice per forward, strand 1 first
 * (models.py:77-87, 125-130), unseeded in its worker processes, so its outputs are only reproducible when h0 is
 * pinned: EXPLICIT takes both tensors (index 2l = layer-l forward, 2l+1 = backward), DEVICE_RNG draws N(0,1) on
 * the GPU (Philox4x32-10 keyed by seed, counter = offset + site index, or per-site keys: below), ZERO is for tests. */
typedef enum { CCSM_H0_EXPLICIT = 0, CCSM_H0_ZERO = 1, CCSM_H0_DEVICE_RNG = 2 } ccsm_h0_mode;
typedef struct {
    int32_t mode;
    const float* h0[2]; /* EXPLICIT: (6, N, 256) fp32 for strand 1 and strand 2 */
    uint64_t seed;
    uint64_t offset;
    /* DEVICE_RNG, optional (NULL: the counter is offset + site index): the caller names each site's random stream itself, counter =
     * (site_key[i], site_sub[i]) with site_sub < 2^28 (NULL = 0).  call_mods passes (64-bit hash of the read name, position of the C in
     * the read): a site's initial states - and with them its probability - then do not depend on the read's place in the file, on
     * how sites are batched or on which GPU handles the read.  Host arrays for the *_host entry points, device arrays for *_device. */
    const uint64_t* site_key;
    const uint32_t* site_sub;
} ccsm_h0;

typedef struct ccsm_model ccsm_model;
typedef struct ccsm_workspace ccsm_workspace;

/* Replaces: ModelAttRNN(...) + torch.load + load_state_dict + .cuda(device) + .eval()
 * (call_modifications.py:315-369).  Packs the weights into MFMA fragments and uploads them to `device`. */
ccsm_status ccsm_create(const ccsm_config* cfg, const ccsm_weights* w, int device, ccsm_model** out);
void ccsm_destroy(ccsm_model* m);

/* Device buffers (activations, h0, staging) + pinned host staging for batches of up to max_sites sites. */
ccsm_status ccsm_workspace_create(const ccsm_model* m, int max_sites, ccsm_workspace** out);
void ccsm_workspace_destroy(ccsm_workspace* ws);

/* Replaces: model(FloatTensor(...) x16) + .cpu()  (call_modifications.py:201-211) for a batch in HOST memory.
 * Copies the features into the workspace's pinned staging buffers, enqueues H2D (hipMemcpyAsync), the forward
 * kernels and D2H on `stream`, waits, and writes logits/probs (N,2 fp32, row-major) to the caller's buffers.
 * batch / h0 pointers are host pointers.  stream = hipStream_t (NULL = default stream). */
ccsm_status ccsm_forward_host(const ccsm_model* m, ccsm_workspace* ws, int n_sites, const ccsm_batch* batch,
                              const ccsm_h0* h0, float* logits, float* probs, void* stream);

/* Pipelined form of the same call: submit returns once the work is enqueued (inputs already copied out of the
 * caller's buffers); wait blocks until the batch is done and copies the results out.  With two workspaces this is
 * the pinned double-buffered hipMemcpyAsync stream the reference's per-tensor pageable copies
 * (utils/constants_torch.py:9-12) are replaced by. */
ccsm_status ccsm_submit_host(const ccsm_model* m, ccsm_workspace* ws, int n_sites, const ccsm_batch* batch,
                             const ccsm_h0* h0, void* stream);
ccsm_status ccsm_wait_host(ccsm_workspace* ws, float* logits, float* probs);

/* Same forward with every pointer (batch, h0, logits, probs) already in DEVICE memory; asynchronous on `stream`
 * (replaces model(...) for callers that keep features resident in HBM). */
ccsm_status ccsm_forward_device(const ccsm_model* m, ccsm_workspace* ws, int n_sites, const ccsm_batch* batch,
                                const ccsm_h0* h0, float* logits, float* probs, void* stream);

/* Micro-batch coalescing: several independent caller batches ("slices", each with its own outputs and initial
 * states) are packed back to back into one workspace and run by ONE launch of each heavy kernel, so that a launch
 * fills the GPU (one 2048-site batch is only 86 workgroups of 96 strand rows; three are exactly 256 = one per CU)
 * and every CU streams the SAME layer's weights from L2.  ccsm_group_add_device binds a device-resident batch to the
 * next free rows (runs the cheap per-batch kernels); ccsm_group_run runs the BiGRU / attention kernels over every
 * slice added since the last run and writes each slice's logits/probs.  Total sites per group <= the workspace's
 * max_sites; at most 16 slices.  ccsm_forward_device == add one slice + run.  Everything is asynchronous on `stream`. */
ccsm_status ccsm_group_add_device(const ccsm_model* m, ccsm_workspace* ws, int n_sites, const ccsm_batch* batch,
                                  const ccsm_h0* h0, float* logits, float* probs, void* stream);
ccsm_status ccsm_group_run(const ccsm_model* m, ccsm_workspace* ws, void* stream);
int ccsm_group_pending(const ccsm_workspace* ws);

/* Read-level entry (SURVEY.md 8 a-2/a-3 + the model): replaces, for a chunk of CCS reads, the host work of
 * extract_features.py:297-405 (CodecV1 decode, per-read float64 z-score rounded to 6 decimals, CG scan, forward and
 * reverse-complement 21-mer windows), the batching of call_modifications.py:95-142 and the forward of :201-211.
 * All arrays are HOST pointers; the per-base arrays are the reads' values concatenated, read r occupying
 * [offset[r], offset[r] + length[r]).  seq = the read's forward sequence, upper-case ASCII (fwd_seq, extract_features.py:288-289);
 * fi/ri/fp/rp = the tags' raw CodecV1 bytes exactly as stored (the reference does not flip ri/rp, :314-319; --no_decode is
 * not offered); fn/rn = subread passes.  Reads whose tag lengths differ from the sequence length are the caller's to skip
 * (:320-325). */
typedef struct ccsm_reads {
    int32_t n_reads;
    const int64_t* offset;   /* (n_reads) */
    const int32_t* length;   /* (n_reads) */
    const uint8_t* seq;
    const uint8_t* fi;
    const uint8_t* ri;
    const uint8_t* fp;
    const uint8_t* rp;
    const float* fn;         /* (n_reads) */
    const float* rn;         /* (n_reads) */
    const uint64_t* h0_key;  /* (n_reads) or NULL.  DEVICE_RNG with keys: site k of read r draws its initial states from the counter
                                (h0_key[r], locs[k]) instead of (offset + running site index): see ccsm_h0.site_key.
                                ccsm_bam_batch.name_hash is such a key */
} ccsm_reads;
/* Outputs (host): first_site (n_reads + 1) prefix of kept CG sites per read; locs (capacity max_sites) position of
 * each site's C in its read, in read order then ascending; logits / probs (capacity max_sites x 2); *n_sites.
 * h0 explicit tensors, if given, are HOST (6, n_sites, 256) per strand.  CCSM_ERR_CAPACITY when the reads hold more
 * than the workspace's max_sites sites (nothing is computed; *n_sites is not set).  Blocks until done. */
/* Pipelined form (two workspaces on two streams keep the GPU busy while the host prepares the next chunk): submit copies the
 * arrays into the workspace's pinned block and enqueues everything on `stream`.  site_counts = the caller's per-read kept-site
 * counts (e.g. ccsm_bam_batch.n_sites): with them nothing in submit waits for the GPU; the device scan must agree, or wait
 * returns CCSM_ERR_INVALID_ARG.  NULL: submit makes one round trip for the counts (that is what ccsm_forward_reads_host does).
 * Explicit h0 tensors must stay valid until submit returns. */
ccsm_status ccsm_submit_reads_host(const ccsm_model* m, ccsm_workspace* ws, const ccsm_reads* reads, const int32_t* site_counts,
                                   const ccsm_h0* h0, void* stream);
ccsm_status ccsm_wait_reads_host(ccsm_workspace* ws, int32_t* first_site, int32_t* locs, float* logits, float* probs,
                                 int32_t* n_sites);
ccsm_status ccsm_forward_reads_host(const ccsm_model* m, ccsm_workspace* ws, const ccsm_reads* reads, const ccsm_h0* h0,
                                    int32_t* first_site, int32_t* locs, float* logits, float* probs, int32_t* n_sites,
                                    void* stream);

/* Diagnostics */
const char* ccsm_last_error(void);
const char* ccsm_version(void);
int ccsm_model_precision(const ccsm_model* m);
/* precision 0 (default) picks the arithmetic by measurement, conservatively: ccsm_create runs up to 8 probe batches of 8192 synthetic sites
 * (deterministic inputs and device-drawn initial states) through SPLIT3 and SPLIT_F8 and serves SPLIT_F8 only if over ALL 65536 sites
 * max |dprob| <= 1.25e-5 (an eighth of the 1e-4 bar) AND max <= 3 x the 99.9th percentile (a light tail); otherwise SPLIT3.  The probe ends
 * at the first batch that breaks the first condition.  SPLIT_MXD and HYBRID are never selected (explicit choices).  The same weights
 * always get the same answer.  CCSM_NO_PRECISION_FALLBACK in the environment forces SPLIT_F8 and says so on stderr when the probe failed.
 * ccsm_model_probe_error = max |dprob| over the probe sites run, _probe_q999 their 99.9th percentile, _probe_tail(m, 4) the share beyond
 * 1e-5, _probe_sites how many were run (-1 / 0 when a precision was forced); _probe_error_hybrid / _error_of / _tail of 5, 6: -1 (those
 * candidates left the probe in round 4; kept for the ABI); ccsm_model_precision = the arithmetic in use; ccsm_model_quant_error =
 * relative RMS quantisation error of the weight correction blobs (worst layer). */
float ccsm_model_probe_error(const ccsm_model* m);
float ccsm_model_probe_error_hybrid(const ccsm_model* m);
float ccsm_model_probe_error_of(const ccsm_model* m, int precision);
float ccsm_model_probe_tail(const ccsm_model* m, int precision);
float ccsm_model_probe_q999(const ccsm_model* m);
int ccsm_model_probe_sites(const ccsm_model* m);
float ccsm_model_quant_error(const ccsm_model* m);
/* The same rule on the CALLER's data.  ccsm_create's probe runs synthetic sites; a caller that wants the rule applied to what it
 * actually feeds (call_mods: the first <= 65536 sites of its input; reference call site call_modifications.py:201-214) runs those sites
 * through the arithmetic in use and through SPLIT3 - ccsm_model_set_precision switches between the two: SPLIT3's weight streams are
 * always resident, of the split-mx family the one ccsm_create probed or was asked for, anything else is CCSM_ERR_INVALID_ARG - and hands
 * both (n, 2) probability arrays to ccsm_model_data_probe_add (any number of calls).  ccsm_model_data_probe_decide applies the rule to all
 * sites added (max |dprob| <= 1.25e-5 and, from 8192 sites on, max <= 3 x the 99.9th percentile), switches the model to SPLIT3 if the
 * candidate fails (unless CCSM_NO_PRECISION_FALLBACK is set), forgets the sites and returns the precision now in use.
 * _data_probe_error / _q999 / _sites: the figures of the last decision (-1 / -1 / 0 before one); _verdict: -1 none, 0 failed, 1 kept.
 * Not thread-safe against forwards on the same model (single-threaded use per model, as everywhere in this interface); refused with
 * CCSM_ERR_BUSY while slices of a group are bound to a workspace of this model and not yet run (they were packed for one arithmetic). */
ccsm_status ccsm_model_set_precision(ccsm_model* m, int precision);
/* The same without touching the model: the NEXT forward / submit / group run issued on this workspace runs in SPLIT3 whatever the model's
 * arithmetic (one-shot; the flag is consumed by that run).  What a caller uses to shadow a sample of its launches with the fp32-class
 * arithmetic while the model keeps serving split-mx on its other workspaces (call_mods: one chunk in 64, behind the two probes - the
 * rule applied to the WHOLE input, not to its first 65536 sites; reference call site call_modifications.py:201-214). */
ccsm_status ccsm_workspace_force_split3(ccsm_workspace* ws);
ccsm_status ccsm_model_data_probe_add(ccsm_model* m, const float* probs_candidate, const float* probs_split3, int n_sites);
int ccsm_model_data_probe_decide(ccsm_model* m);
float ccsm_model_data_probe_error(const ccsm_model* m);
float ccsm_model_data_probe_q999(const ccsm_model* m);
int ccsm_model_data_probe_sites(const ccsm_model* m);
int ccsm_model_data_probe_verdict(const ccsm_model* m);
size_t ccsm_workspace_bytes(const ccsm_workspace* ws);
/* Times (ms, HIP events on `stream`) of the kernels of the LAST forward issued on this workspace with timing
 * enabled: out[0..2] = GRU layers 0..2, out[3] = attention+FC, out[4] = logits/softmax finalize.  Blocks. */
ccsm_status ccsm_workspace_set_timing(ccsm_workspace* ws, int enable);
ccsm_status ccsm_workspace_last_timing(ccsm_workspace* ws, float out_ms[5]);
/* Mean of the same five durations over every run issued on this workspace since timing was enabled (up to the last
 * 128 runs; *n_runs = how many were averaged).  Blocks until those runs have finished. */
ccsm_status ccsm_workspace_timing_mean(ccsm_workspace* ws, float out_ms[5], int* n_runs);
/* Test hook: copy an internal device buffer to the host after a device sync.  which: 0 = layer-0 input fragments,
 * 1/2 = activation fragment buffers A/B (layer 0 and 2 write A, layer 1 writes B), 3 = h0 buffer, 4 = logit halves.
 * NOTE: buffers are laid out for the padded row count of the workspace's max_sites only when n_sites == max_sites;
 * in general tile/row strides follow the padded row count of the LAST forward's n_sites. */
ccsm_status ccsm_debug_read(ccsm_workspace* ws, int which, void* host_dst, size_t bytes);
/* Padded strand-row count (2 * n_sites rounded up to the kernels' row tile) that lays out those buffers. */
int ccsm_debug_rows_padded(int n_sites);
/* Host-side OCP fp8 e4m3fn encoder used to pack the SPLIT_F8 weight fragments (round to nearest even, saturating). */
int ccsm_debug_fp8_e4m3(float v);
/* Row stride of the h0 buffer (= padded row capacity of the workspace). */
int ccsm_debug_rows_capacity(const ccsm_workspace* ws);
/* Runs one 32x32x16 MFMA tile with this library's fragment conventions against a host reference. */
ccsm_status ccsm_selftest_mfma(int device, float* max_abs_err);
/* One 32x32x32 product in SPLIT_F8 arithmetic (host-packed weight fragments, device-packed activation fragments) against
 * a float64 host reference: *err_corr with the fp8 correction MFMA, *err_main_only without it (fp16 operands only). */
ccsm_status ccsm_selftest_split_f8(int device, float* err_corr, float* err_main_only);
/* The same for the split-mx product of the GRU layers (fp6 / fp4 correction blobs, host-packed weights, device-packed
 * activations); blob_mismatch = bytes in which the host's fp6 encoder and v_cvt_scalef32_pk32_fp6_f16 disagree (0 expected). */
ccsm_status ccsm_selftest_split_mx(int device, int weight_fmt /* 2 = fp6 blob, 4 = fp4 blob */, float* err_corr, float* err_main_only,
                                   int* blob_mismatch);

/* What the MFMA pipe of THIS device sustains under its package power cap with random register-resident operands, in fp16-MFMA
 * TFLOP/s (correction products are overhead, as in bench.py's roofline.achieved): mode 0 = v_mfma_f32_32x32x16_f16 only, 1 = the GRU
 * kernels' issue mix (per two fp16 MFMAs one block-scaled fp4 x fp6 K = 64 MFMA), 2 = that mix with the B operands re-read from LDS,
 * 3 = v_mfma_f32_16x16x32_f16 only (what split3's GRU layers issue), 4 = the mix on the 16-wide instructions (16x16x32 + 16x16x128).
 * One 512-thread workgroup per CU (2 waves per SIMD), run for `seconds` (the second half is averaged).  *issue_gcycles (optional) =
 * MFMA issue cycles per second and SIMD, in GHz.  bench.py quotes mode 1 as roofline.peak_power_capped. */
ccsm_status ccsm_measure_mfma_ceiling(int device, int mode, double seconds, float* tflops, float* issue_gcycles);

/* ------------------------------------------------------------------------------------------------------------------
 * Aggregate mode (`ccsmeth call_freqb --call_mode aggregate`, BASELINE config 5).
 * Replaces: AggrAttRNN(...) + load_state_dict (call_mods_freq_bam.py:316-342, models.py:625-694) and the model loop of
 * _cal_modfreq_in_aggregate_mode (call_mods_freq_bam.py:295-304).  The 11-site windows are built on the device from the
 * per-site histogram table (the reference materialises an (M,11,21) tensor on the host).
 * h0: the reference re-seeds per region (torch.manual_seed(tseed), :313) and draws torch.randn(2, B, 32) per batch of
 * 1024, so results ARE deterministic; the library carries a replica of that stream (mt19937 -> 24-bit uniforms ->
 * Box-Muller in blocks of 16, after the 14 753 draws the reference's model construction consumes) for `stream_sites`
 * site evaluations.  `stream_pos` = number of values consumed so far in the region (64 per site; the reference calls
 * all -> hp1 -> hp2 on one stream, :394-414). */
typedef struct {
    const float* weight_ih[2];  /* rnn.weight_ih_l0[_reverse] (96, 21) */
    const float* weight_hh[2];  /* rnn.weight_hh_l0[_reverse] (96, 32) */
    const float* bias_ih[2];    /* (96) */
    const float* bias_hh[2];    /* (96) */
    const float* att_wa;        /* _att3.Wa.weight (32, 64) */
    const float* att_ua;        /* _att3.Ua.weight (32, 64) */
    const float* att_va;        /* _att3.va.weight (1, 32) */
    const float* fc1_weight;    /* fc1.weight (1, 64) */
    const float* fc1_bias;      /* fc1.bias (1) */
} ccsm_aggr_weights;
typedef struct ccsm_aggr_model ccsm_aggr_model;

ccsm_status ccsm_aggr_create(const ccsm_aggr_weights* w, int device, uint64_t seed, int64_t stream_sites, ccsm_aggr_model** out);
void ccsm_aggr_destroy(ccsm_aggr_model* m);
/* --only_close (call_mods_freq_bam.py:285-290): the 21st input becomes "this window site lies exactly 2 bases after its predecessor"
 * (0/1, over the padded position sequence) instead of the distance to the centre site.  Applies to the following forward calls. */
ccsm_status ccsm_aggr_set_only_close(ccsm_aggr_model* m, int only_close);
/* refposes (M) int64 sorted positions, histos (M,20) fp32 normalised histograms (_get_normalized_histo), out (M) fp32 raw
 * fc1 outputs (the caller applies round(clip(y,0,1),6), call_mods_freq_bam.py:302).  Host pointers, synchronous. */
ccsm_status ccsm_aggr_forward_host(ccsm_aggr_model* m, int64_t n_sites, const int64_t* refposes, const float* histos,
                                   int64_t stream_pos, float* out, void* stream);
/* Same with device pointers, asynchronous on `stream`. */
ccsm_status ccsm_aggr_forward_device(ccsm_aggr_model* m, int64_t n_sites, const int64_t* refposes, const float* histos,
                                     int64_t stream_pos, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CCSM_H_ */
