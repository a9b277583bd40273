/* libccsm_train — C-ABI of the attbigru2s training step on MI355X (`ccsmeth trainm`).
 *
 * What it replaces in the reference (PengNi/ccsmeth v0.5.0, train_multigpu.py):
 *   outputs, _ = model(*16 tensors); loss = CrossEntropyLoss(weight=[1, pos_weight])(outputs, labels)   :283-286
 *   optimizer.zero_grad(); loss.backward()                                                               :309-310
 *   torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5); optimizer.step()   (torch.optim.Adam)       :311-312
 * DDP's bucketed gradient all-reduce (:171-172) stays with the caller: the gradients live in ONE flat fp32 device buffer (the
 * caller may supply it, e.g. the storage of a torch tensor) that is all-reduced over RCCL between
 * ccsm_train_forward_backward and ccsm_train_step.
 *
 * Arithmetic: fp32 values throughout; every matrix product - the fused recurrent kernels and the plain ones (ccsm_train_gemm.hip) - runs on
 * the matrix cores with fp16 hi + lo split operands (three passes, fp32 accumulation: fp32-class), hand-written; no BLAS library; the two
 * strands run as one batch of 2N rows through the shared GRU / attention weights, so their gradient contributions add up in
 * the same products.  Parameters, gradients and Adam moments are flat fp32 arrays in the order of model.parameters()
 * (= the state_dict order of SURVEY.md 8 a-4): embed.weight, then for l, sfx: weight_ih, weight_hh, bias_ih, bias_hh, then
 * _att3.Wa.weight, _att3.Ua.weight, _att3.va.weight, fc1.weight, fc1.bias  (3 043 114 values).
 */
#ifndef CCSM_TRAIN_H_
#define CCSM_TRAIN_H_

#include "ccsm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ccsm_trainer ccsm_trainer;

const char* ccsm_train_last_error(void);

/* Number of parameters (3 043 114) and the offset of each tensor in the flat order; offsets has 30 entries (+1 = total). */
int64_t ccsm_train_num_params(void);
int ccsm_train_param_offsets(int64_t* offsets, int n);

/* Replaces ModelAttRNN(...).cuda(rank) + (optional) load of --init_model (train_multigpu.py:116-160) and the optimizer's
 * state.  w = host fp32 tensors (ccsm.h).  max_sites = largest batch.  d_grads = device pointer to a flat fp32 buffer of
 * ccsm_train_num_params() values that receives the gradients, or NULL to let the trainer own one. */
ccsm_status ccsm_train_create(const ccsm_weights* w, int device, int max_sites, float* d_grads, ccsm_trainer** out);
void ccsm_train_destroy(ccsm_trainer* t);

/* One forward + backward of a batch in HOST memory.  labels: (N) 0/1.  h0: as ccsm_forward_host (EXPLICIT / ZERO /
 * DEVICE_RNG; the reference draws torch.randn per forward, models.py:77-87).  pos_weight: CrossEntropyLoss weight of class 1.
 * dropout_rate: nn.GRU inter-layer dropout and dropout1 (masks from a counter-based generator keyed by dropout_seed; 0 = none).
 * On return the flat gradient buffer holds d(loss)/d(parameters) (overwritten, not accumulated), *loss the weighted mean
 * cross entropy, logits (optional, host (N,2)) the forward outputs. */
ccsm_status ccsm_train_forward_backward(ccsm_trainer* t, int n_sites, const ccsm_batch* batch, const int32_t* labels,
                                        const ccsm_h0* h0, float pos_weight, float dropout_rate, uint64_t dropout_seed,
                                        float* loss, float* logits);

/* Forward only (validation, train_multigpu.py:330-376: model.eval(), no dropout): host logits (N,2) and, when labels is not
 * NULL, the weighted loss. */
ccsm_status ccsm_train_eval(ccsm_trainer* t, int n_sites, const ccsm_batch* batch, const int32_t* labels, const ccsm_h0* h0,
                            float pos_weight, float* loss, float* logits);

/* clip_grad_norm_(max_norm) + one Adam step (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay).
 * max_norm <= 0 skips the clipping.  *grad_norm = the total norm before clipping. */
ccsm_status ccsm_train_step(ccsm_trainer* t, float lr, float beta1, float beta2, float eps, float max_norm, float* grad_norm);

/* The fused recurrent backward kernels (batches of >= 384 sites) carry the gate gradients as scaled fp16 pairs, exact up to
 * |gradient| = 14.6; a step in which one exceeds that (a sum-reduced loss, an extreme pos_weight) is detected on the device and its
 * backward pass is repeated step by step in fp32 before ccsm_train_forward_backward returns.  How often that has happened: */
long ccsm_train_fused_fallbacks(const ccsm_trainer* t);
/* Self-test of the training step's matrix-product kernel (ccsm_train_gemm.hip; the products torch autograd runs on a BLAS library
 * inside train_multigpu.py:283-312): C (M x N, ldc) = alpha op(A) op(B) + beta C on `device`, HOST pointers, row-major storage; t_a: A
 * is stored K x M; t_b: B is stored N x K; grad_a: treat A as a gradient operand (scaled into fp16's range by its maximum). */
ccsm_status ccsm_train_selftest_gemm(int device, int t_a, int t_b, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                                     int ldb, float beta, float* C, int ldc, int grad_a);

/* Flat buffers: device pointer of the gradients (for the caller's all-reduce), and host copies in / out. */
ccsm_status ccsm_train_grad_ptr(ccsm_trainer* t, float** d_grads);
ccsm_status ccsm_train_get_params(ccsm_trainer* t, float* host_flat);
ccsm_status ccsm_train_set_params(ccsm_trainer* t, const float* host_flat);
ccsm_status ccsm_train_get_grads(ccsm_trainer* t, float* host_flat);

#ifdef __cplusplus
}
#endif
#endif
