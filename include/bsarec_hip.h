/* bsarec_hip.h -- C ABI of the MI355X (gfx950) BSARec training hot path.
 *
 * The reference (Sun-Sir/BSARec) is pure Python/PyTorch and has no FFI: the seam this library
 * slots under is the Python object protocol between Trainer and the model
 * (src/trainers.py:94-116 -> src/model/bsarec.py:16-37).  Each entry point below names the
 * reference code it replaces.  Conventions:
 *   - plain pointers and sizes only; every device buffer is allocated and owned by the caller
 *     (PyTorch in the shipped host code); the library never allocates or frees device memory;
 *   - all work is enqueued on the caller's hipStream_t (passed as void*), never synchronises, and
 *     can therefore be captured into a hipGraph (the two exceptions say so: bsarec_plan_create
 *     synchronises once, bsarec_profile_read waits for its own events);
 *   - no process-wide mutable state: every option is a field of bsarec_config_t and belongs to the
 *     plan, profiling / diagnostic state is per plan; two plans with different options may be
 *     driven from different threads;
 *   - return value: 0 = OK, < 0 = invalid argument / unsupported shape (nothing was launched),
 *     > 0 = hipError_t of a failed launch;
 *   - fp32 arithmetic and fp32 tensors by default (the reference's arithmetic type); cfg.storage = 1 keeps the
 *     saved activations / inter-block gradients / a shadow of the Linear weights in bf16 and multiplies with
 *     bf16 MFMAs (fp32 accumulation, fp32 masters, LayerNorm / softmax / loss / Adam in fp32) at the fused
 *     shape, and multiplies with bf16 MFMAs on operands rounded while they are staged (fp32 tensors) at every
 *     other shape; ids and answers are int64 as produced by the reference DataLoader (src/dataset.py:108-115).
 */
#ifndef BSAREC_HIP_H
#define BSAREC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSAREC_MAX_LAYERS 16
#define BSAREC_ABI_VERSION 8

/* Hyper-parameters the reference model reads from `args`
 * (src/utils.py:83-96; src/model/bsarec.py:71-88; src/model/_modules.py:79-87). */
typedef struct {
    int batch;          /* B: sequences per call (src/utils.py:67) */
    int seq_len;        /* L = max_seq_length */
    int hidden;         /* d = hidden_size, multiple of 4, <= 256 */
    int heads;          /* num_attention_heads; d/heads multiple of 4 */
    int layers;         /* num_hidden_layers, <= BSAREC_MAX_LAYERS */
    int item_size;      /* V = max item id + 1 (row 0 = padding, still a class) */
    int cutoff_bins;    /* min(c//2 + 1, L//2 + 1): rFFT bins kept by FrequencyLayer */
    float alpha;        /* BSARecLayer mix (src/model/bsarec.py:78) */
    float ln_eps;       /* 1e-12 */
    float p_hidden;     /* hidden_dropout_prob */
    float p_attn;       /* attention_probs_dropout_prob */
    int filter_kind;    /* 0: BSARec's FrequencyLayer (low-pass + beta^2 high-pass, src/model/bsarec.py:90-104);
                         * 1: FMLPRec's learnable complex filter irfft(rfft(x) * W) (src/model/fmlprec.py:96-113): the
                         *    layer's filter_cw tensor is used, cutoff_bins must be L/2 + 1, generic kernels only */
    /* ---- per-plan options; 0 selects the default everywhere, so a zero-filled tail is a valid configuration ---- */
    int hidden_act;     /* FeedForward activation (src/model/_modules.py:38-59, ACT2FN): 0 gelu (erf form, the default), 1 relu,
                         * 2 swish, 3 tanh, 4 sigmoid.  Non-default activations run on the generic tiled kernels */
    int storage;        /* 0: fp32 everywhere (the reference's arithmetic); 1 (config C2 / the bf16 half of C3): bf16 MFMA with fp32
                         *    accumulation, fp32 master weights, fp32 LayerNorm / softmax / loss head / Adam.  At the fused shape
                         *    (hidden = 64, L <= 64) the saved activations, the inter-block gradients and a shadow of the Linear weights
                         *    are STORED as bf16; at every other shape (generic tiled kernels) all tensors stay fp32 and the operands of
                         *    every matrix product of the block stack are rounded to bf16 as they are staged into LDS
                         *    (bsarec_buffer_is_bf16 = 0 everywhere, `shadow` unused) */
    int no_fused;       /* 1: never take the fused per-sequence block kernels (hidden = 64, L <= 64, cutoff_bins <= 8) */
    int no_prune_top;   /* 1: bsarec_forward_last evaluates the full top block (no one-row evaluation) */
    int dw_tiled;       /* 1: LDS-tiled grouped weight-gradient kernel at the fused shape too (default: direct split-K) */
    int splits;         /* split-K slab slices of the weight-gradient products (0: 32 at the fused shape, 40 elsewhere) */
    int top_slabs;      /* slab slices of the one-row top block's weight-gradient products (0: 2) */
    int separate_embed; /* 1: the embedding front-end runs as its own kernel on the fused path too */
    int separate_top;   /* 1: the one-row top block of the loss path runs as its own kernels instead of as the tail /
                         * head of the launches of the block below it */
    int chain_kernels;  /* 1: the register-chain forward block kernel (fused_chain.h: lane = token, accumulators chained as MFMA
                         * operands, LDS weight ring) instead of the LDS-phase kernel (fused_layer.h) at the fused shape in
                         * fp32; measured equal in speed on MI355X, kept selectable */
    int x3_products;    /* 1: the fused block kernels evaluate every fp32 product of their matrix multiplications on the bf16
                         * matrix cores as six bf16 x bf16 partial products of the operands' exact three-way bf16 splits
                         * (fp32 accumulation; error <= 2^-26 per product, below fp32 rounding; fp32 tensors, fp32 storage).
                         * 0 (default): v_mfma_f32_32x32x2_f32.  Ignored under storage = 1 */
} bsarec_config_t;

/* The 19 tensors of one BSARecBlock, in state_dict order (+ the sibling model's filter weight)
 * (item_encoder.blocks.{l}.layer.filter_layer.* , .attention_layer.* , .feed_forward.*). */
typedef struct {
    float *sqrt_beta, *filter_ln_w, *filter_ln_b;
    float *query_w, *query_b, *key_w, *key_b, *value_w, *value_b, *dense_w, *dense_b;
    float *attn_ln_w, *attn_ln_b;
    float *ffn1_w, *ffn1_b, *ffn2_w, *ffn2_b, *ffn_ln_w, *ffn_ln_b;
    float *filter_cw;   /* filter_kind 1 only: complex_weight [L/2+1, d, 2] (re, im) as the reference stores it; else null */
} bsarec_layer_t;

/* All 4 + 19 N tensors (parameters, or their gradients) as device pointers.  Linear weights are
 * [out, in] row-major exactly as nn.Linear stores them. */
typedef struct {
    float *item_emb;    /* [V, d]  item_embeddings.weight */
    float *pos_emb;     /* [L, d]  position_embeddings.weight */
    float *ln_w, *ln_b; /* [d]     LayerNorm.{weight,bias} */
    bsarec_layer_t layer[BSAREC_MAX_LAYERS];
} bsarec_tensors_t;

/* Named device buffers inside the workspace (byte offsets via bsarec_buffer_offset). */
enum {
    BSAREC_BUF_LAYER_OUT = 0,  /* [B,L,d]   output of layer l-1 (l = 0: embedding output), l in [0, N] */
    BSAREC_BUF_LOGITS = 1,     /* [B,Vp]    full-catalogue logits, Vp = 4*ceil(V/4), pad columns = 0 */
    BSAREC_BUF_LOSS = 2,       /* [1]       mean cross-entropy */
    BSAREC_BUF_DSP = 3,        /* [B,L,d]   FrequencyLayer output of layer l (generic path only; stays on chip in the fused path) */
    BSAREC_BUF_HMIX = 4,       /* [B,L,d]   alpha*dsp + (1-alpha)*gsp of layer l */
    BSAREC_BUF_PROBS = 5,      /* [B,h,L,Lp] attention probabilities of layer l (before dropout) */
    BSAREC_BUF_DLAYER_IN = 6,  /* [B,L,d]   gradient w.r.t. layer output l (ping-pong pair: only l = 0, 1 survive backward; on the fused path l = 0 is never materialised: the bottom block emits the embedding gradient directly) */
    BSAREC_BUF_LOSS_ROWS = 7,  /* [B]       per-sequence cross-entropy */
    BSAREC_BUF_CTX = 8,        /* [B,L,d]   attention context of layer l */
    BSAREC_BUF_DLOGITS = 9     /* [B,Vp]    d loss / d logits = (softmax - onehot(answer)) / B, pad columns = 0 (after bsarec_loss) */
};

typedef struct bsarec_plan bsarec_plan_t;   /* host-side launch plan (no device memory of its own) */

int bsarec_abi_version(void);

/* Bytes of device workspace a plan for `cfg` needs (activations kept for backward, scratch,
 * split-K slabs, reduction job table).  0 if cfg is unsupported. */
size_t bsarec_workspace_bytes(const bsarec_config_t *cfg);

/* Device step state: 8 x uint64.  [0] seed of the Philox dropout stream, [1] forward-step counter,
 * [2] Adam step t, [3] two packed floats written by bsarec_adam_step ({lr/bc1, sqrt(bc2)}), [4] ticket counter of
 * the Adam kernel (uint32), [5], [6] beta1^t, beta2^t as doubles, [7] reserved.  The caller zero-initialises it and
 * sets [0]; kernels read/advance it on the device so a captured graph replays correctly; t = 0 restarts Adam. */
#define BSAREC_STATE_BYTES 64

/* Build a launch plan.  `params`/`grads` hold device pointers (copied into the plan); `workspace`
 * must hold bsarec_workspace_bytes(cfg) bytes, 256-byte aligned; `twiddle` is the float[2L] table
 * (cos, sin)(2 pi j / L) built on the host in double precision.  Enqueues one small H2D copy of the
 * reduction job table on `stream` and waits for it (the one synchronising call of the training path).
 * `shadow` (cfg.storage = 1 at the fused shape only, else null / ignored): the same tensor set as `params` but as bf16 arrays (uint16_t behind the
 * float* fields) -- the bf16 shadow of the fp32 masters that the MFMA products read; only the six Linear weights of
 * every layer are used.  The caller keeps it current: bsarec_shadow_refresh after it changes the masters itself, or
 * bsarec_adam_t.shadow_bf16 so that the fused Adam writes both.
 * Replaces: model construction wiring of src/model/bsarec.py:8-14. */
int bsarec_plan_create(bsarec_plan_t **out, const bsarec_config_t *cfg, const bsarec_tensors_t *params,
                       const bsarec_tensors_t *grads, const bsarec_tensors_t *shadow, void *workspace,
                       size_t workspace_bytes, void *state, const float *twiddle, void *stream);
/* cfg.storage = 1: bf16 shadow <- fp32 masters for the Linear weights of every layer (one launch per layer). */
int bsarec_shadow_refresh(bsarec_plan_t *plan, void *stream);
/* Element type of a named workspace buffer under this plan: 0 = fp32, 1 = bf16 (cfg.storage = 1: every saved
 * activation of the block stack except the last layer's output; logits, loss and the statistics stay fp32). */
int bsarec_buffer_is_bf16(const bsarec_plan_t *plan, int buffer, int layer);

/* 1 if the plan resolved to the fused per-sequence block kernels (hidden = 64, L <= 64, ...), 0 for the generic tiled kernels. */
int bsarec_plan_is_fused(const bsarec_plan_t *plan);
/* The same question before a plan exists (does cfg.storage = 1 need a `shadow`?): 1 fused, 0 generic, < 0 invalid cfg. */
int bsarec_config_is_fused(const bsarec_config_t *cfg);
void bsarec_plan_destroy(bsarec_plan_t *plan);

/* Data-parallel bucketing: the dense part of the item-table gradient, dE = dlogits^T . h_last, is complete
 * right after the logits backward -- long before the rest of the gradient.  `hook` (may be null) is called by
 * bsarec_backward / bsarec_*_step_indexed on the calling thread as soon as that kernel is ENQUEUED on `stream`, so that
 * the host can start exchanging grads->item_emb on a side stream under the whole encoder backward (record an event on
 * `stream`, let the side stream wait for it).  `lookup_grad` (may be null): [V, d] buffer that then receives the
 * lookup-path rows (the embedding scatter at the END of the backward) instead of grads->item_emb, so that the early
 * exchange is not disturbed; hand it to the update as bsarec_adam_t.grads2. */
typedef void (*bsarec_hook_t)(void *user, void *stream);
int bsarec_plan_set_dense_grad_hook(bsarec_plan_t *plan, bsarec_hook_t hook, void *user, float *lookup_grad);

/* Byte offset of a named buffer inside the workspace, or -1. */
long bsarec_buffer_offset(const bsarec_plan_t *plan, int buffer, int layer);

/* Advance the forward-step counter (new dropout masks).  Call once per training step. */
int bsarec_step_begin(bsarec_plan_t *plan, void *stream);

/* BSARecModel.forward(input_ids, all_sequence_output=True)   (src/model/bsarec.py:16-28):
 * ids int64[B,L] (left-padded with 0) -> layer outputs in BSAREC_BUF_LAYER_OUT[0..N].
 * train != 0 applies dropout (model.train()), 0 is model.eval(). */
int bsarec_forward(bsarec_plan_t *plan, const int64_t *ids, int train, void *stream);

/* The rest of BSARecModel.calculate_loss (src/model/bsarec.py:32-35) on the forward's result:
 * logits = h[:, -1, :] @ E^T, mean CE against answers int64[B]; also prepares dlogits. */
int bsarec_loss(bsarec_plan_t *plan, const int64_t *answers, void *stream);

/* bsarec_forward for callers that consume only position L-1 of the last layer -- bsarec_loss, bsarec_logits and
 * bsarec_backward, i.e. calculate_loss (src/model/bsarec.py:30-37) and Trainer.predict_full of the last position
 * (src/trainers.py:126-129).  At the fused shape with >= 2 layers the top block is then evaluated on that row only
 * (it still attends to all positions) and its backward uses the exact one-row structure of the upstream gradient;
 * loss, logits and all parameter gradients are those of bsarec_forward.  Other rows of BSAREC_BUF_LAYER_OUT[N] are
 * left unspecified.  cfg.no_prune_top = 1 makes this identical to bsarec_forward. */
int bsarec_forward_last(bsarec_plan_t *plan, const int64_t *ids, int train, void *stream);

/* Sibling model SASRec (src/model/sasrec.py:41-63): the BCE head on one positive and one negative item at the last
 * position, over the rows with pos_ids != 0, instead of the full-catalogue CE.  Run it on a plan created with
 * alpha = 0 (then a BSARecBlock is exactly the reference's TransformerBlock, src/model/_modules.py:142-151).  Loss ->
 * BSAREC_BUF_LOSS; bsarec_backward then back-propagates this head (pos_ids / neg_ids must stay valid until then). */
int bsarec_loss_bce(bsarec_plan_t *plan, const int64_t *pos_ids, const int64_t *neg_ids, void *stream);
/* Sibling model FMLPRec's head (src/model/fmlprec.py:41-62): mean over ALL rows of
 * -log(sigmoid(x_pos) + 1e-24) - log(1 - sigmoid(x_neg) + 1e-24); otherwise as bsarec_loss_bce. */
int bsarec_loss_logsig(bsarec_plan_t *plan, const int64_t *pos_ids, const int64_t *neg_ids, void *stream);

/* logits only (Trainer.predict_full, src/trainers.py:62-68). */
int bsarec_logits(bsarec_plan_t *plan, void *stream);

/* loss.backward() (src/trainers.py:106): gradients of all tensors -> `grads` (overwritten, not
 * accumulated).  Must follow bsarec_forward(train as given there) + bsarec_loss on the same plan. */
int bsarec_backward(bsarec_plan_t *plan, void *stream);

/* Backward of BSARecModel.forward itself (src/model/bsarec.py:16-28 as an autograd graph): `d_out` = gradient w.r.t.
 * the LAST layer's output on ALL positions, fp32 [B, L, d].  Must follow bsarec_forward (not _forward_last) on the
 * same plan.  Gradients of all tensors -> `grads` (overwritten); the item table receives its lookup-path rows only
 * (there is no logits product on this path).  What sibling models with their own heads on the sequence output need
 * (DuoRec's contrastive terms, src/model/duorec.py:95-127). */
int bsarec_backward_seq(bsarec_plan_t *plan, const float *d_out, void *stream);

/* The same with an upstream gradient for EVERY element of forward(all_sequence_output=True)'s list
 * (src/model/bsarec.py:46-54: index 0 = embedding output, index l = output of block l - 1, index N = the last layer):
 * d_outs[0..N], fp32 [B, L, d] each; d_outs[N] must be given, d_outs[l < N] may be NULL (no gradient for that output). */
int bsarec_backward_seq_multi(bsarec_plan_t *plan, const float *const *d_outs, void *stream);

/* torch.optim.Adam (src/trainers.py:27-28,107) over flat arenas: one struct for every entry point that updates. */
typedef struct {
    float *params;            /* [n] fp32 master parameters (n % 4 == 0) */
    const float *grads;       /* [n] gradients (written by bsarec_backward through the plan's `grads` pointers) */
    float *exp_avg, *exp_avg_sq;   /* [n] Adam moments */
    long n;
    float lr, beta1, beta2, eps, weight_decay;
    float grad_scale;         /* g is scaled by this first (1/world_size after a summing all-reduce; else 1) */
    void *shadow_bf16;        /* null, or bf16[n] mirror of params (cfg.storage = 1): the update also writes the rounded */
    long shadow_from;         /*   parameter to shadow_bf16[i] for i >= shadow_from (the tensors after the item table) */
    /* ---- data-parallel gradient sources (all optional; zero-filled = g is `grads`) ---- */
    float *grads2;            /* a second arena ADDED to the first grads2_n elements of g and zeroed by the update: the   */
    long grads2_n;            /*   lookup-path rows of the item table when its dense part was exchanged early (buckets)    */
    int n_grad_srcs;          /* > 0: g = sum of grad_srcs[0 .. n) in index order (`grads` ignored) -- the one-shot       */
    const float *grad_srcs[8];/*   peer-to-peer exchange of bsarec_comm.h: every rank's arena in RANK order, so that every */
                              /*   replica forms the identical sum                                                        */
} bsarec_adam_t;

/* Advance Adam's t / bias corrections in `state`, then update. */
int bsarec_adam_step(const bsarec_adam_t *adam, void *state, void *stream);
/* The parameter update alone (t and the bias corrections were already advanced by bsarec_grad_step_indexed(lr > 0)). */
int bsarec_adam_apply(const bsarec_adam_t *adam, void *state, void *stream);

/* Trainer.iteration's per-batch body (src/trainers.py:100-107) in one call:
 * step_begin + forward(train) + loss + backward + Adam. */
int bsarec_train_step(bsarec_plan_t *plan, const int64_t *ids, const int64_t *answers, const bsarec_adam_t *adam,
                      void *stream);

/* Device-side batch assembly from a resident sample table (replaces RandomSampler + DataLoader collation,
 * src/dataset.py:207-211): ids_out[b,:] = table[perm[*cursor + b], :], answers_out[b] = answers_table[perm[*cursor + b]].
 * `cursor` is one int64 on the device. */
int bsarec_gather_batch(const int64_t *table, const int64_t *answers_table, const int64_t *perm, long n_samples,
                        const void *cursor, int B, int L, int64_t *ids_out, int64_t *answers_out, void *stream);

/* bsarec_train_step fed from the resident table: the embedding kernel assembles the batch at *cursor (and fills
 * ids_buf / answers_buf), forward + loss + backward + Adam; the closing block of the gradient reduction ends the step
 * (mean loss, forward-step index += 1, *cursor += B).  A captured graph of this call replays a whole epoch with no
 * host work.  The dropout step counter is used as it stands and advanced at the END (bsarec_train_step /
 * bsarec_step_begin advance it BEFORE use): call bsarec_step_begin once between a begin-style step and this one. */
int bsarec_train_step_indexed(bsarec_plan_t *plan, const int64_t *table, const int64_t *answers_table,
                              const int64_t *perm, long n_samples, void *cursor, int64_t *ids_buf, int64_t *answers_buf,
                              const bsarec_adam_t *adam, void *stream);

/* The data-parallel half of the above: gather + forward + loss + backward (no Adam).  The caller then exchanges the
 * flat gradient arena (RCCL all-reduce, or the peer-to-peer read of bsarec_comm.h) and calls
 * bsarec_adam_step(grad_scale = 1/world) -- or, when lr > 0 is given here, the step's closing block also advances
 * Adam's t and publishes the bias corrections, and the caller follows the exchange with bsarec_adam_apply (one launch
 * fewer per step).  lr <= 0: no Adam bookkeeping here. */
int bsarec_grad_step_indexed(bsarec_plan_t *plan, const int64_t *table, const int64_t *answers_table,
                             const int64_t *perm, long n_samples, void *cursor, int64_t *ids_buf, int64_t *answers_buf,
                             float lr, float beta1, float beta2, void *stream);

/* Evaluation branch of Trainer.iteration (src/trainers.py:134): scores[b][indices[j]] = 0 (not -inf) for every item j
 * of the CSR row indptr[users[b]] .. indptr[users[b] + 1] -- the items user b has already seen (the reference's
 * train_matrix, src/dataset.py:126-168, uploaded as int64 CSR).  scores: [B][ld] on the device. */
int bsarec_mask_seen(float *scores, long ld, int B, const int64_t *users, const int64_t *indptr,
                     const int64_t *indices, void *stream);

/* The same masking AND the reference's top-k (src/trainers.py:134-149: argpartition of the 20 best + argsort) in one
 * launch, one workgroup per user: scores[b][seen items] = 0, then out_idx[b][0..k) = item ids of the k largest scores of
 * scores[b][0..V) in descending order (equal scores: smaller id first), out_val (nullable) their scores.  indptr == NULL:
 * no masking.  k <= 24, k <= V <= ld. */
int bsarec_topk_seen(float *scores, long ld, int B, int V, const int64_t *users, const int64_t *indptr,
                     const int64_t *indices, int k, int64_t *out_idx, float *out_val, void *stream);

/* Stand-alone FrequencyLayer (src/model/bsarec.py:90-104) for per-op parity tests:
 * y = LN(Drop(low + beta^2 (x - low)) + x); backward given dy. */
int bsarec_freq_layer_fwd(const float *x, const float *sqrt_beta, const float *ln_w, const float *ln_b,
                          const float *twiddle, int B, int L, int d, int cutoff_bins, float eps, float p_drop,
                          const void *state, int site, float *y, float *xhat, float *rstd, void *stream);
long bsarec_freq_layer_bwd_scratch_floats(int B, int L, int d);
int bsarec_freq_layer_bwd(const float *x, const float *dy, const float *xhat, const float *rstd,
                          const float *sqrt_beta, const float *ln_w, const float *twiddle, int B, int L, int d,
                          int cutoff_bins, float p_drop, const void *state, int site,
                          float *scratch /* bsarec_freq_layer_bwd_scratch_floats(B,L,d) floats */,
                          float *dx, float *dsqrt_beta, float *dln_w, float *dln_b, void *stream);

/* Tag kernel launches of one id with hipEvents for in-process roofline timing (bench.py):
 * when set, every launch of kernel class `kclass` (see BSAREC_K_*) on the next calls of THIS plan is bracketed
 * by hipEventRecord on the launch stream; bsarec_profile_read returns the summed milliseconds and
 * launch count, then resets.  Not for use under graph capture. */
enum { BSAREC_K_NONE = 0, BSAREC_K_FFN1 = 1, BSAREC_K_FFN2 = 2, BSAREC_K_QKV = 3, BSAREC_K_LOGITS = 4,
       BSAREC_K_DU = 5, BSAREC_K_DW1 = 6, BSAREC_K_FUSED_FWD = 7, BSAREC_K_FUSED_BWD = 8 };
int bsarec_profile_select(bsarec_plan_t *plan, int kclass);
/* Diagnostic: device buffer of 32*2*layers int64 that receives per-phase shader-clock stamps of workgroup 0
 * of this plan's fused kernels (null disables). */
int bsarec_debug_stamps(bsarec_plan_t *plan, void *dev_buf);
/* Waits for the plan's recorded event pairs (the one call besides bsarec_plan_create that blocks the host). */
int bsarec_profile_read(bsarec_plan_t *plan, double *ms_total, int *launches);
/* Milliseconds one such hipEvent bracket reads with NO kernel inside it (average of `reps` back-to-back pairs on
 * `stream`): the marker-packet cost bench.py subtracts so that its per-launch time agrees with rocprofv3's. */
int bsarec_profile_event_overhead(void *stream, int reps, double *ms_avg);

#ifdef __cplusplus
}
#endif
#endif /* BSAREC_HIP_H */
