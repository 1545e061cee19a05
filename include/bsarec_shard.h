/* bsarec_shard.h -- catalogue-sharded head of the BSARec training step (configuration C5: V = 10 M items).
 *
 * New functionality: the reference is single-device (src/main.py:19); what is sharded here is its full-catalogue head
 *     logits = seq_output[:, -1, :] @ item_embeddings.weight^T ;  loss = CrossEntropyLoss(logits, answers)
 * (src/model/bsarec.py:32-35) and the item-embedding lookup (src/model/_abstract_model.py:14-24).  At C5 a replicated
 * table costs 41 GB per GPU (weights + gradient + Adam moments) and a dense 10.24 GB gradient all-reduce per step; with
 * the catalogue rows sharded W ways (rank r owns rows [r*rows_per, (r+1)*rows_per), their gradient and their moments)
 * the table gradient never crosses a link.  The encoder (everything but the item table) stays a data-parallel
 * replica.  One step, per rank (host side: bsarec_amd/catalogue.py):
 *
 *   bsarec_shard_gather_rows      lookup rows of the local batch, read out of the owners' shards (IPC-mapped, xGMI)
 *   bsarec_forward                the ordinary encoder plan over the staging table (item_size = B*L + 1)
 *   all-gather h_last, answers    [Bg, d] + [Bg]  (Bg = W*B; torch.distributed)
 *   bsarec_shard_logits           partial logits of all Bg sequences against the owned rows
 *   bsarec_shard_ce_stats         per-row (max, sum exp, target logit) of the owned slice
 *   all-gather stats              [W, 3, Bg]
 *   bsarec_shard_ce_grad          lse / loss from everybody's statistics; d loss / d logits of the owned slice
 *   bsarec_shard_head_bwd         dE of the owned rows (complete, local) + partial d h_last of all Bg sequences
 *   reduce-scatter d h_last       [Bg, d] -> [B, d]
 *   bsarec_backward_seq           encoder backward; the staging table's gradient = one row per token
 *   bsarec_comm_barrier           (bsarec_comm.h)
 *   bsarec_shard_scatter_rows     owners pull the token rows of their items out of every rank's staging gradient
 *   bsarec_adam_step / _apply     encoder: sum of every rank's gradient arena (grad_srcs); shard: local dE
 *
 * Same conventions as bsarec_hip.h: plain pointers and sizes, caller-owned memory, caller's stream, no synchronisation,
 * return 0 / <0 invalid argument / >0 hipError_t.  fp32.  hidden % 4 == 0, world <= 8.
 */
#ifndef BSAREC_SHARD_H
#define BSAREC_SHARD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Staging table of one rank's batch.  ids[n] are catalogue ids of the n = B*L tokens; shards[r] is rank r's table
 * shard [rows_per_shard, hidden] (device pointers valid in THIS process: the local shard and IPC mappings of the
 * peers').  stage[(n+1), hidden]: row 0 = E[0], row j+1 = E[ids[j]]; local_ids[j] = j+1, or 0 where ids[j] == 0.
 * Ids outside [0, item_size) are clamped like everywhere else in the library. */
int bsarec_shard_gather_rows(const int64_t *ids, long n, const float *const *shards, int world, long rows_per_shard,
                             long item_size, int hidden, float *stage, int64_t *local_ids, void *stream);

/* logits[b, v] = h[b, :] . E_shard[v, :] for b < Bg, v < Vs; row stride ld >= Vs, ld % 4 == 0.  h has row stride ldh. */
int bsarec_shard_logits(const float *h, long ldh, int Bg, const float *E_shard, int Vs, int hidden, float *logits, long ld,
                        void *stream);

/* stats[3][Bg] = per row: max over the owned slice, sum exp(x - max), logits[b, answers[b] - lo] if this rank owns the
 * answer else 0.  (An empty slice reports -inf, 0, 0.) */
int bsarec_shard_ce_stats(const float *logits, long ld, int Bg, int Vs, const int64_t *answers, long lo, long item_size,
                          float *stats, void *stream);

/* stats_all[world][3][Bg] (every rank's stats, rank order) -> loss_rows[Bg] = lse - target logit, loss[0] = their mean
 * (identical on every rank), and logits := (softmax - onehot) / Bg in place over the owned slice (columns Vs..ld-1 := 0). */
int bsarec_shard_ce_grad(float *logits, long ld, int Bg, int Vs, const int64_t *answers, long lo, long item_size,
                         const float *stats_all, int world, float *loss_rows, float *loss, void *stream);

/* dE_shard[Vs, hidden] = dlogits^T . h (overwritten: the dense part of the owned rows' gradient, complete);
 * dh[Bg, hidden] = dlogits . E_shard (this rank's partial sum over its rows; split-K slabs in `scratch`). */
long bsarec_shard_head_bwd_scratch_floats(int Bg, int Vs, int hidden);
int bsarec_shard_head_bwd(const float *dlogits, long ld, int Bg, int Vs, const float *h, long ldh, const float *E_shard,
                          int hidden, float *dE_shard, float *dh, float *scratch, void *stream);

/* Lookup-path gradient: ids_all[world][n] (every rank's token ids, rank order), stage_grads[r] = rank r's staging-table
 * gradient [(n+1), hidden] (row j+1 = token j; IPC mappings for the peers).  dE_shard[id - lo] += row for every token
 * whose id is owned (lo <= id < lo + Vs) and not the padding id.  Float atomics: the order of additions is not fixed. */
int bsarec_shard_scatter_rows(const int64_t *ids_all, long n, int world, const float *const *stage_grads, long lo, long Vs,
                              long item_size, int hidden, float *dE_shard, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BSAREC_SHARD_H */
