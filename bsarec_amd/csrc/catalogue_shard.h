// Catalogue-sharded head of the training step (the C5 variant; new functionality -- the reference is
// single-device, src/main.py:19).  Rank r owns rows [r*rows_per, (r+1)*rows_per) of the item table (and their Adam
// moments); the encoder stays a data-parallel replica.  What the reference computes in three lines
// (src/model/bsarec.py:32-35: seq_output[:, -1] @ item_emb^T, CrossEntropyLoss) becomes, per rank:
//   lookup rows read straight out of the owners' shards over xGMI          shard_gather_rows_kernel
//   partial logits of ALL ranks' sequences against the owned rows          (gemm_kernel NT)
//   per-row (max, sum exp, target logit) of the owned slice                shard_ce_stats_kernel
//   combined with every rank's statistics -> lse, loss, d loss / d logits  shard_ce_grad_kernel
//   dE of the owned rows (no all-reduce of the table gradient at all)      (gemm_kernel TN)
//   partial d h_last of all sequences, split-K over the owned rows         (gemm_kernel NN) + shard_slab_sum_kernel
//   lookup-path gradient rows pulled from every rank's staging table       shard_scatter_rows_kernel
#pragma once
#include "kernels.h"

struct ShardPtrs { const float* p[8]; };

// Staging table of one rank's batch: row 0 = E[0] (the padding id keeps ITS row: init_weights overwrites nn.Embedding's
// zero padding row, src/model/_abstract_model.py:27-38), row j+1 = E[ids[j]].  local_ids[j] = j+1, or 0 for padding, so
// that the encoder plan (item_size = n+1) reads the staging table with its ordinary embedding front-end.  Peers' rows
// are read with system-scope loads: their Adam rewrote them during the previous step.
__global__ void __launch_bounds__(256)
shard_gather_rows_kernel(const int64_t* __restrict__ ids, long n, const ShardPtrs S, long rows_per, long V, int d4,
                         float* __restrict__ stage, int64_t* __restrict__ local_ids) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = idx / d4;
    const int c = (int)(idx - row * d4);
    if (row > n) return;
    long id = row == 0 ? 0 : (long)ids[row - 1];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    if (row > 0 && c == 0) local_ids[row - 1] = id > 0 ? row : 0;
    if (row > 0 && id == 0) return;                         // padding tokens read staging row 0
    const long owner = id / rows_per;
    const float* src = S.p[owner] + ((id - owner * rows_per) * d4 + c) * 4;
    st4(stage + (row * d4 + c) * 4, ld4_sys(src));
}

// Owner side of the lookup-path gradient: every rank's staging-table gradient holds one row per token (row j+1 =
// d loss / d embedding of token j); the owner of item ids[j] adds it to its dE row.  Padding tokens carry no gradient
// (padding_idx = 0).  Hot items are hit by many tokens: float atomics at the memory side, rows in no fixed order.
__global__ void __launch_bounds__(256)
shard_scatter_rows_kernel(const int64_t* __restrict__ ids_all, long n, int world, const ShardPtrs G, long lo, long Vs,
                          long V, int d4, float* __restrict__ dE) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long t = idx / d4;
    const int c = (int)(idx - t * d4);
    if (t >= n * world) return;
    long id = (long)ids_all[t];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    if (id == 0 || id < lo || id >= lo + Vs) return;
    const int r = (int)(t / n);
    const long j = t - (long)r * n;
    const f32x4 g = ld4_sys(G.p[r] + ((j + 1) * d4 + c) * 4);
    float* dst = dE + ((id - lo) * d4 + c) * 4;
    unsafeAtomicAdd(dst + 0, g.x); unsafeAtomicAdd(dst + 1, g.y); unsafeAtomicAdd(dst + 2, g.z); unsafeAtomicAdd(dst + 3, g.w);
}

// (max, sum exp(x - max), target logit) of one sequence's row over the owned catalogue slice: one pass, online
// rescaling.  stats = [3][Bg].  An empty slice reports (-inf, 0, 0).
__global__ void __launch_bounds__(ROW_THREADS)
shard_ce_stats_kernel(const float* __restrict__ logits, long ld, int Vs, const int64_t* __restrict__ answers, long lo, long V,
                      float* __restrict__ stats, int Bg) {
    __shared__ float red_m[ROW_THREADS / 64], red_s[ROW_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (long)b * ld;
    float m = -INFINITY, s = 0.f;
    const int V4 = Vs & ~3;
    for (int v = tid * 4; v < V4; v += ROW_THREADS * 4) {
        const f32x4 x = ld4(row + v);
        const float mx = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
        if (mx > m) { s *= expf(m - mx); m = mx; }
        s += expf(x.x - m) + expf(x.y - m) + expf(x.z - m) + expf(x.w - m);
    }
    for (int v = V4 + tid; v < Vs; v += ROW_THREADS) {
        const float x = row[v];
        if (x > m) { s *= expf(m - x); m = x; }
        s += expf(x - m);
    }
    const float wm = group_max<64>(m);
    s = group_sum<64>(m == -INFINITY ? 0.f : s * expf(m - wm));
    if ((tid & 63) == 0) { red_m[tid >> 6] = wm; red_s[tid >> 6] = s; }
    __syncthreads();
    if (tid == 0) {
        float M = red_m[0];
        for (int i = 1; i < ROW_THREADS / 64; ++i) M = fmaxf(M, red_m[i]);
        float S = 0.f;
        for (int i = 0; i < ROW_THREADS / 64; ++i) if (red_m[i] != -INFINITY) S += red_s[i] * expf(red_m[i] - M);
        long ans = (long)answers[b];
        ans = ans < 0 ? 0 : (ans >= V ? V - 1 : ans);
        stats[b] = M; stats[Bg + b] = S;
        stats[2 * Bg + b] = (ans >= lo && ans < lo + Vs) ? row[ans - lo] : 0.f;
    }
}

// Every rank's statistics combined (rank order, the same arithmetic on every rank) -> lse and the row's loss; the
// owned slice of the logits row becomes d loss / d logits = (softmax - onehot) / Bg in place; pad columns -> 0.
__global__ void __launch_bounds__(ROW_THREADS)
shard_ce_grad_kernel(float* __restrict__ logits, long ld, int Vs, const int64_t* __restrict__ answers, long lo, long V,
                     const float* __restrict__ stats_all, int world, int Bg, float inv_bg, float* __restrict__ loss_rows) {
    const int b = blockIdx.x, tid = threadIdx.x;
    float M = -INFINITY;
    for (int r = 0; r < world; ++r) M = fmaxf(M, stats_all[((long)r * 3 + 0) * Bg + b]);
    float S = 0.f, tgt = 0.f;
    for (int r = 0; r < world; ++r) {
        const float mr = stats_all[((long)r * 3 + 0) * Bg + b];
        if (mr != -INFINITY) S += stats_all[((long)r * 3 + 1) * Bg + b] * expf(mr - M);
        tgt += stats_all[((long)r * 3 + 2) * Bg + b];
    }
    const float lse = M + logf(S);
    long ans = (long)answers[b];
    ans = ans < 0 ? 0 : (ans >= V ? V - 1 : ans);
    const long a_loc = ans - lo;
    float* row = logits + (long)b * ld;
    for (int v = tid * 4; v < (int)ld; v += ROW_THREADS * 4) {
        f32x4 x = ld4(row + v);
        float* e = reinterpret_cast<float*>(&x);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            e[k] = (v + k < Vs) ? (expf(e[k] - lse) - ((long)(v + k) == a_loc ? 1.0f : 0.0f)) * inv_bg : 0.f;
        st4(row + v, x);
    }
    if (tid == 0) loss_rows[b] = lse - tgt;
}

// sum of split-K slabs [nsplit][n] -> out[n]
__global__ void __launch_bounds__(ROW_THREADS)
shard_slab_sum_kernel(const float* __restrict__ slabs, int nsplit, long n4, float* __restrict__ out) {
    const long i = (long)blockIdx.x * ROW_THREADS + threadIdx.x;
    if (i >= n4) return;
    f32x4 a = ld4(slabs + 4 * i);
    for (int s = 1; s < nsplit; ++s) a += ld4(slabs + ((long)s * n4 + i) * 4);
    st4(out + 4 * i, a);
}
