// MI355X (gfx950) BSARec training hot path: launch plan + C ABI.  See include/bsarec_hip.h.
#include "../../include/bsarec_hip.h"
#include "../../include/bsarec_shard.h"
#include "epilogues.h"
#include "kernels.h"
#include "fused_layer.h"
#include "dw_direct.h"
#include "fused_top.h"
#include "fused_chain.h"
#include "comm.h"
#include "catalogue_shard.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)
#define RET(x) do { int r_ = (x); if (r_ != 0) return r_; } while (0)

// Dry run: walk the launch sequence, set per-kernel attributes (large dynamic LDS), launch nothing.
// bsarec_plan_create does one dry pass so that the first real pass may already be under graph capture.
static thread_local bool g_dry = false;
#define LAUNCH(...) do { if (!g_dry) hipLaunchKernelGGL(__VA_ARGS__); } while (0)

// Every option lives in bsarec_config_t and belongs to the plan (no process-wide knobs).  Defaults of the 0 values:
//   top_slabs 2  (slab slices of the pruned top block's weight-gradient products, K = B or B*h rows only; measured
//                 1/2/4/8 slabs: 0.2115 / 0.2095 / 0.2122 / 0.2130 ms per step)
//   splits    32 at the fused shape, 40 elsewhere (slab slices of the full-block weight-gradient products)
struct bsarec_plan;
static thread_local bsarec_plan* t_plan = nullptr;   // plan of the C call this thread is inside (ProfScope, stamps)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline long rup(long a, long b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------------------------
// in-process kernel timing (bench.py roofline): hipEvent pairs around one kernel class
// ---------------------------------------------------------------------------------------------
struct ProfState {                    // per plan (bsarec_profile_select / _read)
    int kclass = BSAREC_K_NONE;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t used = 0;
    ~ProfState() { for (auto& e : events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); } }
};
static ProfState* prof_of(bsarec_plan* p);
static bool bf_products_of(const bsarec_plan* p);     // cfg.storage = 1 on the generic tiled path: bf16 products (gemm.h)

struct ProfScope {
    hipStream_t s; bool on; hipEvent_t stop;
    ProfScope(int kclass, hipStream_t st) : s(st), on(false) {
        ProfState* ps = t_plan ? prof_of(t_plan) : nullptr;
        on = ps && kclass != BSAREC_K_NONE && kclass == ps->kclass;
        if (!on) return;
        if (ps->used == ps->events.size()) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            ps->events.push_back({a, b});
        }
        (void)hipEventRecord(ps->events[ps->used].first, s);
        stop = ps->events[ps->used].second;
        ++ps->used;
    }
    ~ProfScope() { if (on) (void)hipEventRecord(stop, s); }
};

// ---------------------------------------------------------------------------------------------
// GEMM launcher
// ---------------------------------------------------------------------------------------------
static GemmP gemm_defaults(int M, int N, int K) {
    GemmP P;
    memset(&P, 0, sizeof(P));
    P.M = M; P.N = N; P.K = K; P.Nb = N; P.Kv = K;
    P.nseg = 1; P.nprob = 1; P.nsplit = 1; P.kchunk = K; P.nh = 1;
    return P;
}

template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int AXF, int BXF, bool BG, bool BF, class Epi>
static int launch_gemm_as(const GemmP& P, const XformP& X, const Epi& epi, float* bgrad, int nbatch, hipStream_t s, int kclass) {
    auto kern = gemm_kernel<BM, BN, WM, WN, AKM, BKM, AXF, BXF, BG, BF, Epi>;
    constexpr size_t smem = GemmSmem<BM, BN, AKM, BKM, BF>::BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        if (smem > 48 * 1024)
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_done = true;
    }
    if (P.M <= 0 || P.N <= 0) return 0;
    dim3 grid(cdiv(P.M, BM), cdiv(P.N, BN), nbatch * P.nprob * P.nsplit);
    if (g_dry) return 0;
    ProfScope prof(kclass, s);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), smem, s, P, X, epi, bgrad);
    return (int)hipGetLastError();
}

// fp32_only: the products of the loss head (logits and their backward) stay fp32 in the bf16-product mode, as they do under
// the fused bf16 storage
template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int AXF, int BXF, bool BG, class Epi>
static int launch_gemm(const GemmP& P, const XformP& X, const Epi& epi, float* bgrad, int nbatch, hipStream_t s,
                       int kclass = BSAREC_K_NONE, bool fp32_only = false) {
    if (!fp32_only && t_plan && bf_products_of(t_plan)) {
        // token-parallel products (M = B L rows, one problem): a 64 x 64 tile issues ~120 instructions and 16 KB of fp32 operand
        // loads per 0.26 MFLOP k-step, which is what bounds it once the matrix time is gone -- 128 x 128 tiles quarter that
        if constexpr (BM == 64 && BN == 64 && !BG)
            if (P.M >= 8192 && P.N >= 128 && nbatch == 1)
                return launch_gemm_as<128, 128, WM, WN, AKM, BKM, AXF, BXF, BG, true, Epi>(P, X, epi, bgrad, nbatch, s, kclass);
        return launch_gemm_as<BM, BN, WM, WN, AKM, BKM, AXF, BXF, BG, true, Epi>(P, X, epi, bgrad, nbatch, s, kclass);
    }
    // fp32, 256-wide tiles (LayerNorm / softmax / dS epilogues at hidden or L > 128): 8 waves as 2 x 4 -- the 92 KB of
    // fp32 staging allow one workgroup per CU, and four waves at 344 registers left every SIMD with a single wave
    if constexpr (BN == 256 && WM * WN == 4)
        return launch_gemm_as<BM, BN, 2, 4, AKM, BKM, AXF, BXF, BG, false, Epi>(P, X, epi, bgrad, nbatch, s, kclass);
    else
        return launch_gemm_as<BM, BN, WM, WN, AKM, BKM, AXF, BXF, BG, false, Epi>(P, X, epi, bgrad, nbatch, s, kclass);
}

static XformP no_xform() { XformP X; memset(&X, 0, sizeof(X)); return X; }

template <bool BIAS, bool ADD, bool GGRAD>
static EpiLinear<BIAS, ADD, GGRAD> epi_linear(float* C, long ldc) {
    EpiLinear<BIAS, ADD, GGRAD> e;
    memset(&e, 0, sizeof(e));
    e.C[0] = C; e.ldc = ldc;
    return e;
}

// ---------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------
struct LayerBufs {
    float *dsp, *xhat_f, *rstd_f, *q, *k, *v, *probs, *ctx, *xhat_a, *rstd_a, *hmix, *u, *xhat_ff, *rstd_ff;
    float* gp;       // fused path: gelu'(pre-activation) [T, 4d]; there `u` holds gelu(pre-activation) (one-row top block: u itself)
};

struct bsarec_plan {
    bsarec_config_t cfg;
    bsarec_tensors_t P, G, S;                  // parameters, gradients, bf16 shadow of the parameters (storage = 1)
    bool bf;                                   // cfg.storage == 1 at the fused shape: bf16 storage + bf16 MFMA in the block kernels
    bool bf_products;                          // cfg.storage == 1 elsewhere: fp32 tensors, bf16 products in the tiled GEMMs (gemm.h)
    char* ws; size_t ws_bytes;
    uint64_t* state;
    const float* twiddle;
    int T, Lp, Vp, dh, nblk, rows_pb, nsplit, kchunk, vsplit, vchunk;
    bool fused;        // fused per-sequence block kernels (decided at plan creation)
    bool train;        // mode of the last forward
    // activations kept for backward
    int* ids32;
    float* X[BSAREC_MAX_LAYERS + 1];
    float *xhat0, *rstd0;
    LayerBufs lb[BSAREC_MAX_LAYERS];
    float *logits, *dlogits, *loss_rows, *loss;
    // backward scratch (shared by all layers)
    float *dXa, *dXb, *dz, *dT, *dU, *dH, *dXacc, *dO, *dF, *dC, *dS, *dq, *dk, *dv, *dXtmp, *dlast_slab;
    float *slab_wL[BSAREC_MAX_LAYERS], *slab_bL[BSAREC_MAX_LAYERS], *part_lnL[BSAREC_MAX_LAYERS], *part_betaL[BSAREC_MAX_LAYERS];
    float *slab_w, *slab_b, *part_ln, *part_beta;   // the current layer's set (selected by the backward loop)
    float *part_ln0, *part_pos, *trash;
    int pos_slices;
    ReduceJob* jobs; int jobs_per_layer;
    ReduceJob* jobs_pruned;                    // same table with the top layer's key / value bias jobs fed from partials
    bool prune_ok;                             // the loss path may run the pruned top block (fused shape, >= 2 layers)
    bool pruned;                               // mode of the last forward
    int loss_kind;                             // head of the last loss call: 0 = full-catalogue CE, 1 = SASRec's BCE pair
    const float* ext_dy = nullptr;             // bsarec_backward_seq: upstream gradient of the last layer's output, all positions
    const float* ext_mid[BSAREC_MAX_LAYERS] = {};   // bsarec_backward_seq_multi: upstream gradients of layer outputs 0 .. N-1 (null: none)
    const int64_t *bce_pos, *bce_neg;
    float *part_kvb, *slab_dummy;
    float* part_cwL[BSAREC_MAX_LAYERS];        // FMLPRec: per-sequence d(complex_weight) [B][cb][d][2]
    float *top_dq, *top_dO, *top_dT, *top_dU, *top_ak, *top_rk, *top_av, *top_rv;              // [2][B][d] key / value bias partials of the pruned top block; [nsplit][4d] sink
    int* blockmap; int red_blocks;           // flat block -> (job, chunk) table of the final gradient reduction
    long red_elems; const float *red_lo, *red_hi;   // what the reduction jobs write: element count, address range
    // options resolved from cfg (0 = default there)
    int top_slabs; bool embed_in_block, direct_dw;
    bool scatter_in_block;                     // fused path: the embedding-gradient scatter rides in block 0's weight-gradient launch
    ProfState prof;
    long long* stamps = nullptr;             // diagnostic stamp buffer (bsarec_debug_stamps)
    bsarec_hook_t dense_hook = nullptr; void* dense_hook_user = nullptr;   // bsarec_plan_set_dense_grad_hook
    float* lookup_grad = nullptr;            // target of the embedding scatter when the dense dE is exchanged early
};
static ProfState* prof_of(bsarec_plan* p) { return &p->prof; }
static bool bf_products_of(const bsarec_plan* p) { return p->bf_products; }
struct PlanScope {                           // marks the plan a C call works on for this thread (nesting-safe)
    bsarec_plan* prev;
    explicit PlanScope(bsarec_plan* p) : prev(t_plan) { t_plan = p; }
    ~PlanScope() { t_plan = prev; }
};

struct Carver {
    char* base; size_t off;
    explicit Carver(char* b) : base(b), off(0) {}
    template <class T> T* take(size_t n) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += rup((long)(n * sizeof(T)), 256);
        return p;
    }
};

static int check_cfg(const bsarec_config_t& c) {
    if (c.batch < 1 || c.seq_len < 1 || c.seq_len > 256) return -1;
    if (c.hidden < 4 || c.hidden > 256 || c.hidden % 4) return -2;
    if (c.heads < 1 || c.hidden % c.heads || (c.hidden / c.heads) % 4) return -3;
    if (c.layers < 1 || c.layers > BSAREC_MAX_LAYERS) return -4;
    if (c.item_size < 2) return -5;
    if (c.cutoff_bins < 1 || c.cutoff_bins > c.seq_len / 2 + 1) return -6;
    if ((long)c.cutoff_bins * c.hidden > 8192) return -7;      // spectrum must fit the LDS carve
    if (c.p_hidden < 0.f || c.p_hidden >= 1.f || c.p_attn < 0.f || c.p_attn >= 1.f) return -8;
    if (c.filter_kind != 0 && c.filter_kind != 1) return -9;
    if (c.filter_kind == 1 && c.cutoff_bins != c.seq_len / 2 + 1) return -9;     // the learnable filter has every rFFT bin
    if (c.hidden_act < 0 || c.hidden_act > 4) return -14;
    if (c.storage < 0 || c.storage > 1) return -15;
    if (c.splits < 0 || c.splits > 1024 || c.top_slabs < 0 || c.top_slabs > 16) return -16;
    return 0;
}

static bool fused_shape_ok(const bsarec_config_t& c) {
    const int dh = c.hidden / c.heads;
    if (c.no_fused || c.hidden_act != 0 || c.hidden != 64 || c.seq_len > 64 || !(dh == 16 || dh == 32 || dh == 64)) return false;
    // filter_kind 1 (the sibling model FMLPRec: learnable complex filter over all L/2 + 1 bins, no attention branch) has its own
    // instantiation of the block kernels, fp32 storage only
    if (c.filter_kind == 1) return c.storage == 0 && c.alpha == 1.0f && c.cutoff_bins <= 36;
    return c.cutoff_bins <= FUSED_MAX_CB;
}

static void derive(bsarec_plan& p) {
    const bsarec_config_t& c = p.cfg;
    p.fused = fused_shape_ok(c);
    p.bf = c.storage == 1 && p.fused;
    p.bf_products = c.storage == 1 && !p.fused;
    p.T = c.batch * c.seq_len;
    p.Lp = (int)rup(c.seq_len, 4);
    p.Vp = (int)rup(c.item_size, 4);
    p.dh = c.hidden / c.heads;
    // LayerNorm gamma/beta partials: one row per 64-token block, or one per sequence on the fused path
    p.nblk = p.fused ? c.batch : cdiv(p.T, 64);
    p.rows_pb = p.fused ? c.seq_len : 64;
    // split-K over tokens for the weight-gradient products: 32 / 40 slab slices (below), 32-aligned chunks (the direct kernel of
    // the fused shape cuts every slice into 4 more quarters inside a workgroup)
    // fused shape, C1 (12 weight-gradient units): 32 slices = 384 workgroups leave 128 of the kernel's 512 slots to the
    // embedding scatter blocks from the first cycle (measured with exact k-block counts, 24 / 28 / 32 / 40 slices:
    // 0.1722 / 0.1706 / 0.1694 / 0.1745 ms per step)
    const int want_splits = c.splits > 0 ? c.splits : (p.fused ? 32 : 40);
    p.top_slabs = c.top_slabs > 0 ? c.top_slabs : 2;
    p.embed_in_block = !c.separate_embed || p.bf;     // bf16 storage: X[0] is written by the block kernel only
    p.direct_dw = !c.dw_tiled && (long)p.T * 4 * c.hidden * 4 < (1L << 31);      // its operands sit behind 32-bit buffer offsets
    p.scatter_in_block = p.fused && p.direct_dw && !c.separate_embed;
    // (bf16 storage: 64-aligned chunks -- a wave's quarter is then whole 16-row k-blocks of the bf16 matrix instruction)
    long ch = rup(cdiv(p.T, want_splits), p.bf ? 2 * GEMM_BK : GEMM_BK);
    if (ch < 64) ch = 64;
    if (ch > 2048) ch = 2048;
    p.kchunk = (int)ch;
    p.nsplit = cdiv(p.T, p.kchunk);
    // split-K over the catalogue for d(h_last) = dlogits . E
    long vc = rup(cdiv(p.Vp, p.fused ? 32 : 16), GEMM_BK);     // 32 catalogue slices for the direct kernel of the fused shape
    if (vc < 64) vc = 64;
    p.vchunk = (int)vc;
    p.vsplit = cdiv(p.Vp, p.vchunk);
}

static void carve(bsarec_plan& p, char* base, size_t* total) {
    const bsarec_config_t& c = p.cfg;
    const long T = p.T, d = c.hidden, B = c.batch, L = c.seq_len, h = c.heads, N = c.layers;
    const long Td = T * d;
    Carver cv(base);
    p.jobs = cv.take<ReduceJob>((size_t)(N * 20 + 3));
    p.jobs_pruned = cv.take<ReduceJob>((size_t)(N * 20 + 3));
    p.blockmap = cv.take<int>((size_t)(N * (12 * cdiv(d * d, 64) + 16 * cdiv(4 * d, 64) + cdiv((long)c.cutoff_bins * d * 2, 64)) +
                                       cdiv(L * d, 64) + 2 * cdiv(d, 64) + 64));
    p.ids32 = cv.take<int>(T);
    for (int l = 0; l <= N; ++l) p.X[l] = cv.take<float>(Td);
    p.xhat0 = cv.take<float>(Td); p.rstd0 = cv.take<float>(T);
    for (int l = 0; l < N; ++l) {
        LayerBufs& b = p.lb[l];
        b.dsp = cv.take<float>(Td); b.xhat_f = cv.take<float>(Td); b.rstd_f = cv.take<float>(T);
        b.q = cv.take<float>(Td); b.k = cv.take<float>(Td); b.v = cv.take<float>(Td);
        b.probs = cv.take<float>(B * h * L * p.Lp); b.ctx = cv.take<float>(Td);
        b.xhat_a = cv.take<float>(Td); b.rstd_a = cv.take<float>(T); b.hmix = cv.take<float>(Td);
        b.u = cv.take<float>(4 * Td); b.xhat_ff = cv.take<float>(Td); b.rstd_ff = cv.take<float>(T);
        b.gp = p.fused ? cv.take<float>(4 * Td) : nullptr;
    }
    p.logits = cv.take<float>(B * p.Vp); p.dlogits = cv.take<float>(B * p.Vp);
    p.loss_rows = cv.take<float>(B); p.loss = cv.take<float>(4);
    p.dXa = cv.take<float>(Td); p.dXb = cv.take<float>(Td); p.dz = cv.take<float>(Td); p.dT = cv.take<float>(Td);
    p.dU = cv.take<float>(4 * Td); p.dH = cv.take<float>(Td); p.dXacc = cv.take<float>(Td); p.dO = cv.take<float>(Td);
    p.dF = cv.take<float>(Td); p.dC = cv.take<float>(Td); p.dS = cv.take<float>(B * h * L * p.Lp);
    p.dq = cv.take<float>(Td); p.dk = cv.take<float>(Td); p.dv = cv.take<float>(Td); p.dXtmp = cv.take<float>(Td);
    p.dlast_slab = cv.take<float>((long)p.vsplit * B * d);
    for (int l = 0; l < N; ++l) {                                // per layer, so that ONE reduction launch ends the backward
        p.slab_wL[l] = cv.take<float>((long)p.nsplit * 12 * d * d);  // wq wk wv wo (d*d each) + w1 w2 (4 d*d each)
        p.slab_bL[l] = cv.take<float>((long)p.nsplit * 9 * d);       // bq bk bv bo (d) + b1 (4d) + b2 (d)
        p.part_lnL[l] = cv.take<float>((long)p.nblk * 6 * d);        // gamma/beta partials of the 3 LayerNorms
        p.part_betaL[l] = cv.take<float>(B * d);
    }
    p.slab_w = p.slab_wL[0]; p.slab_b = p.slab_bL[0]; p.part_ln = p.part_lnL[0]; p.part_beta = p.part_betaL[0];
    p.part_ln0 = cv.take<float>((long)p.nblk * 2 * d);
    p.pos_slices = cdiv(B, 64);
    p.part_pos = cv.take<float>((long)p.pos_slices * L * d);
    p.trash = cv.take<float>(1024);
    p.part_kvb = cv.take<float>(2 * B * d);
    for (int l = 0; l < N; ++l) p.part_cwL[l] = c.filter_kind == 1 ? cv.take<float>(B * c.cutoff_bins * d * 2) : nullptr;
    // pruned top block: its last-row gradient operands and the rank-1 key / value operands live in their own compact
    // buffers ([B][d], [B][4d], [B*h][d]) so that they survive the next block's backward and ride in ITS weight-gradient launch
    p.top_dq = cv.take<float>(B * d); p.top_dO = cv.take<float>(B * d); p.top_dT = cv.take<float>(B * d);
    p.top_dU = cv.take<float>(4 * B * d);
    p.top_ak = cv.take<float>(B * h * d); p.top_rk = cv.take<float>(B * h * d);
    p.top_av = cv.take<float>(B * h * d); p.top_rv = cv.take<float>(B * h * d);
    p.slab_dummy = cv.take<float>((long)p.nsplit * 4 * d);
    // (no guard pad: the direct weight-gradient kernels prefetch past a slice without predicates, but through buffer
    // descriptors sized to their operand -- dw_direct.h)
    *total = cv.off;
}

extern "C" int bsarec_abi_version(void) { return BSAREC_ABI_VERSION; }

extern "C" size_t bsarec_workspace_bytes(const bsarec_config_t* cfg) {
    if (!cfg || check_cfg(*cfg) != 0) return 0;
    bsarec_plan p;
    p.cfg = *cfg;
    derive(p);
    size_t total = 0;
    carve(p, nullptr, &total);
    return total;
}

// slab sub-offsets (floats) inside slab_w / slab_b, per split-K slice
struct SlabMap { long wq, wk, wv, wo, w1, w2, wtot, bq, bk, bv, bo, b1, b2, btot; };
static SlabMap slab_map(long d) {
    SlabMap m;
    m.wq = 0; m.wk = d * d; m.wv = 2 * d * d; m.wo = 3 * d * d; m.w1 = 4 * d * d; m.w2 = 8 * d * d; m.wtot = 12 * d * d;
    m.bq = 0; m.bk = d; m.bv = 2 * d; m.bo = 3 * d; m.b1 = 4 * d; m.b2 = 8 * d; m.btot = 9 * d;
    return m;
}

// Split-K slabs are stored [tensor][split][elements] so that one reduce job reads a fixed stride.
static float* slab_w_ptr(const bsarec_plan& p, long tensor_off) { return p.slab_w + tensor_off * p.nsplit; }
static float* slab_b_ptr(const bsarec_plan& p, long tensor_off) { return p.slab_b + tensor_off * p.nsplit; }

extern "C" int bsarec_plan_create(bsarec_plan_t** out, const bsarec_config_t* cfg, const bsarec_tensors_t* params,
                                  const bsarec_tensors_t* grads, const bsarec_tensors_t* shadow, void* workspace,
                                  size_t workspace_bytes, void* state, const float* twiddle, void* stream) {
    if (!out || !cfg || !params || !workspace || !state || !twiddle) return -10;
    RET(check_cfg(*cfg));
    if (cfg->storage == 1 && fused_shape_ok(*cfg) && !shadow) return -15;     // (the generic path rounds its operands itself)
    if (((uintptr_t)workspace & 255) != 0) return -11;
    bsarec_plan* p = new bsarec_plan();
    p->cfg = *cfg;
    p->P = *params;
    if (grads) p->G = *grads; else memset(&p->G, 0, sizeof(p->G));
    if (shadow) p->S = *shadow; else memset(&p->S, 0, sizeof(p->S));
    p->ws = (char*)workspace; p->ws_bytes = workspace_bytes;
    p->state = (uint64_t*)state; p->twiddle = twiddle; p->train = false;
    derive(*p);
    size_t total = 0;
    carve(*p, p->ws, &total);
    if (total > workspace_bytes) { delete p; return -12; }

    // reduction job table: per layer 19 jobs (state_dict order), then the 2 embedding LayerNorm jobs
    const long d = cfg->hidden;
    const SlabMap sm = slab_map(d);
    std::vector<ReduceJob> jobs;
    auto add = [&](const float* src, float* dst, int nsplit, long len) {
        ReduceJob j; j.src = src; j.dst = dst; j.nsplit = nsplit; j.len = (int)len; j.stride = len; j.scale = 1.f; j.pad = 0;
        jobs.push_back(j);
    };
    const int ns = p->nsplit, nb = p->nblk;
    for (int l = 0; l < cfg->layers; ++l) {
        const bsarec_layer_t& g = p->G.layer[l];
        p->slab_w = p->slab_wL[l]; p->slab_b = p->slab_bL[l]; p->part_ln = p->part_lnL[l]; p->part_beta = p->part_betaL[l];
        add(p->part_beta, g.sqrt_beta, cfg->batch, d);
        add(p->part_ln + 4L * nb * d, g.filter_ln_w, nb, d);
        add(p->part_ln + 5L * nb * d, g.filter_ln_b, nb, d);
        add(slab_w_ptr(*p, sm.wq), g.query_w, ns, d * d); add(slab_b_ptr(*p, sm.bq), g.query_b, ns, d);
        add(slab_w_ptr(*p, sm.wk), g.key_w, ns, d * d);   add(slab_b_ptr(*p, sm.bk), g.key_b, ns, d);
        add(slab_w_ptr(*p, sm.wv), g.value_w, ns, d * d); add(slab_b_ptr(*p, sm.bv), g.value_b, ns, d);
        add(slab_w_ptr(*p, sm.wo), g.dense_w, ns, d * d); add(slab_b_ptr(*p, sm.bo), g.dense_b, ns, d);
        add(p->part_ln + 2L * nb * d, g.attn_ln_w, nb, d);
        add(p->part_ln + 3L * nb * d, g.attn_ln_b, nb, d);
        add(slab_w_ptr(*p, sm.w1), g.ffn1_w, ns, 4 * d * d); add(slab_b_ptr(*p, sm.b1), g.ffn1_b, ns, 4 * d);
        add(slab_w_ptr(*p, sm.w2), g.ffn2_w, ns, 4 * d * d); add(slab_b_ptr(*p, sm.b2), g.ffn2_b, ns, d);
        add(p->part_ln + 0L * nb * d, g.ffn_ln_w, nb, d);
        add(p->part_ln + 1L * nb * d, g.ffn_ln_b, nb, d);
    }
    add(p->part_ln0 + 0L * nb * d, p->G.ln_w, nb, d);
    add(p->part_ln0 + 1L * nb * d, p->G.ln_b, nb, d);
    if (p->scatter_in_block) {   // the position gradient = sum over the batch of the embedding gradient rows the block kernel wrote
        ReduceJob j; j.src = p->dz; j.dst = p->G.pos_emb; j.nsplit = cfg->batch; j.len = (int)((long)cfg->seq_len * d);
        j.stride = (long)cfg->seq_len * d; j.scale = 1.f; j.pad = 0;
        jobs.push_back(j);
    } else add(p->part_pos, p->G.pos_emb, p->pos_slices, (long)cfg->seq_len * d);
    if (cfg->filter_kind == 1)                 // FMLPRec: d(complex_weight) of every layer, summed over the sequences
        for (int l = 0; l < cfg->layers; ++l)
            add(p->part_cwL[l], p->G.layer[l].filter_cw, cfg->batch, (long)cfg->cutoff_bins * d * 2);
    p->jobs_per_layer = 19;
    p->prune_ok = p->fused && cfg->layers >= 2 && !cfg->no_prune_top && cfg->filter_kind == 0;      // (the FMLPRec block keeps its full kernels)
    p->pruned = false;
    p->loss_kind = 0; p->bce_pos = nullptr; p->bce_neg = nullptr;
    std::vector<ReduceJob> jobs_pr = jobs;          // key_b is job 6, value_b job 8 of a layer's 19 (state_dict order)
    {
        const size_t base = (size_t)(cfg->layers - 1) * 19;
        ReduceJob& jk = jobs_pr[base + 6]; jk.src = p->part_kvb; jk.nsplit = cfg->batch; jk.stride = d;
        ReduceJob& jv = jobs_pr[base + 8]; jv.src = p->part_kvb + (long)cfg->batch * d; jv.nsplit = cfg->batch; jv.stride = d;
        for (int j : {3, 4, 5, 7, 9, 10, 13, 14, 15, 16})      // weights and the other biases: the pruned products fill TOP_SLABS slabs
            jobs_pr[base + j].nsplit = std::min(p->top_slabs, ns);
    }
    std::vector<int> bmap;
    for (size_t j = 0; j < jobs.size(); ++j)
        for (int ch = 0; ch < cdiv(jobs[j].len, 64); ++ch) bmap.push_back((int)(j << 16) | ch);
    p->red_blocks = (int)bmap.size();
    p->red_elems = 0; p->red_lo = nullptr; p->red_hi = nullptr;
    for (const ReduceJob& j : jobs) {
        p->red_elems += j.len;
        if (!p->red_lo || j.dst < p->red_lo) p->red_lo = j.dst;
        if (!p->red_hi || j.dst + j.len > p->red_hi) p->red_hi = j.dst + j.len;
    }
    hipError_t e = hipMemcpyAsync(p->jobs, jobs.data(), jobs.size() * sizeof(ReduceJob), hipMemcpyHostToDevice,
                                  (hipStream_t)stream);
    if (e == hipSuccess) e = hipMemcpyAsync(p->blockmap, bmap.data(), bmap.size() * sizeof(int), hipMemcpyHostToDevice,
                                            (hipStream_t)stream);
    if (e == hipSuccess) e = hipMemcpyAsync(p->jobs_pruned, jobs_pr.data(), jobs_pr.size() * sizeof(ReduceJob),
                                            hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);   // jobs vector is host-temporary
    if (e != hipSuccess) { delete p; return (int)e; }
    // dry pass: sets every kernel's dynamic-LDS attribute for this shape without launching anything
    g_dry = true;
    int rc = 0;
    for (int last = 0; last < 2 && rc == 0; ++last) {
        rc = last ? bsarec_forward_last(p, reinterpret_cast<const int64_t*>(p->ws), 1, stream)
                  : bsarec_forward(p, reinterpret_cast<const int64_t*>(p->ws), 1, stream);
        if (rc == 0) rc = bsarec_loss(p, reinterpret_cast<const int64_t*>(p->ws), stream);
        if (rc == 0 && p->G.item_emb) rc = bsarec_backward(p, stream);
    }
    g_dry = false;
    p->pruned = false;
    p->train = false;
    if (rc != 0) { delete p; return rc; }
    *out = p;
    return 0;
}

extern "C" void bsarec_plan_destroy(bsarec_plan_t* plan) { delete plan; }

extern "C" long bsarec_buffer_offset(const bsarec_plan_t* p, int buffer, int layer) {
    if (!p) return -1;
    const int N = p->cfg.layers;
    const void* ptr = nullptr;
    switch (buffer) {
        case BSAREC_BUF_LAYER_OUT: if (layer < 0 || layer > N) return -1; ptr = p->X[layer]; break;
        case BSAREC_BUF_LOGITS: ptr = p->logits; break;
        case BSAREC_BUF_DLOGITS: ptr = p->dlogits; break;
        case BSAREC_BUF_LOSS: ptr = p->loss; break;
        case BSAREC_BUF_LOSS_ROWS: ptr = p->loss_rows; break;
        case BSAREC_BUF_DSP: if (layer < 0 || layer >= N) return -1; ptr = p->lb[layer].dsp; break;
        case BSAREC_BUF_HMIX: if (layer < 0 || layer >= N) return -1; ptr = p->lb[layer].hmix; break;
        case BSAREC_BUF_PROBS: if (layer < 0 || layer >= N) return -1; ptr = p->lb[layer].probs; break;
        case BSAREC_BUF_CTX: if (layer < 0 || layer >= N) return -1; ptr = p->lb[layer].ctx; break;
        case BSAREC_BUF_DLAYER_IN: if (layer < 0 || layer > N) return -1; ptr = (layer & 1) ? p->dXb : p->dXa; break;
        default: return -1;
    }
    return (long)((const char*)ptr - p->ws);
}

extern "C" int bsarec_plan_set_dense_grad_hook(bsarec_plan_t* p, bsarec_hook_t hook, void* user, float* lookup_grad) {
    if (!p) return -10;
    p->dense_hook = hook; p->dense_hook_user = user; p->lookup_grad = lookup_grad;
    return 0;
}

extern "C" int bsarec_plan_is_fused(const bsarec_plan_t* p) { return p && p->fused ? 1 : 0; }
extern "C" int bsarec_config_is_fused(const bsarec_config_t* cfg) {
    if (!cfg) return -10;
    RET(check_cfg(*cfg));
    return fused_shape_ok(*cfg) ? 1 : 0;
}

extern "C" int bsarec_buffer_is_bf16(const bsarec_plan_t* p, int buffer, int layer) {
    if (!p || !p->bf) return 0;
    switch (buffer) {
        case BSAREC_BUF_LAYER_OUT: return layer < p->cfg.layers ? 1 : 0;     // the last layer's output stays fp32
        case BSAREC_BUF_HMIX: case BSAREC_BUF_PROBS: case BSAREC_BUF_CTX: return 1;
        case BSAREC_BUF_DLAYER_IN: return 1;
        default: return 0;
    }
}

extern "C" int bsarec_shadow_refresh(bsarec_plan_t* p, void* stream) {
    if (!p) return -10;
    if (!p->bf) return 0;
    const long d = p->cfg.hidden;
    for (int l = 0; l < p->cfg.layers; ++l) {
        const bsarec_layer_t& w = p->P.layer[l];
        const bsarec_layer_t& sh = p->S.layer[l];
        CastJobs6 J;
        const float* src[6] = {w.query_w, w.key_w, w.value_w, w.dense_w, w.ffn1_w, w.ffn2_w};
        float* dst[6] = {sh.query_w, sh.key_w, sh.value_w, sh.dense_w, sh.ffn1_w, sh.ffn2_w};
        for (int i = 0; i < 6; ++i) { J.src[i] = src[i]; J.dst[i] = (unsigned short*)dst[i]; J.n4[i] = (i < 4 ? d * d : 4 * d * d) / 4; }
        hipLaunchKernelGGL(cast_bf16_kernel, dim3(cdiv(4 * d * d / 4, ROW_THREADS), 6), dim3(ROW_THREADS), 0, (hipStream_t)stream, J);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------
static uint32_t drop_thresh(float p) {
    double t = (double)p * 4294967296.0;
    if (t <= 0) return 0;
    if (t >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)t;
}
static DropP make_drop(const bsarec_plan& p, float prob, int site, bool train) {
    DropP d;
    d.thresh = train ? drop_thresh(prob) : 0;
    d.scale = (train && prob > 0.f) ? (float)(1.0 / (1.0 - (double)prob)) : 1.0f;
    d.rng = p.state; d.site = (uint32_t)site;
    return d;
}
static int lpr_for(int d) { return d <= 64 ? 16 : (d <= 128 ? 32 : 64); }

#define DISPATCH_LPR(d, ...) \
    do { switch (lpr_for(d)) { case 16: { constexpr int LPR = 16; __VA_ARGS__; } break; \
                               case 32: { constexpr int LPR = 32; __VA_ARGS__; } break; \
                               default: { constexpr int LPR = 64; __VA_ARGS__; } break; } } while (0)

// tiles whose epilogue needs a whole row of width n (<= 256): pick BN
#define DISPATCH_BN(n, ...) \
    do { if ((n) <= 64) { constexpr int BN = 64; __VA_ARGS__; } \
         else if ((n) <= 128) { constexpr int BN = 128; __VA_ARGS__; } \
         else { constexpr int BN = 256; __VA_ARGS__; } } while (0)

static size_t freq_smem(int L, int d, int cb, int nsrc) {
    return (size_t)(rup(2 * L, 4) + (long)nsrc * cb * 2 * d + (long)nsrc * 8192) * 4;
}

template <int LPR>
static int launch_freq_fwd(const float* X, const float* sb, const float* g, const float* be, float eps, DropP drop,
                           const float* tw, int B, int L, int d, int cb, float* dsp, float* xhat, float* rstd, hipStream_t s,
                           const float* cw = nullptr) {
    auto kern = freq_fwd_kernel<LPR>;
    const size_t smem = freq_smem(L, d, cb, 1);
    static size_t attr = 0;
    if (smem > 48 * 1024 && smem > attr) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr = smem;
    }
    LAUNCH(kern, dim3(B), dim3(ROW_THREADS), smem, s, X, sb, g, be, eps, drop, tw, L, d, cb, dsp, xhat, rstd, cw);
    return (int)hipGetLastError();
}

template <int LPR>
static int launch_freq_bwd(const float* X, const float* dF, const float* dXin, const float* sb, const float* tw, int B, int L,
                           int d, int cb, float* dX, float* pbeta, hipStream_t s, const float* cw = nullptr, float* pcw = nullptr) {
    auto kern = freq_bwd_kernel<LPR>;
    const size_t smem = freq_smem(L, d, cb, 2);
    static size_t attr = 0;
    if (smem > 48 * 1024 && smem > attr) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr = smem;
    }
    LAUNCH(kern, dim3(B), dim3(ROW_THREADS), smem, s, X, dF, dXin, sb, tw, L, d, cb, dX, pbeta, cw, pcw);
    return (int)hipGetLastError();
}

static bool fused_ok(const bsarec_plan& p) { return p.fused; }

// Host-side check of everything a direct (buffer-descriptor) kernel is about to dereference.  These kernels prefetch
// without predicates and rely on the descriptor's range check; a descriptor whose BASE is null with a non-zero range
// reads virtual address 0 + offset -- a memory fault (which on this pool can reset the node), not a wrong number.
// Round 2 saw exactly such a fault ("address (nil)") from an experimental 8-wave variant of the logits backward whose
// source was not kept; whatever its cause was, a launch with a null or oversized operand is now refused here
// (return < 0, nothing enqueued) instead of being found by the GPU.
static bool dw_problem_ok(const DwProblem& q) {
    if (!q.A || !q.B || !q.slab) return false;
    if (q.K < 1 || q.M < 1 || q.N < 1 || q.kchunk < 8 || (q.kchunk & 7) || q.nslab < 1) return false;
    if (q.lda < q.M || q.ldb < q.N) return false;
    const long esz = q.bf16 ? 2 : 4;
    if (((long)(q.K - 1) * q.lda + q.M) * esz >= (1L << 31) || ((long)(q.K - 1) * q.ldb + q.N) * esz >= (1L << 31)) return false;
    return true;
}
static bool dh_problem_ok(const DhP& h) {
    if (!h.A || !h.E || !h.slab) return false;
    if (h.B < 1 || h.V < 1 || h.kchunk < 8 || (h.kchunk & 7) || h.nsplit < 1 || h.lda < h.V) return false;
    if ((long)h.B * h.lda * 4 >= (1L << 31) || (long)h.V * 256 >= (1L << 31)) return false;
    return true;
}

static void fill_top_fwd(bsarec_plan& p, int l, bool tr, TopFwdP& F);

static int launch_fused_fwd(bsarec_plan& p, int l, bool tr, hipStream_t s, const int64_t* ids = nullptr, const GatherP* gp = nullptr,
                            bool top_tail = false /* block l + 1 is the pruned top block: run it as this launch's tail */) {
    const bsarec_config_t& c = p.cfg;
    const bsarec_layer_t& w = p.P.layer[l];
    const bsarec_layer_t& wm = p.bf ? p.S.layer[l] : w;      // MFMA operands: bf16 shadow of the Linear weights (storage = 1)
    LayerBufs& b = p.lb[l];
    FusedFwdP F;
    memset(&F, 0, sizeof(F));
    F.X = p.X[l]; F.Xout = p.X[l + 1];
    F.xout_f32 = (l == c.layers - 1);
    F.sqrt_beta = w.sqrt_beta; F.f_g = w.filter_ln_w; F.f_b = w.filter_ln_b;
    F.wq = wm.query_w; F.bq = w.query_b; F.wk = wm.key_w; F.bk = w.key_b; F.wv = wm.value_w; F.bv = w.value_b;
    F.wo = wm.dense_w; F.bo = w.dense_b; F.a_g = w.attn_ln_w; F.a_b = w.attn_ln_b;
    F.w1 = wm.ffn1_w; F.b1 = w.ffn1_b; F.w2 = wm.ffn2_w; F.b2 = w.ffn2_b; F.ff_g = w.ffn_ln_w; F.ff_b = w.ffn_ln_b;
    F.tw = p.twiddle; F.ids32 = p.ids32;
    F.xhat_f = b.xhat_f; F.rstd_f = b.rstd_f; F.q = b.q; F.k = b.k; F.v = b.v; F.probs = b.probs; F.ctx = b.ctx;
    F.xhat_a = b.xhat_a; F.rstd_a = b.rstd_a; F.hmix = b.hmix; F.u = b.u; F.xhat_ff = b.xhat_ff; F.rstd_ff = b.rstd_ff;
    F.gp = b.gp;
    F.dsp = nullptr;          // FrequencyLayer output stays in LDS on the fused path (BSAREC_BUF_DSP is generic-path only)
    if (l == 0 && gp) {       // the embedding front-end rides in the bottom block's phase 0
        F.e_E = p.P.item_emb; F.e_pos = p.P.pos_emb; F.e_g = p.P.ln_w; F.e_b = p.P.ln_b; F.e_ids = ids; F.e_gp = *gp;
        F.e_drop = make_drop(p, c.p_hidden, 0, tr); F.e_V = c.item_size;
        F.e_X0 = p.X[0]; F.e_xhat = p.xhat0; F.e_rstd = p.rstd0; F.e_ids32 = p.ids32;
    }
    F.L = c.seq_len; F.Lp = p.Lp; F.cb = c.cutoff_bins; F.heads = c.heads;
    F.alpha = c.alpha; F.oma = (float)(1.0 - (double)c.alpha); F.eps = c.ln_eps;
    F.drop_f = make_drop(p, c.p_hidden, 1 + 4 * l, tr); F.drop_p = make_drop(p, c.p_attn, 2 + 4 * l, tr);
    F.drop_o = make_drop(p, c.p_hidden, 3 + 4 * l, tr); F.drop_ff = make_drop(p, c.p_hidden, 4 + 4 * l, tr);
    F.trash = p.trash;
    F.stamps = p.stamps ? p.stamps + 32 * (2 * l) : nullptr;
    TopFwdP TF;
    if (top_tail) fill_top_fwd(p, l + 1, tr, TF);
    if (c.filter_kind == 1) {        // FMLPRec block: whole-spectrum complex filter + feed-forward (no attention branch)
        F.filter_cw = w.filter_cw;
        const size_t fsm = fused_fwd_smem_bytes();
        static bool attr_fm = false;
        if (!attr_fm) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_layer_fwd_kernel<32, false, NoTail, false, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)fsm)); attr_fm = true; }
        ProfScope prof(BSAREC_K_FUSED_FWD, s);
        LAUNCH((fused_layer_fwd_kernel<32, false, NoTail, false, true>), dim3(c.batch), dim3(512), fsm, s, F, NoTail());
        return (int)hipGetLastError();
    }
    if (!p.bf && c.chain_kernels && !c.x3_products) {
        // register-chain forward (fused_chain.h): one wave per 16-token tile, two workgroup barriers
        const size_t csm = fused_chain_fwd_smem_bytes(top_tail);
#define CHAIN_FWD_CASE(DHV) { \
        static bool attr = false, attr_t = false; \
        if (!attr) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_chain_fwd_kernel<DHV, NoTail>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_chain_fwd_smem_bytes(false))); attr = true; } \
        if (!attr_t) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_chain_fwd_kernel<DHV, TopFwdP>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_chain_fwd_smem_bytes(true))); attr_t = true; } \
        ProfScope prof(BSAREC_K_FUSED_FWD, s); \
        if (top_tail) LAUNCH((fused_chain_fwd_kernel<DHV, TopFwdP>), dim3(c.batch), dim3(512), csm, s, F, TF); \
        else LAUNCH((fused_chain_fwd_kernel<DHV, NoTail>), dim3(c.batch), dim3(512), csm, s, F, NoTail()); }
        if (p.dh == 16) CHAIN_FWD_CASE(16) else if (p.dh == 32) CHAIN_FWD_CASE(32) else CHAIN_FWD_CASE(64)
#undef CHAIN_FWD_CASE
        return (int)hipGetLastError();
    }
    const size_t smem = fused_fwd_smem_bytes();
#define FUSED_FWD_CASE(DHV, BFV, X3V) { \
        static bool attr = false, attr_t = false; \
        if (!attr) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_layer_fwd_kernel<DHV, BFV, NoTail, X3V>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); attr = true; } \
        if (!attr_t) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_layer_fwd_kernel<DHV, BFV, TopFwdP, X3V>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); attr_t = true; } \
        ProfScope prof(BSAREC_K_FUSED_FWD, s); \
        if (top_tail) LAUNCH((fused_layer_fwd_kernel<DHV, BFV, TopFwdP, X3V>), dim3(c.batch), dim3(512), smem, s, F, TF); \
        else LAUNCH((fused_layer_fwd_kernel<DHV, BFV, NoTail, X3V>), dim3(c.batch), dim3(512), smem, s, F, NoTail()); }
    if (p.bf) { if (p.dh == 16) FUSED_FWD_CASE(16, true, false) else if (p.dh == 32) FUSED_FWD_CASE(32, true, false) else FUSED_FWD_CASE(64, true, false) }
    else if (c.x3_products) { if (p.dh == 16) FUSED_FWD_CASE(16, false, true) else if (p.dh == 32) FUSED_FWD_CASE(32, false, true) else FUSED_FWD_CASE(64, false, true) }
    else { if (p.dh == 16) FUSED_FWD_CASE(16, false, false) else if (p.dh == 32) FUSED_FWD_CASE(32, false, false) else FUSED_FWD_CASE(64, false, false) }
#undef FUSED_FWD_CASE
    return (int)hipGetLastError();
}

static int launch_fused_bwd(bsarec_plan& p, int l, bool tr, const float* dY, float* dXout, hipStream_t s, bool top,
                            const TopBwdP* head = nullptr /* the pruned top block's backward runs as this launch's head */) {
    const bsarec_config_t& c = p.cfg;
    const bsarec_layer_t& w = p.P.layer[l];
    const bsarec_layer_t& wm = p.bf ? p.S.layer[l] : w;
    LayerBufs& b = p.lb[l];
    const long nb = p.nblk, d = c.hidden;
    FusedBwdP F;
    memset(&F, 0, sizeof(F));
    F.dY = dY; F.dX = dXout; F.X = p.X[l];
    F.sqrt_beta = w.sqrt_beta; F.f_g = w.filter_ln_w; F.wq = wm.query_w; F.wk = wm.key_w; F.wv = wm.value_w; F.wo = wm.dense_w;
    F.a_g = w.attn_ln_w; F.w1 = wm.ffn1_w; F.w2 = wm.ffn2_w; F.ff_g = w.ffn_ln_w; F.tw = p.twiddle;
    F.xhat_f = b.xhat_f; F.rstd_f = b.rstd_f; F.q = b.q; F.k = b.k; F.v = b.v; F.probs = b.probs;
    F.xhat_a = b.xhat_a; F.rstd_a = b.rstd_a; F.u = b.gp; F.xhat_ff = b.xhat_ff; F.rstd_ff = b.rstd_ff;
    if (top) { F.dh_slabs = p.dlast_slab; F.dh_nsplit = p.loss_kind == 1 ? 1 : p.vsplit; F.dh_stride = (long)c.batch * d; }
    if (l == 0) {       // the embedding front-end's backward (Drop + LayerNorm) rides in the bottom block's epilogue
        F.e_dz = p.dz; F.e_xhat = p.xhat0; F.e_rstd = p.rstd0; F.e_g = p.P.ln_w;
        F.e_pg = p.part_ln0; F.e_pb = p.part_ln0 + nb * d; F.e_drop = make_drop(p, c.p_hidden, 0, tr);
        F.e_dx_extra = p.ext_dy ? p.ext_mid[0] : nullptr;
    }
    F.dT = p.dT; F.dU = p.dU; F.dO = p.dO; F.dq = p.dq; F.dk = p.dk; F.dv = p.dv;
    F.pg_ff = p.part_ln + 0 * nb * d; F.pb_ff = p.part_ln + 1 * nb * d; F.pg_a = p.part_ln + 2 * nb * d;
    F.pb_a = p.part_ln + 3 * nb * d; F.pg_f = p.part_ln + 4 * nb * d; F.pb_f = p.part_ln + 5 * nb * d;
    F.pbeta = p.part_beta;
    F.L = c.seq_len; F.Lp = p.Lp; F.cb = c.cutoff_bins; F.heads = c.heads;
    F.alpha = c.alpha; F.oma = (float)(1.0 - (double)c.alpha);
    F.drop_f = make_drop(p, c.p_hidden, 1 + 4 * l, tr); F.drop_p = make_drop(p, c.p_attn, 2 + 4 * l, tr);
    F.drop_o = make_drop(p, c.p_hidden, 3 + 4 * l, tr); F.drop_ff = make_drop(p, c.p_hidden, 4 + 4 * l, tr);
    F.trash = p.trash;
    F.stamps = p.stamps ? p.stamps + 32 * (2 * l + 1) : nullptr;
    const size_t smem = fused_bwd_smem_bytes();
    if (c.filter_kind == 1) {
        F.filter_cw = w.filter_cw; F.pcw = p.part_cwL[l];
        static bool attr_fm = false;
        if (!attr_fm) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_layer_bwd_kernel<32, false, NoTail, false, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); attr_fm = true; }
        ProfScope prof(BSAREC_K_FUSED_BWD, s);
        LAUNCH((fused_layer_bwd_kernel<32, false, NoTail, false, true>), dim3(c.batch), dim3(512), smem, s, F, NoTail());
        return (int)hipGetLastError();
    }
#define FUSED_BWD_CASE(DHV, BFV, X3V) { \
        static bool attr = false; \
        if (!attr) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_layer_bwd_kernel<DHV, BFV, NoTail, X3V>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
                     HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_layer_bwd_kernel<DHV, BFV, TopBwdP, X3V>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); attr = true; } \
        ProfScope prof(BSAREC_K_FUSED_BWD, s); \
        if (head) LAUNCH((fused_layer_bwd_kernel<DHV, BFV, TopBwdP, X3V>), dim3(c.batch), dim3(512), smem, s, F, *head); \
        else LAUNCH((fused_layer_bwd_kernel<DHV, BFV, NoTail, X3V>), dim3(c.batch), dim3(512), smem, s, F, NoTail()); }
    if (p.bf) { if (p.dh == 16) FUSED_BWD_CASE(16, true, false) else if (p.dh == 32) FUSED_BWD_CASE(32, true, false) else FUSED_BWD_CASE(64, true, false) }
    else if (c.x3_products) { if (p.dh == 16) FUSED_BWD_CASE(16, false, true) else if (p.dh == 32) FUSED_BWD_CASE(32, false, true) else FUSED_BWD_CASE(64, false, true) }
    else { if (p.dh == 16) FUSED_BWD_CASE(16, false, false) else if (p.dh == 32) FUSED_BWD_CASE(32, false, false) else FUSED_BWD_CASE(64, false, false) }
#undef FUSED_BWD_CASE
    return (int)hipGetLastError();
}

static void fill_top_fwd(bsarec_plan& p, int l, bool tr, TopFwdP& F) {
    const bsarec_config_t& c = p.cfg;
    const bsarec_layer_t& w = p.P.layer[l];
    LayerBufs& b = p.lb[l];
    memset(&F, 0, sizeof(F));
    F.X = p.X[l]; F.Xout = p.X[l + 1];
    F.sqrt_beta = w.sqrt_beta; F.f_g = w.filter_ln_w; F.f_b = w.filter_ln_b;
    F.wq = w.query_w; F.bq = w.query_b; F.wk = w.key_w; F.bk = w.key_b; F.wv = w.value_w; F.bv = w.value_b;
    F.wo = w.dense_w; F.bo = w.dense_b; F.a_g = w.attn_ln_w; F.a_b = w.attn_ln_b;
    F.w1 = w.ffn1_w; F.b1 = w.ffn1_b; F.w2 = w.ffn2_w; F.b2 = w.ffn2_b; F.ff_g = w.ffn_ln_w; F.ff_b = w.ffn_ln_b;
    F.tw = p.twiddle; F.ids32 = p.ids32;
    F.xhat_f = b.xhat_f; F.rstd_f = b.rstd_f; F.q = b.q; F.k = b.k; F.v = b.v; F.probs = b.probs; F.ctx = b.ctx;
    F.xhat_a = b.xhat_a; F.rstd_a = b.rstd_a; F.hmix = b.hmix; F.u = b.u; F.xhat_ff = b.xhat_ff; F.rstd_ff = b.rstd_ff;
    F.low = b.dsp;
    F.wk_sh = p.S.layer[l].key_w; F.wv_sh = p.S.layer[l].value_w;
    F.L = c.seq_len; F.Lp = p.Lp; F.cb = c.cutoff_bins; F.heads = c.heads;
    F.alpha = c.alpha; F.oma = (float)(1.0 - (double)c.alpha); F.eps = c.ln_eps;
    F.drop_f = make_drop(p, c.p_hidden, 1 + 4 * l, tr); F.drop_p = make_drop(p, c.p_attn, 2 + 4 * l, tr);
    F.drop_o = make_drop(p, c.p_hidden, 3 + 4 * l, tr); F.drop_ff = make_drop(p, c.p_hidden, 4 + 4 * l, tr);
    F.stamps = p.stamps ? p.stamps + 32 * (2 * l) : nullptr;
}

static int launch_top_fwd(bsarec_plan& p, int l, bool tr, hipStream_t s) {
    const bsarec_config_t& c = p.cfg;
    TopFwdP F;
    fill_top_fwd(p, l, tr, F);
    const size_t smem = top_fwd_smem_bytes();
#define TOP_FWD_CASE(DHV, BFV) { \
        static bool attr = false; \
        if (!attr) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(top_fwd_kernel<DHV, BFV>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); attr = true; } \
        LAUNCH((top_fwd_kernel<DHV, BFV>), dim3(c.batch), dim3(256), smem, s, F); }
    if (p.bf) { if (p.dh == 16) TOP_FWD_CASE(16, true) else if (p.dh == 32) TOP_FWD_CASE(32, true) else TOP_FWD_CASE(64, true) }
    else { if (p.dh == 16) TOP_FWD_CASE(16, false) else if (p.dh == 32) TOP_FWD_CASE(32, false) else TOP_FWD_CASE(64, false) }
#undef TOP_FWD_CASE
    return (int)hipGetLastError();
}

static void fill_top_bwd(bsarec_plan& p, int l, bool tr, float* dXout, TopBwdP& F) {
    const bsarec_config_t& c = p.cfg;
    const bsarec_layer_t& w = p.P.layer[l];
    LayerBufs& b = p.lb[l];
    const long nb = p.nblk, d = c.hidden;
    memset(&F, 0, sizeof(F));
    F.dX = dXout; F.X = p.X[l];
    F.sqrt_beta = w.sqrt_beta; F.f_g = w.filter_ln_w; F.wq = w.query_w; F.wk = w.key_w; F.wv = w.value_w; F.wo = w.dense_w;
    F.a_g = w.attn_ln_w; F.w1 = w.ffn1_w; F.w2 = w.ffn2_w; F.ff_g = w.ffn_ln_w; F.tw = p.twiddle;
    F.xhat_f = b.xhat_f; F.rstd_f = b.rstd_f; F.q = b.q; F.k = b.k; F.v = b.v; F.probs = b.probs;
    F.xhat_a = b.xhat_a; F.rstd_a = b.rstd_a; F.u = b.u; F.xhat_ff = b.xhat_ff; F.rstd_ff = b.rstd_ff; F.low = b.dsp;
    F.dh_slabs = p.dlast_slab; F.dh_nsplit = p.loss_kind == 1 ? 1 : p.vsplit; F.dh_stride = (long)c.batch * d;
    F.dT = p.top_dT; F.dU = p.top_dU; F.dO = p.top_dO; F.dq = p.top_dq;
    F.ak = p.top_ak; F.rk = p.top_rk; F.av = p.top_av; F.rv = p.top_rv;
    F.pbk = p.part_kvb; F.pbv = p.part_kvb + (long)c.batch * d;
    F.pg_ff = p.part_ln + 0 * nb * d; F.pb_ff = p.part_ln + 1 * nb * d; F.pg_a = p.part_ln + 2 * nb * d;
    F.pb_a = p.part_ln + 3 * nb * d; F.pg_f = p.part_ln + 4 * nb * d; F.pb_f = p.part_ln + 5 * nb * d;
    F.pbeta = p.part_beta;
    F.L = c.seq_len; F.Lp = p.Lp; F.cb = c.cutoff_bins; F.heads = c.heads;
    F.alpha = c.alpha; F.oma = (float)(1.0 - (double)c.alpha);
    F.drop_f = make_drop(p, c.p_hidden, 1 + 4 * l, tr); F.drop_p = make_drop(p, c.p_attn, 2 + 4 * l, tr);
    F.drop_o = make_drop(p, c.p_hidden, 3 + 4 * l, tr); F.drop_ff = make_drop(p, c.p_hidden, 4 + 4 * l, tr);
    F.stamps = p.stamps ? p.stamps + 32 * (2 * l + 1) : nullptr;
}

static int launch_top_bwd(bsarec_plan& p, int l, bool tr, float* dXout, hipStream_t s) {
    const bsarec_config_t& c = p.cfg;
    TopBwdP F;
    fill_top_bwd(p, l, tr, dXout, F);      // (reads p.part_ln / p.part_beta of THIS layer: call inside the layer's iteration)
    const size_t smem = top_bwd_smem_bytes();
#define TOP_BWD_CASE(DHV, BFV) { \
        static bool attr = false; \
        if (!attr) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(top_bwd_kernel<DHV, BFV>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); attr = true; } \
        LAUNCH((top_bwd_kernel<DHV, BFV>), dim3(c.batch), dim3(256), smem, s, F); }
    if (p.bf) { if (p.dh == 16) TOP_BWD_CASE(16, true) else if (p.dh == 32) TOP_BWD_CASE(32, true) else TOP_BWD_CASE(64, true) }
    else { if (p.dh == 16) TOP_BWD_CASE(16, false) else if (p.dh == 32) TOP_BWD_CASE(32, false) else TOP_BWD_CASE(64, false) }
#undef TOP_BWD_CASE
    return (int)hipGetLastError();
}

extern "C" int bsarec_debug_stamps(bsarec_plan_t* p, void* dev_buf) { if (!p) return -10; p->stamps = (long long*)dev_buf; return 0; }

extern "C" int bsarec_step_begin(bsarec_plan_t* p, void* stream) {
    if (!p) return -10;
    LAUNCH(step_begin_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, p->state, (long long*)nullptr, 0);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
static int forward_impl(bsarec_plan_t* p, const int64_t* ids, int train, void* stream, const GatherP& gp, bool last_only);

extern "C" int bsarec_forward(bsarec_plan_t* p, const int64_t* ids, int train, void* stream) {
    GatherP none;
    memset(&none, 0, sizeof(none));
    return forward_impl(p, ids, train, stream, none, false);
}

extern "C" int bsarec_forward_last(bsarec_plan_t* p, const int64_t* ids, int train, void* stream) {
    GatherP none;
    memset(&none, 0, sizeof(none));
    return forward_impl(p, ids, train, stream, none, true);
}

static int forward_impl(bsarec_plan_t* p, const int64_t* ids, int train, void* stream, const GatherP& gp, bool last_only) {
    if (!p || (!ids && !gp.table)) return -10;
    PlanScope scope(p);
    hipStream_t s = (hipStream_t)stream;
    const bsarec_config_t& c = p->cfg;
    const int T = p->T, d = c.hidden, L = c.seq_len, B = c.batch, h = c.heads, dh = p->dh, Lp = p->Lp;
    const bool tr = train != 0;
    p->train = tr;
    p->pruned = last_only && p->prune_ok;
    const XformP nox = no_xform();

    const bool embed_in_block = fused_ok(*p) && p->embed_in_block;
    if (!embed_in_block)
    DISPATCH_LPR(d, {
        constexpr int RPB = ROW_THREADS / LPR;
        LAUNCH(embed_fwd_kernel<LPR>, dim3(cdiv(T, RPB)), dim3(ROW_THREADS), 0, s, ids, gp, p->P.item_emb,
                           p->P.pos_emb, p->P.ln_w, p->P.ln_b, c.ln_eps, make_drop(*p, c.p_hidden, 0, tr), T, L, d,
                           c.item_size, p->X[0], p->xhat0, p->rstd0, p->ids32);
        HIPCHK(hipGetLastError());
    });

    for (int l = 0; l < c.layers; ++l) {
        const bsarec_layer_t& w = p->P.layer[l];
        LayerBufs& b = p->lb[l];
        const float* X = p->X[l];
        if (fused_ok(*p)) {
            const bool tail = p->pruned && c.layers >= 2 && !c.separate_top;     // top block = tail of the launch below it
            if (p->pruned && l == c.layers - 1) { if (!tail) RET(launch_top_fwd(*p, l, tr, s)); }
            else RET(launch_fused_fwd(*p, l, tr, s, ids, (l == 0 && embed_in_block) ? &gp : nullptr, tail && l == c.layers - 2));
            continue;
        }
        // K2 FrequencyLayer
        DISPATCH_LPR(d, RET(launch_freq_fwd<LPR>(X, w.sqrt_beta, w.filter_ln_w, w.filter_ln_b, c.ln_eps,
                                                 make_drop(*p, c.p_hidden, 1 + 4 * l, tr), p->twiddle, B, L, d,
                                                 c.cutoff_bins, b.dsp, b.xhat_f, b.rstd_f, s,
                                                 c.filter_kind == 1 ? w.filter_cw : nullptr)));
        // K3 Q, K, V projections (one launch, 3 problems)
        {
            GemmP g = gemm_defaults(T, d, d);
            g.nprob = 3; g.lda = d; g.ldb = d;
            g.A[0] = g.A[1] = g.A[2] = X;
            g.B[0] = w.query_w; g.B[1] = w.key_w; g.B[2] = w.value_w;
            auto e = epi_linear<true, false, false>(b.q, d);
            e.C[1] = b.k; e.C[2] = b.v;
            e.bias[0] = w.query_b; e.bias[1] = w.key_b; e.bias[2] = w.value_b;
            RET((launch_gemm<64, 64, 2, 2, false, false, XF_NONE, XF_NONE, false>(g, nox, e, nullptr, 1, s, BSAREC_K_QKV)));
        }
        // K4a scores + mask + softmax -> probs
        {
            GemmP g = gemm_defaults(L, Lp, dh);
            g.Nb = L; g.lda = d; g.ldb = d; g.nh = h;
            g.A[0] = b.q; g.B[0] = b.k;
            g.a_sb = (long)L * d; g.a_sh = dh; g.b_sb = (long)L * d; g.b_sh = dh;
            EpiSoftmax e; e.ids = p->ids32; e.L = L; e.Lp = Lp; e.sqrt_dh = sqrtf((float)dh); e.P = b.probs;
            DISPATCH_BN(Lp, RET((launch_gemm<64, BN, 2, 2, false, false, XF_NONE, XF_NONE, false>(g, nox, e, nullptr, B * h, s))));
        }
        // K4b context = Drop(probs) . V
        {
            GemmP g = gemm_defaults(L, dh, Lp);
            g.Kv = L; g.lda = Lp; g.ldb = d; g.nh = h;
            g.A[0] = b.probs; g.B[0] = b.v;
            g.a_sb = (long)h * L * Lp; g.a_sh = (long)L * Lp; g.b_sb = (long)L * d; g.b_sh = dh;
            XformP xf = nox; xf.drop = make_drop(*p, c.p_attn, 2 + 4 * l, tr); xf.L = L; xf.Lp = Lp;
            auto e = epi_linear<false, false, false>(b.ctx, d);
            e.c_sb = (long)L * d; e.c_sh = dh;
            RET((launch_gemm<64, 64, 2, 2, false, true, XF_DROP, XF_NONE, false>(g, xf, e, nullptr, B * h, s)));
        }
        // K5 dense + dropout + residual + LayerNorm + alpha mix
        {
            GemmP g = gemm_defaults(T, d, d);
            g.lda = d; g.ldb = d; g.A[0] = b.ctx; g.B[0] = w.dense_w;
            EpiLN<true> e;
            e.bias = w.dense_b; e.R = X; e.drop = make_drop(*p, c.p_hidden, 3 + 4 * l, tr);
            e.gamma = w.attn_ln_w; e.beta = w.attn_ln_b; e.eps = c.ln_eps;
            e.Y = b.hmix; e.xhat = b.xhat_a; e.rstd = b.rstd_a;
            e.dsp = b.dsp; e.alpha = c.alpha; e.oma = (float)(1.0 - (double)c.alpha);
            DISPATCH_BN(d, RET((launch_gemm<64, BN, 2, 2, false, false, XF_NONE, XF_NONE, false>(g, nox, e, nullptr, 1, s))));
        }
        // K6a dense_1
        {
            GemmP g = gemm_defaults(T, 4 * d, d);
            g.lda = d; g.ldb = d; g.A[0] = b.hmix; g.B[0] = w.ffn1_w;
            auto e = epi_linear<true, false, false>(b.u, 4 * d);
            e.bias[0] = w.ffn1_b;
            RET((launch_gemm<64, 64, 2, 2, false, false, XF_NONE, XF_NONE, false>(g, nox, e, nullptr, 1, s, BSAREC_K_FFN1)));
        }
        // K6b gelu + dense_2 + dropout + residual + LayerNorm
        {
            GemmP g = gemm_defaults(T, d, 4 * d);
            g.lda = 4 * d; g.ldb = 4 * d; g.A[0] = b.u; g.B[0] = w.ffn2_w;
            EpiLN<false> e;
            e.bias = w.ffn2_b; e.R = b.hmix; e.drop = make_drop(*p, c.p_hidden, 4 + 4 * l, tr);
            e.gamma = w.ffn_ln_w; e.beta = w.ffn_ln_b; e.eps = c.ln_eps;
            e.Y = p->X[l + 1]; e.xhat = b.xhat_ff; e.rstd = b.rstd_ff;
            e.dsp = nullptr; e.alpha = 0.f; e.oma = 1.f;
            XformP xa = nox; xa.act = c.hidden_act;
            DISPATCH_BN(d, RET((launch_gemm<64, BN, 2, 2, false, false, XF_GELU, XF_NONE, false>(g, xa, e, nullptr, 1, s, BSAREC_K_FFN2))));
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// logits / loss
// ---------------------------------------------------------------------------------------------
extern "C" int bsarec_logits(bsarec_plan_t* p, void* stream) {
    if (!p) return -10;
    PlanScope scope(p);
    hipStream_t s = (hipStream_t)stream;
    const bsarec_config_t& c = p->cfg;
    const int d = c.hidden, L = c.seq_len;
    // hidden = 64 and a small problem: straight from global memory into the MFMA operands (dw_direct.h).  Measured on one box:
    // C1 (B = 256, V = 3,417) 0.1569 -> 0.1551 ms/step; at B = 256 x V = 12,102 and 1,024 x 20,034 the direct form LOSES (every
    // 32-row block re-reads the whole table, 4-byte stores: 0.134 -> 0.140 and 0.661 -> 0.687 ms) -- those keep the tiled GEMM
    if (p->fused && d == 64 && p->Vp <= 4096 && c.batch <= 512 && !g_dry) {
        LogitsP G;
        G.H = p->X[c.layers] + (long)(L - 1) * d; G.ldh = (long)L * d; G.E = p->P.item_emb; G.C = p->logits;
        G.B = c.batch; G.V = c.item_size; G.Vp = p->Vp;
        const long units = (long)cdiv(c.batch, 32) * cdiv(p->Vp, 32);
        ProfScope prof(BSAREC_K_LOGITS, s);
        hipLaunchKernelGGL(logits_direct_kernel, dim3((unsigned)cdiv(units, 4)), dim3(256), 0, s, G);
        return (int)hipGetLastError();
    }
    GemmP g = gemm_defaults(c.batch, p->Vp, d);
    g.Nb = c.item_size; g.lda = (long)L * d; g.ldb = d;
    g.A[0] = p->X[c.layers] + (long)(L - 1) * d; g.B[0] = p->P.item_emb;
    auto e = epi_linear<false, false, false>(p->logits, p->Vp);
    return launch_gemm<64, 64, 2, 2, false, false, XF_NONE, XF_NONE, false>(g, no_xform(), e, nullptr, 1, s, BSAREC_K_LOGITS, true);
}

static int loss_impl(bsarec_plan_t* p, const int64_t* answers, void* stream, bool with_mean) {
    if (!p || !answers) return -10;
    hipStream_t s = (hipStream_t)stream;
    RET(bsarec_logits(p, stream));
    p->loss_kind = 0;
    const bsarec_config_t& c = p->cfg;
    // up to 4,096 classes: scalar row in registers (C1); up to 24,576: float4 row in registers (C2's Beauty, C4's Yelp catalogue);
    // beyond: streamed in three passes (C5: 40 MB rows)
    const int V = c.item_size;
    const float inv_b = 1.0f / (float)c.batch;
#define CE_VEC(N4) LAUNCH(ce_rows_vec_kernel<N4>, dim3(c.batch), dim3(ROW_THREADS), 0, s, p->logits, answers, V, p->Vp, inv_b, p->dlogits, p->loss_rows)
    if (V <= CE_MAX_PER_THREAD * ROW_THREADS || V > 24 * 4 * ROW_THREADS)
        LAUNCH(ce_rows_kernel, dim3(c.batch), dim3(ROW_THREADS), 0, s, p->logits, answers, V, p->Vp, inv_b, p->dlogits, p->loss_rows);
    else if (V <= 8 * 4 * ROW_THREADS) CE_VEC(8);
    else if (V <= 12 * 4 * ROW_THREADS) CE_VEC(12);
    else if (V <= 16 * 4 * ROW_THREADS) CE_VEC(16);
    else if (V <= 20 * 4 * ROW_THREADS) CE_VEC(20);
    else CE_VEC(24);
#undef CE_VEC
    HIPCHK(hipGetLastError());
    if (with_mean) LAUNCH(loss_mean_kernel, dim3(1), dim3(ROW_THREADS), 0, s, p->loss_rows, c.batch, p->loss);
    return (int)hipGetLastError();
}

extern "C" int bsarec_loss(bsarec_plan_t* p, const int64_t* answers, void* stream) { return loss_impl(p, answers, stream, true); }

static int loss_pair(bsarec_plan_t* p, const int64_t* pos_ids, const int64_t* neg_ids, void* stream, int logsig);

// SASRec's head (sibling model on the same encoder; run the plan with alpha = 0): src/model/sasrec.py:41-63
extern "C" int bsarec_loss_bce(bsarec_plan_t* p, const int64_t* pos_ids, const int64_t* neg_ids, void* stream) {
    return loss_pair(p, pos_ids, neg_ids, stream, 0);
}
// FMLPRec's head: src/model/fmlprec.py:41-62
extern "C" int bsarec_loss_logsig(bsarec_plan_t* p, const int64_t* pos_ids, const int64_t* neg_ids, void* stream) {
    return loss_pair(p, pos_ids, neg_ids, stream, 1);
}

static int loss_pair(bsarec_plan_t* p, const int64_t* pos_ids, const int64_t* neg_ids, void* stream, int logsig) {
    if (!p || !pos_ids || !neg_ids) return -10;
    hipStream_t s = (hipStream_t)stream;
    const bsarec_config_t& c = p->cfg;
    const int L = c.seq_len, d = c.hidden;
    p->loss_kind = 1; p->bce_pos = pos_ids; p->bce_neg = neg_ids;
    LAUNCH(bce_rows_kernel, dim3(1), dim3(ROW_THREADS), 0, s, p->X[c.layers] + (long)(L - 1) * d, (long)L * d, p->P.item_emb,
           pos_ids, neg_ids, c.batch, d, c.item_size, p->dlogits, p->loss, logsig);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
static int backward_impl(bsarec_plan_t* p, void* stream, const TickP& tick, const bsarec_adam_t* fuse_adam = nullptr);

extern "C" int bsarec_backward(bsarec_plan_t* p, void* stream) {
    TickP none;
    memset(&none, 0, sizeof(none));
    return backward_impl(p, stream, none);
}

extern "C" int bsarec_backward_seq(bsarec_plan_t* p, const float* d_out, void* stream) {
    if (!p || !d_out) return -10;
    if (p->pruned) return -13;                 // the forward kept only the last row of the top block: run bsarec_forward
    TickP none;
    memset(&none, 0, sizeof(none));
    p->ext_dy = d_out;
    const int rc = backward_impl(p, stream, none);
    p->ext_dy = nullptr;
    return rc;
}

extern "C" int bsarec_backward_seq_multi(bsarec_plan_t* p, const float* const* d_outs, void* stream) {
    if (!p || !d_outs) return -10;
    const int N = p->cfg.layers;
    if (!d_outs[N]) return -10;                // the last layer's gradient is always given (zeros if the caller has none)
    if (p->pruned) return -13;                 // the forward kept only the last row of the top block: run bsarec_forward
    for (int l = 0; l < N; ++l) p->ext_mid[l] = d_outs[l];
    const int rc = bsarec_backward_seq(p, d_outs[N], stream);
    for (int l = 0; l < N; ++l) p->ext_mid[l] = nullptr;
    return rc;
}


// Can the final gradient reduction and Adam be one launch (reduce_adam_kernel)?  Needs the direct weight-gradient
// launch of block 0 to host the step tick, plain single-GPU gradient sources, and every gradient tensor inside the flat
// arena the update walks (item table + the reduction jobs' targets = the whole arena).
static bool can_fuse_adam(const bsarec_plan& p, const bsarec_adam_t& a) {
    if (!p.fused || !p.direct_dw || p.loss_kind != 0) return false;
    if (a.grads2 || a.n_grad_srcs > 0 || a.grad_scale != 1.0f) return false;
    if (!p.G.item_emb || p.G.item_emb < a.grads) return false;
    const long item = (long)p.cfg.item_size * p.cfg.hidden;
    const long off = p.G.item_emb - a.grads;
    if (off < 0 || off + item > a.n || (off & 3) || (item & 3)) return false;
    return p.red_elems + item == a.n && p.red_lo >= a.grads && p.red_hi <= a.grads + a.n;
}

static int backward_impl(bsarec_plan_t* p, void* stream, const TickP& tick, const bsarec_adam_t* fuse_adam) {
    if (!p) return -10;
    if (!p->G.item_emb) return -13;
    PlanScope scope(p);
    hipStream_t s = (hipStream_t)stream;
    const bsarec_config_t& c = p->cfg;
    const int T = p->T, d = c.hidden, L = c.seq_len, B = c.batch, h = c.heads, dh = p->dh, Lp = p->Lp, N = c.layers;
    const bool tr = p->train;
    const XformP nox = no_xform();
    const SlabMap sm = slab_map(d);
    const int ns = p->nsplit, nb = p->nblk;
    const float* hlast = p->X[N] + (long)(L - 1) * d;
    const bool direct_logits = !p->ext_dy && p->fused && p->direct_dw && p->loss_kind == 0 && (long)B * p->Vp * 4 < (1L << 30) &&
                               (long)c.item_size * d * 4 < (1L << 31);

    if (p->ext_dy) {             // backward of forward(): no head on this path, the item table gets its lookup rows only
        if (!g_dry) HIPCHK(hipMemsetAsync(p->G.item_emb, 0, (size_t)c.item_size * d * sizeof(float), s));
    } else if (p->loss_kind == 1) {     // SASRec's BCE pair: two embedding rows per sequence instead of the dense logits path
        if (!g_dry) HIPCHK(hipMemsetAsync(p->G.item_emb, 0, (size_t)c.item_size * d * sizeof(float), s));
        LAUNCH(bce_bwd_kernel, dim3(B), dim3(64), 0, s, hlast, (long)L * d, p->P.item_emb, p->bce_pos, p->bce_neg, p->dlogits, B, d,
               c.item_size, p->dlast_slab, p->G.item_emb);
        HIPCHK(hipGetLastError());
    } else if (direct_logits) {
        // fused shape: dE = dlogits^T . h_last (K = B rows, written straight into the gradient buffer) and the split-K
        // slabs of d(h_last) = dlogits . E by the direct kernels (dw_direct.h), one launch
        DwProblem q;
        memset(&q, 0, sizeof(q));
        q.A = p->dlogits; q.B = hlast; q.lda = p->Vp; q.ldb = (long)L * d; q.M = c.item_size; q.N = d; q.K = B;
        q.kchunk = (int)rup(B, 32); q.nslab = 1; q.slab = p->G.item_emb; q.bslab = nullptr; q.gelu = 0;
        DhP H;
        memset(&H, 0, sizeof(H));
        H.A = p->dlogits; H.lda = p->Vp; H.E = p->P.item_emb; H.B = B; H.V = c.item_size; H.kchunk = p->vchunk;
        H.nsplit = p->vsplit; H.slab = p->dlast_slab;
        const int tiles = cdiv(c.item_size, 64);
        if (!dw_problem_ok(q) || !dh_problem_ok(H)) return -21;
        const dim3 lb_grid(tiles + cdiv(cdiv(B, 32) * p->vsplit, 4));
        if (p->bf) LAUNCH(logits_bwd_direct_kernel<true>, lb_grid, dim3(256), 0, s, q, tiles, H);
        else LAUNCH(logits_bwd_direct_kernel<false>, lb_grid, dim3(256), 0, s, q, tiles, H);
        HIPCHK(hipGetLastError());
    } else
    // dE (dense, logits path) = dlogits^T . h_last  [V, d] (overwrites the gradient buffer) and the split-K slabs of
    // d(h_last) = dlogits . E -- one launch
    {
        PairP G;
        memset(&G, 0, sizeof(G));
        G.A = gemm_defaults(c.item_size, d, B);
        G.A.lda = p->Vp; G.A.ldb = (long)L * d; G.A.A[0] = p->dlogits; G.A.B[0] = hlast;
        G.EA = epi_linear<false, false, false>(p->G.item_emb, d);
        G.B = gemm_defaults(B, d, p->Vp);
        G.B.Kv = c.item_size; G.B.lda = p->Vp; G.B.ldb = d; G.B.A[0] = p->dlogits; G.B.B[0] = p->P.item_emb;
        G.B.nsplit = p->vsplit; G.B.kchunk = p->vchunk;
        G.EB = epi_linear<false, false, false>(p->dlast_slab, d);
        G.EB.c_split = (long)B * d;
        G.tilesA = cdiv(c.item_size, 64) * cdiv(d, 64);
        G.tilesB_m = cdiv(B, 64);
        if (d <= 64) {
            constexpr size_t smem = GemmSmem<64, 64, true, true>::BYTES > GemmSmem<64, 64, false, true>::BYTES
                                        ? GemmSmem<64, 64, true, true>::BYTES : GemmSmem<64, 64, false, true>::BYTES;
            LAUNCH(gemm_logits_bwd_kernel, dim3(G.tilesA + G.tilesB_m * p->vsplit), dim3(GEMM_THREADS), smem, s, G);
            HIPCHK(hipGetLastError());
        } else {                                   // wider hidden sizes: two plain launches (N needs several tiles)
            RET((launch_gemm<64, 64, 2, 2, true, true, XF_NONE, XF_NONE, false>(G.A, nox, G.EA, nullptr, 1, s, BSAREC_K_NONE, true)));
            RET((launch_gemm<64, 64, 2, 2, false, true, XF_NONE, XF_NONE, false>(G.B, nox, G.EB, nullptr, 1, s, BSAREC_K_NONE, true)));
        }
    }
    // the dense item-table gradient is complete (enqueued): a data-parallel host may start exchanging it now
    if (p->dense_hook && !g_dry && !p->ext_dy) p->dense_hook(p->dense_hook_user, stream);
    float* dY = (N & 1) ? p->dXb : p->dXa;       // gradient w.r.t. X[l+1]; ping-pong so that dX[0] lands in dXa
    if (p->ext_dy) {
        if (p->bf) {      // the block kernels read inter-block gradients as bf16: convert the caller's fp32 tensor once
            CastJobs6 J;
            memset(&J, 0, sizeof(J));
            J.src[0] = p->ext_dy; J.dst[0] = (unsigned short*)dY; J.n4[0] = (long)T * d / 4;
            LAUNCH(cast_bf16_kernel, dim3(cdiv((long)T * d / 4, ROW_THREADS), 1), dim3(ROW_THREADS), 0, s, J);
            HIPCHK(hipGetLastError());
        } else dY = const_cast<float*>(p->ext_dy);
    } else
    if (!p->fused) {      // the fused top-layer backward synthesises this gradient from the slabs itself
        LAUNCH(dlast_kernel, dim3(cdiv((long)T * d / 4, ROW_THREADS)), dim3(ROW_THREADS), 0, s, p->dlast_slab,
               p->loss_kind == 1 ? 1 : p->vsplit, (long)B * d, T, L, d, dY);
        HIPCHK(hipGetLastError());
    }

    DwP DW;                       // weight-gradient problems waiting for their launch (dw_direct.h)
    memset(&DW, 0, sizeof(DW));
    int dw_np = 0, dw_nu = 0;
    TopBwdP top_head;
    bool have_head = false;
    for (int l = N - 1; l >= 0; --l) {
        const bsarec_layer_t& w = p->P.layer[l];
        LayerBufs& b = p->lb[l];
        const float* X = p->X[l];
        float* dXout = (dY == p->dXa) ? p->dXb : p->dXa;
        p->slab_w = p->slab_wL[l]; p->slab_b = p->slab_bL[l]; p->part_ln = p->part_lnL[l]; p->part_beta = p->part_betaL[l];
        const bool top_pruned = p->fused && p->pruned && l == N - 1;
        if (top_pruned) {
            // rides in the next launch -- when this layer's weight-gradient products do too (the direct kernel defers them; the
            // tiled fallback launches them inside this iteration and needs the top block's operands now)
            if (N >= 2 && !c.separate_top && p->direct_dw) { fill_top_bwd(*p, l, tr, dXout, top_head); have_head = true; }
            else RET(launch_top_bwd(*p, l, tr, dXout, s));
        } else if (p->fused) {
            RET(launch_fused_bwd(*p, l, tr, dY, dXout, s, l == N - 1 && !p->ext_dy, (have_head && l == N - 2) ? &top_head : nullptr));
        } else {
        // ---- FeedForward backward
        {
            LnBranch a; memset(&a, 0, sizeof(a));
            a.xhat = b.xhat_ff; a.rstd = b.rstd_ff; a.gamma = w.ffn_ln_w; a.in_scale = 1.f;
            a.drop = make_drop(*p, c.p_hidden, 4 + 4 * l, tr); a.dT = p->dT;
            a.pgamma = p->part_ln + 0L * nb * d; a.pbeta = p->part_ln + 1L * nb * d;
            DISPATCH_LPR(d, LAUNCH((ln_bwd_kernel<LPR, 0>), dim3(nb), dim3(ROW_THREADS), 0, s, dY, a, a, p->dz, T, d, p->rows_pb));
            HIPCHK(hipGetLastError());
        }
        {   // dU = (dT2 . W2) * gelu'(U)
            GemmP g = gemm_defaults(T, 4 * d, d);
            g.lda = d; g.ldb = 4 * d; g.A[0] = p->dT; g.B[0] = w.ffn2_w;
            auto e = epi_linear<false, false, true>(p->dU, 4 * d);
            e.U = b.u; e.ldu = 4 * d; e.act = c.hidden_act;
            RET((launch_gemm<64, 64, 2, 2, false, true, XF_NONE, XF_NONE, false>(g, nox, e, nullptr, 1, s, BSAREC_K_DU)));
        }
        {   // dH = dU . W1 + dz
            GemmP g = gemm_defaults(T, d, 4 * d);
            g.lda = 4 * d; g.ldb = d; g.A[0] = p->dU; g.B[0] = w.ffn1_w;
            auto e = epi_linear<false, true, false>(p->dH, d);
            e.R = p->dz; e.ldr = d;
            RET((launch_gemm<64, 64, 2, 2, false, true, XF_NONE, XF_NONE, false>(g, nox, e, nullptr, 1, s)));
        }
        // ---- mix + the two LayerNorms (attention branch scaled by 1-alpha, filter branch by alpha)
        {
            LnBranch a; memset(&a, 0, sizeof(a));
            a.xhat = b.xhat_a; a.rstd = b.rstd_a; a.gamma = w.attn_ln_w; a.in_scale = (float)(1.0 - (double)c.alpha);
            a.drop = make_drop(*p, c.p_hidden, 3 + 4 * l, tr); a.dT = p->dO;
            a.pgamma = p->part_ln + 2L * nb * d; a.pbeta = p->part_ln + 3L * nb * d;
            LnBranch f; memset(&f, 0, sizeof(f));
            f.xhat = b.xhat_f; f.rstd = b.rstd_f; f.gamma = w.filter_ln_w; f.in_scale = c.alpha;
            f.drop = make_drop(*p, c.p_hidden, 1 + 4 * l, tr); f.dT = p->dF;
            f.pgamma = p->part_ln + 4L * nb * d; f.pbeta = p->part_ln + 5L * nb * d;
            DISPATCH_LPR(d, LAUNCH((ln_bwd_kernel<LPR, 1>), dim3(nb), dim3(ROW_THREADS), 0, s, p->dH, a, f, p->dXacc, T, d, p->rows_pb));
            HIPCHK(hipGetLastError());
        }
        // ---- attention backward
        {   // dC = dO . Wo
            GemmP g = gemm_defaults(T, d, d);
            g.lda = d; g.ldb = d; g.A[0] = p->dO; g.B[0] = w.dense_w;
            auto e = epi_linear<false, false, false>(p->dC, d);
            RET((launch_gemm<64, 64, 2, 2, false, true, XF_NONE, XF_NONE, false>(g, nox, e, nullptr, 1, s)));
        }
        XformP xfa = nox; xfa.drop = make_drop(*p, c.p_attn, 2 + 4 * l, tr); xfa.L = L; xfa.Lp = Lp;
        {   // dS = P * (dA - rowsum(dA P)) / sqrt(dh),  dA = (dC . V^T) * keep/(1-p)
            GemmP g = gemm_defaults(L, Lp, dh);
            g.Nb = L; g.lda = d; g.ldb = d; g.nh = h; g.A[0] = p->dC; g.B[0] = b.v;
            g.a_sb = (long)L * d; g.a_sh = dh; g.b_sb = (long)L * d; g.b_sh = dh;
            EpiDS e; e.P = b.probs; e.drop = xfa.drop; e.L = L; e.Lp = Lp; e.sqrt_dh = sqrtf((float)dh); e.dS = p->dS;
            DISPATCH_BN(Lp, RET((launch_gemm<64, BN, 2, 2, false, false, XF_NONE, XF_NONE, false>(g, nox, e, nullptr, B * h, s))));
        }
        {   // dV = Drop(P)^T . dC
            GemmP g = gemm_defaults(L, dh, L);
            g.lda = Lp; g.ldb = d; g.nh = h; g.A[0] = b.probs; g.B[0] = p->dC;
            g.a_sb = (long)h * L * Lp; g.a_sh = (long)L * Lp; g.b_sb = (long)L * d; g.b_sh = dh;
            auto e = epi_linear<false, false, false>(p->dv, d);
            e.c_sb = (long)L * d; e.c_sh = dh;
            RET((launch_gemm<64, 64, 2, 2, true, true, XF_DROP, XF_NONE, false>(g, xfa, e, nullptr, B * h, s)));
        }
        {   // dK = dS^T . Q
            GemmP g = gemm_defaults(L, dh, L);
            g.lda = Lp; g.ldb = d; g.nh = h; g.A[0] = p->dS; g.B[0] = b.q;
            g.a_sb = (long)h * L * Lp; g.a_sh = (long)L * Lp; g.b_sb = (long)L * d; g.b_sh = dh;
            auto e = epi_linear<false, false, false>(p->dk, d);
            e.c_sb = (long)L * d; e.c_sh = dh;
            RET((launch_gemm<64, 64, 2, 2, true, true, XF_NONE, XF_NONE, false>(g, nox, e, nullptr, B * h, s)));
        }
        {   // dQ = dS . K
            GemmP g = gemm_defaults(L, dh, Lp);
            g.Kv = L; g.lda = Lp; g.ldb = d; g.nh = h; g.A[0] = p->dS; g.B[0] = b.k;
            g.a_sb = (long)h * L * Lp; g.a_sh = (long)L * Lp; g.b_sb = (long)L * d; g.b_sh = dh;
            auto e = epi_linear<false, false, false>(p->dq, d);
            e.c_sb = (long)L * d; e.c_sh = dh;
            RET((launch_gemm<64, 64, 2, 2, false, true, XF_NONE, XF_NONE, false>(g, nox, e, nullptr, B * h, s)));
        }
        {   // dXtmp = dQ.Wq + dK.Wk + dV.Wv + (dzA + dzF)
            GemmP g = gemm_defaults(T, d, d);
            g.lda = d; g.ldb = d; g.nseg = 3;
            g.A[0] = p->dq; g.A[1] = p->dk; g.A[2] = p->dv;
            g.B[0] = w.query_w; g.B[1] = w.key_w; g.B[2] = w.value_w;
            auto e = epi_linear<false, true, false>(p->dXtmp, d);
            e.R = p->dXacc; e.ldr = d;
            RET((launch_gemm<64, 64, 2, 2, false, true, XF_NONE, XF_NONE, false>(g, nox, e, nullptr, 1, s)));
        }
        }
        {   // all six weight gradients + bias gradients of the block: one grouped split-K launch
            GroupedTN G;
            memset(&G, 0, sizeof(G));
            struct Spec { const float* A; long lda; const float* B; long ldb; int M, N; long woff, boff; int gelu; };
            const Spec sp[6] = {
                {p->dq, d, X, d, d, d, sm.wq, sm.bq, 0},        {p->dk, d, X, d, d, d, sm.wk, sm.bk, 0},
                {p->dv, d, X, d, d, d, sm.wv, sm.bv, 0},        {p->dO, d, b.ctx, d, d, d, sm.wo, sm.bo, 0},
                {p->dU, 4 * d, b.hmix, d, 4 * d, d, sm.w1, sm.b1, 0},
                // dW2 = dT2^T . act(u): the full fused forward already saved gelu(u) in `u`; elsewhere apply it while loading
                {p->dT, d, b.u, 4 * d, d, 4 * d, sm.w2, sm.b2, (p->fused && !top_pruned) ? 0 : 1}};
            int tiles = 0;
            // tile size of the grouped tiled kernel (the direct kernel has its own units): 128 for bf16 products at hidden >= 128 (C3:
            // 405 -> 337 us per launch); the fp32 form needs 296 registers at 128 x 128 = one wave per SIMD and loses (765 -> 987 us)
            const int gts = (p->bf_products && d >= 128) ? 128 : 64;
            // Top block: only position L-1 of each sequence carries an upstream gradient (bsarec.py:32), so dq, dO, dU
            // and dT2 are zero on every other row: their four products reduce over the B last positions only
            // (row stride L*ld), exactly; dk and dv still reduce over all tokens.
            const bool top = (l == N - 1) && !p->ext_dy;       // (an external upstream gradient has every row)
            for (int i = 0; i < 6; ++i) {
                const bool last_only = top && i != 1 && i != 2;
                GemmP g = gemm_defaults(sp[i].M, sp[i].N, last_only ? B : T);
                g.lda = sp[i].lda; g.ldb = sp[i].ldb; g.A[0] = sp[i].A; g.B[0] = sp[i].B; g.nsplit = ns; g.kchunk = p->kchunk;
                if (last_only) {
                    // row L-1 of every sequence (element offsets: the operands are bf16 tensors under storage = 1)
                    const long esz = p->bf ? 2 : 4;
                    g.A[0] = (const float*)((const char*)g.A[0] + (long)(L - 1) * sp[i].lda * esz);
                    g.B[0] = (const float*)((const char*)g.B[0] + (long)(L - 1) * sp[i].ldb * esz);
                    g.lda *= L; g.ldb *= L;
                    g.kchunk = (int)rup(cdiv(B, ns), GEMM_BK);           // slices beyond ceil(B / kchunk) write zero slabs
                    if (top_pruned) {                                    // gradient rows come compact ([B][M]) from top_bwd_kernel
                        g.A[0] = i == 0 ? p->top_dq : i == 3 ? p->top_dO : i == 4 ? p->top_dU : p->top_dT;
                        g.lda = sp[i].M;
                        g.nsplit = std::min(p->top_slabs, ns); g.kchunk = (int)rup(cdiv(B, g.nsplit), GEMM_BK);
                    }
                }
                // pruned top block: dK, dV are rank-1 per (sequence, head) -> dWk = AK^T RK, dWv = AV^T RV over B*h rows
                // (fused_top.h); their bias gradients come from per-sequence partials, the slab output is discarded
                const bool compact = top_pruned && (i == 1 || i == 2);
                if (compact) {
                    g = gemm_defaults(d, d, B * h);
                    g.A[0] = i == 1 ? p->top_ak : p->top_av; g.B[0] = i == 1 ? p->top_rk : p->top_rv; g.lda = d; g.ldb = d;
                    g.nsplit = std::min(p->top_slabs, ns); g.kchunk = (int)rup(cdiv(B * h, g.nsplit), GEMM_BK);
                }
                G.P[i] = g;
                G.E[i] = epi_linear<false, false, false>(slab_w_ptr(*p, sp[i].woff), sp[i].N);
                G.E[i].c_split = (long)sp[i].M * sp[i].N;
                G.bgrad[i] = compact ? p->slab_dummy : slab_b_ptr(*p, sp[i].boff);
                G.tile0[i] = tiles;
                G.tiles_n[i] = cdiv(sp[i].N, gts);
                G.b_gelu[i] = sp[i].gelu;
                tiles += cdiv(sp[i].M, gts) * G.tiles_n[i];
            }
            G.tile0[6] = tiles; G.nprob = 6; G.act = c.hidden_act;
            if (p->fused && p->direct_dw) {
                // hidden = 64: direct split-K products, one workgroup per (problem, 64x64 tile, slab slice) -- dw_direct.h.
                // The pruned top block's six (tiny) problems are not launched on their own: they wait in DW and ride in
                // the next block's launch.
                for (int i = 0; i < 6; ++i) {
                    const GemmP& g = G.P[i];
                    DwProblem& q = DW.P[dw_np];
                    q.A = g.A[0]; q.B = g.B[0]; q.lda = g.lda; q.ldb = g.ldb; q.M = g.M; q.N = g.N; q.K = g.K;
                    q.kchunk = g.kchunk; q.nslab = g.nsplit; q.slab = G.E[i].C[0]; q.bslab = G.bgrad[i]; q.gelu = G.b_gelu[i];
                    q.bf16 = p->bf ? 1 : 0;
                    for (int m0 = 0; m0 < q.M; m0 += 64)
                        for (int n0 = 0; n0 < q.N; n0 += 64) DW.U[dw_nu++] = DwUnit{(short)dw_np, (short)m0, (short)n0, 0};
                    ++dw_np;
                }
                if (top_pruned) { DW.nsmall = dw_nu; DW.small_slabs = std::min(p->top_slabs, ns); }
                else {
                    DW.nunits = dw_nu; DW.nslab = ns;
                    TickP tk_here;                     // the step tick rides in block 0's launch when Adam is fused into the reduction
                    memset(&tk_here, 0, sizeof(tk_here));
                    if (fuse_adam && l == 0) tk_here = tick;
                    ScatterP sc;                       // ... and so does the embedding-gradient scatter (block 0's launch: dz is complete)
                    memset(&sc, 0, sizeof(sc));
                    if (p->scatter_in_block && l == 0) {
                        sc.de = p->dz; sc.ids32 = p->ids32; sc.T = T; sc.dE = p->lookup_grad ? p->lookup_grad : p->G.item_emb;
                        sc.nblocks = cdiv(T, SCATTER_FLOATS / 64);
                    }
                    for (int i = 0; i < dw_np; ++i) if (!dw_problem_ok(DW.P[i])) return -21;
                    if (sc.nblocks && (!sc.de || !sc.ids32 || !sc.dE)) return -21;
                    ProfScope prof(BSAREC_K_DW1, s);
                    const dim3 dw_grid(8 * cdiv(ns, 8) * (dw_nu - DW.nsmall) + DW.nsmall * DW.small_slabs + sc.nblocks + (tk_here.state ? 1 : 0));
                    LAUNCH(dw_direct_kernel, dw_grid, dim3(256), 0, s, DW, tk_here, sc);
                    HIPCHK(hipGetLastError());
                    dw_np = 0; dw_nu = 0; DW.nsmall = 0; DW.small_slabs = 0;
                }
            } else {
            ProfScope prof(BSAREC_K_DW1, s);
#define GROUPED_LAUNCH(BFV, TSV) do { \
                constexpr size_t sm_ = GemmSmem<TSV, TSV, true, true, BFV>::BYTES; \
                static bool attr_ = false; \
                if (!attr_) { if (sm_ > 48 * 1024) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_grouped_tn_kernel<BFV, TSV>), \
                                                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_)); attr_ = true; } \
                LAUNCH((gemm_grouped_tn_kernel<BFV, TSV>), dim3(tiles, ns), dim3(GEMM_THREADS), sm_, s, G); } while (0)
            if (p->bf_products) { if (gts == 128) GROUPED_LAUNCH(true, 128); else GROUPED_LAUNCH(true, 64); }
            else GROUPED_LAUNCH(false, 64);
#undef GROUPED_LAUNCH
            HIPCHK(hipGetLastError());
            }
        }
        // ---- FrequencyLayer backward: completes dX of this layer
        if (!p->fused)
            DISPATCH_LPR(d, RET(launch_freq_bwd<LPR>(X, p->dF, p->dXtmp, w.sqrt_beta, p->twiddle, B, L, d, c.cutoff_bins,
                                                     dXout, p->part_beta, s, c.filter_kind == 1 ? w.filter_cw : nullptr,
                                                     c.filter_kind == 1 ? p->part_cwL[l] : nullptr)));
        dY = dXout;
        // forward(all_sequence_output=True): layer output l may carry an upstream gradient of its own -- it joins the gradient
        // coming down from the blocks above (the fused bottom block adds output 0's inside its epilogue: its dX never
        // reaches memory)
        if (p->ext_dy && p->ext_mid[l] && !(p->fused && l == 0)) {
            const long n4 = (long)T * d / 4;
            LAUNCH(grad_join_kernel, dim3((unsigned)std::min<long>(cdiv(n4, ROW_THREADS), 2048)), dim3(ROW_THREADS), 0, s, dXout,
                   p->ext_mid[l], n4, p->bf ? 1 : 0);
            HIPCHK(hipGetLastError());
        }
    }
    // ---- embedding front-end backward
    {
        hipStream_t se = s;
        LnBranch a; memset(&a, 0, sizeof(a));
        a.xhat = p->xhat0; a.rstd = p->rstd0; a.gamma = p->P.ln_w; a.in_scale = 1.f;
        a.drop = make_drop(*p, c.p_hidden, 0, tr); a.dT = nullptr;
        a.pgamma = p->part_ln0 + 0L * nb * d; a.pbeta = p->part_ln0 + 1L * nb * d;
        if (!p->fused) {                  // fused path: done by the bottom block's backward kernel
            DISPATCH_LPR(d, LAUNCH((ln_bwd_kernel<LPR, 2>), dim3(nb), dim3(ROW_THREADS), 0, se, dY, a, a, p->dz, T, d, p->rows_pb));
            HIPCHK(hipGetLastError());
        }
        if (!p->scatter_in_block)
        DISPATCH_LPR(d, {
            constexpr int CHUNK = SCATTER_FLOATS / (LPR * 4);
            constexpr size_t smem = SCATTER_FLOATS * 4 + 2 * CHUNK * 4;
            static bool attr = false;
            if (!attr) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(embed_bwd_kernel<LPR>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                attr = true;
            }
            const int sb = cdiv(T, CHUNK);
            LAUNCH(embed_bwd_kernel<LPR>, dim3(sb + L * p->pos_slices), dim3(ROW_THREADS), smem, se, p->dz, p->ids32, B, L, d,
                   p->lookup_grad ? p->lookup_grad : p->G.item_emb, p->part_pos, sb);
            HIPCHK(hipGetLastError());
        });
        // ---- ONE deterministic second-stage reduction for every split-K slab and LayerNorm / beta partial
        // (no empty blocks: flat block map); its extra last block closes the optimisation step when asked to
        if (fuse_adam) {
            const bsarec_adam_t& a = *fuse_adam;
            AdamFuseP A;
            memset(&A, 0, sizeof(A));
            A.w = a.params; A.g = const_cast<float*>(a.grads); A.m = a.exp_avg; A.v = a.exp_avg_sq;
            A.b1 = a.beta1; A.b2 = a.beta2; A.eps = a.eps; A.wd = a.weight_decay;
            A.shadow = (unsigned short*)a.shadow_bf16; A.shadow_from = a.shadow_bf16 ? a.shadow_from : a.n;
            A.item_off = p->G.item_emb - a.grads; A.item_n4 = (long)c.item_size * d / 4;
            int ab = cdiv(A.item_n4, ROW_THREADS);
            if (ab > 1024) ab = 1024;
            LAUNCH(reduce_adam_kernel, dim3(p->red_blocks + ab), dim3(ROW_THREADS), 0, s, p->pruned ? p->jobs_pruned : p->jobs,
                   p->blockmap, p->red_blocks, (const uint64_t*)p->state, A);
        } else
        LAUNCH(multi_reduce_flat_kernel, dim3(p->red_blocks + (tick.state ? 1 : 0)), dim3(ROW_THREADS), 0, s,
               p->pruned ? p->jobs_pruned : p->jobs, p->blockmap, p->red_blocks, tick);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Adam / fused step
// ---------------------------------------------------------------------------------------------
static TickP make_tick(void* state, int adam, float lr, float b1, float b2, const float* loss_rows, int B, float* loss_out,
                       void* cursor, int advance, int bump_step) {
    TickP t;
    memset(&t, 0, sizeof(t));
    t.state = (uint64_t*)state; t.adam = adam; t.lr = lr; t.b1 = b1; t.b2 = b2;
    t.loss_rows = loss_rows; t.B = B; t.loss_out = loss_out;
    t.cursor = (long long*)cursor; t.advance = advance; t.bump_step = bump_step;
    return t;
}

static int adam_check(const bsarec_adam_t* a) {
    if (!a || !a->params || !a->grads || !a->exp_avg || !a->exp_avg_sq || a->n <= 0 || (a->n & 3)) return -10;
    if (a->shadow_bf16 && (a->shadow_from < 0 || (a->shadow_from & 3))) return -10;
    if (a->grads2 && (a->grads2_n < 0 || a->grads2_n > a->n || (a->grads2_n & 3))) return -10;
    if (a->n_grad_srcs < 0 || a->n_grad_srcs > 8) return -10;
    for (int r = 0; r < a->n_grad_srcs; ++r) if (!a->grad_srcs[r]) return -10;
    return 0;
}

static int adam_launch(const bsarec_adam_t& a, void* state, hipStream_t s) {
    const long n4 = a.n / 4;
    int blocks = cdiv(n4, ROW_THREADS);
    if (blocks > 2048) blocks = 2048;
    GradSrcs S;
    memset(&S, 0, sizeof(S));
    S.nsrc = a.n_grad_srcs;
    for (int r = 0; r < a.n_grad_srcs; ++r) S.src[r] = a.grad_srcs[r];
    S.g2 = a.grads2; S.n2_4 = a.grads2 ? a.grads2_n / 4 : 0;
    LAUNCH(adam_kernel, dim3(blocks), dim3(ROW_THREADS), 0, s, a.params, a.grads, a.exp_avg, a.exp_avg_sq, n4,
           (const uint64_t*)state, a.beta1, a.beta2, a.eps, a.weight_decay, a.grad_scale, (unsigned short*)a.shadow_bf16,
           a.shadow_bf16 ? a.shadow_from / 4 : n4, S);
    return (int)hipGetLastError();
}

extern "C" int bsarec_adam_step(const bsarec_adam_t* a, void* state, void* stream) {
    RET(adam_check(a));
    if (!state) return -10;
    hipStream_t s = (hipStream_t)stream;
    LAUNCH(adam_tick_kernel, dim3(1), dim3(ROW_THREADS), 0, s, make_tick(state, 1, a->lr, a->beta1, a->beta2, nullptr, 0, nullptr, nullptr, 0, 0));
    HIPCHK(hipGetLastError());
    return adam_launch(*a, state, s);
}

extern "C" int bsarec_adam_apply(const bsarec_adam_t* a, void* state, void* stream) {
    RET(adam_check(a));
    if (!state) return -10;
    return adam_launch(*a, state, (hipStream_t)stream);
}

extern "C" int bsarec_gather_batch(const int64_t* table, const int64_t* answers_table, const int64_t* perm, long n_samples,
                                   const void* cursor, int B, int L, int64_t* ids_out, int64_t* answers_out, void* stream) {
    if (!table || !answers_table || !perm || !cursor || !ids_out || !answers_out || B < 1 || L < 1) return -10;
    LAUNCH(gather_batch_kernel, dim3(cdiv((long)B * L, ROW_THREADS)), dim3(ROW_THREADS), 0, (hipStream_t)stream, table,
           answers_table, perm, n_samples, (const long long*)cursor, B, L, ids_out, answers_out);
    return (int)hipGetLastError();
}

extern "C" int bsarec_train_step_indexed(bsarec_plan_t* p, const int64_t* table, const int64_t* answers_table,
                                         const int64_t* perm, long n_samples, void* cursor, int64_t* ids_buf,
                                         int64_t* answers_buf, const bsarec_adam_t* a, void* stream) {
    if (!p) return -10;
    RET(adam_check(a));
    hipStream_t s = (hipStream_t)stream;
    if (!table || !answers_table || !perm || !cursor || !ids_buf || !answers_buf) return -10;
    GatherP gp{table, answers_table, perm, n_samples, (const long long*)cursor, ids_buf, answers_buf};
    RET(forward_impl(p, ids_buf, 1, stream, gp, true));       // batch assembly rides in the embedding kernel
    RET(loss_impl(p, answers_buf, stream, false));
    // the extra block of the final gradient reduction closes the step: mean loss, Adam t and bias corrections, next
    // forward-step index, cursor += B
    const TickP tk = make_tick(p->state, 1, a->lr, a->beta1, a->beta2, p->loss_rows, p->cfg.batch, p->loss, cursor, p->cfg.batch, 1);
    if (can_fuse_adam(*p, *a) && !p->cfg.separate_embed)       // 7 launches: the last one reduces and updates
        return backward_impl(p, stream, tk, a);
    RET(backward_impl(p, stream, tk));
    return adam_launch(*a, p->state, s);
}

extern "C" int bsarec_grad_step_indexed(bsarec_plan_t* p, const int64_t* table, const int64_t* answers_table,
                                        const int64_t* perm, long n_samples, void* cursor, int64_t* ids_buf,
                                        int64_t* answers_buf, float lr, float b1, float b2, void* stream) {
    if (!p || !table || !answers_table || !perm || !cursor || !ids_buf || !answers_buf) return -10;
    GatherP gp{table, answers_table, perm, n_samples, (const long long*)cursor, ids_buf, answers_buf};
    RET(forward_impl(p, ids_buf, 1, stream, gp, true));
    RET(bsarec_loss(p, answers_buf, stream));
    // same convention as bsarec_train_step_indexed: the step index / cursor advance when the step is done
    return backward_impl(p, stream, make_tick(p->state, lr > 0.f ? 1 : 0, lr, b1, b2, nullptr, 0, nullptr, cursor, p->cfg.batch, 1));
}

extern "C" int bsarec_train_step(bsarec_plan_t* p, const int64_t* ids, const int64_t* answers, const bsarec_adam_t* a,
                                 void* stream) {
    if (!p) return -10;
    RET(adam_check(a));
    RET(bsarec_step_begin(p, stream));
    RET(bsarec_forward_last(p, ids, 1, stream));
    RET(bsarec_loss(p, answers, stream));
    RET(bsarec_backward(p, stream));
    return bsarec_adam_step(a, p->state, stream);
}

extern "C" int bsarec_mask_seen(float* scores, long ld, int B, const int64_t* users, const int64_t* indptr,
                                const int64_t* indices, void* stream) {
    if (!scores || !users || !indptr || !indices || B < 1 || ld < 1) return -10;
    hipLaunchKernelGGL(mask_seen_kernel, dim3(B), dim3(ROW_THREADS), 0, (hipStream_t)stream, scores, ld, users, indptr, indices);
    return (int)hipGetLastError();
}

extern "C" int bsarec_topk_seen(float* scores, long ld, int B, int V, const int64_t* users, const int64_t* indptr,
                                const int64_t* indices, int k, int64_t* out_idx, float* out_val, void* stream) {
    if (!scores || !out_idx || B < 1 || V < 1 || ld < V || k < 1 || k > TOPK_MAX || k > V) return -10;
    if (indptr && (!users || !indices)) return -10;
    hipLaunchKernelGGL(topk_seen_kernel, dim3(B), dim3(ROW_THREADS), 0, (hipStream_t)stream, scores, ld, V, users, indptr, indices, k,
                       out_idx, out_val);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// stand-alone FrequencyLayer (per-op parity tests)
// ---------------------------------------------------------------------------------------------
static DropP standalone_drop(float p, const void* state, int site) {
    DropP d;
    d.thresh = drop_thresh(p);
    d.scale = p > 0.f ? (float)(1.0 / (1.0 - (double)p)) : 1.f;
    d.rng = (const uint64_t*)state; d.site = (uint32_t)site;
    return d;
}

extern "C" int bsarec_freq_layer_fwd(const float* x, const float* sqrt_beta, const float* ln_w, const float* ln_b,
                                     const float* twiddle, int B, int L, int d, int cb, float eps, float p_drop,
                                     const void* state, int site, float* y, float* xhat, float* rstd, void* stream) {
    if (!x || !y || d % 4 || d > 256 || L > 256 || (long)cb * d > 8192 || (p_drop > 0.f && !state)) return -10;
    DISPATCH_LPR(d, RET(launch_freq_fwd<LPR>(x, sqrt_beta, ln_w, ln_b, eps, standalone_drop(p_drop, state, site), twiddle,
                                             B, L, d, cb, y, xhat, rstd, (hipStream_t)stream)));
    return 0;
}

extern "C" long bsarec_freq_layer_bwd_scratch_floats(int B, int L, int d) {
    const long T = (long)B * L;
    return 3 * T * d + (long)B * d + 2L * cdiv(T, 64) * d + 64;
}

extern "C" int bsarec_freq_layer_bwd(const float* x, const float* dy, const float* xhat, const float* rstd,
                                     const float* sqrt_beta, const float* ln_w, const float* twiddle, int B, int L, int d,
                                     int cb, float p_drop, const void* state, int site, float* scratch, float* dx,
                                     float* dsqrt_beta, float* dln_w, float* dln_b, void* stream) {
    if (!x || !dy || !scratch || d % 4 || d > 256 || L > 256 || (long)cb * d > 8192 || (p_drop > 0.f && !state)) return -10;
    hipStream_t s = (hipStream_t)stream;
    const int T = B * L, nb = cdiv(T, 64);
    float* dz = scratch; float* dF = dz + (long)T * d; float* pbeta = dF + (long)T * d;
    float* pg = pbeta + (long)B * d; float* pb = pg + (long)nb * d;
    LnBranch a; memset(&a, 0, sizeof(a));
    a.xhat = xhat; a.rstd = rstd; a.gamma = ln_w; a.in_scale = 1.f; a.drop = standalone_drop(p_drop, state, site);
    a.dT = dF; a.pgamma = pg; a.pbeta = pb;
    DISPATCH_LPR(d, LAUNCH((ln_bwd_kernel<LPR, 0>), dim3(nb), dim3(ROW_THREADS), 0, s, dy, a, a, dz, T, d, 64));
    HIPCHK(hipGetLastError());
    // y = LN(Drop(f(x)) + x): the residual contributes dz directly
    DISPATCH_LPR(d, RET(launch_freq_bwd<LPR>(x, dF, dz, sqrt_beta, twiddle, B, L, d, cb, dx, pbeta, s)));
    ReduceJobs3 hj;                      // the three jobs travel in the kernarg block: no staging copy, no synchronisation
    hj.j[0] = ReduceJob{pbeta, dsqrt_beta, B, d, d, 1.f, 0};
    hj.j[1] = ReduceJob{pg, dln_w, nb, d, d, 1.f, 0};
    hj.j[2] = ReduceJob{pb, dln_b, nb, d, d, 1.f, 0};
    LAUNCH(multi_reduce3_kernel, dim3(cdiv(d, 64), 3), dim3(ROW_THREADS), 0, s, hj);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// profiling hooks
// ---------------------------------------------------------------------------------------------
extern "C" int bsarec_profile_select(bsarec_plan_t* p, int kclass) {
    if (!p) return -10;
    p->prof.kclass = kclass;
    p->prof.used = 0;
    return 0;
}

extern "C" int bsarec_profile_read(bsarec_plan_t* p, double* ms_total, int* launches) {
    if (!p) return -10;
    ProfState& ps = p->prof;
    double tot = 0.0;
    for (size_t i = 0; i < ps.used; ++i) {
        HIPCHK(hipEventSynchronize(ps.events[i].second));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ps.events[i].first, ps.events[i].second));
        tot += ms;
    }
    if (ms_total) *ms_total = tot;
    if (launches) *launches = (int)ps.used;
    ps.used = 0;
    return 0;
}

extern "C" int bsarec_profile_event_overhead(void* stream, int reps, double* ms_avg) {
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    double tot = 0.0;
    for (int i = 0; i < reps; ++i) {
        HIPCHK(hipEventRecord(a, s)); HIPCHK(hipEventRecord(b, s));
        HIPCHK(hipEventSynchronize(b));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, a, b));
        tot += ms;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if (ms_avg) *ms_avg = reps > 0 ? tot / reps : 0.0;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// catalogue-sharded head (include/bsarec_shard.h, catalogue_shard.h)
// ---------------------------------------------------------------------------------------------
static int shard_ptrs(ShardPtrs& S, const float* const* p, int world) {
    if (!p || world < 1 || world > 8) return -10;
    memset(&S, 0, sizeof(S));
    for (int r = 0; r < world; ++r) { if (!p[r]) return -10; S.p[r] = p[r]; }
    return 0;
}

extern "C" int bsarec_shard_gather_rows(const int64_t* ids, long n, const float* const* shards, int world, long rows_per,
                                        long V, int d, float* stage, int64_t* local_ids, void* stream) {
    if (!ids || !stage || !local_ids || n < 1 || rows_per < 1 || V < 1 || d < 4 || (d & 3)) return -10;
    if ((V + rows_per - 1) / rows_per > world) return -11;
    ShardPtrs S;
    RET(shard_ptrs(S, shards, world));
    const int d4 = d / 4;
    LAUNCH(shard_gather_rows_kernel, dim3(cdiv((n + 1) * d4, 256)), dim3(256), 0, (hipStream_t)stream, ids, n, S, rows_per, V, d4,
           stage, local_ids);
    return (int)hipGetLastError();
}

extern "C" int bsarec_shard_logits(const float* h, long ldh, int Bg, const float* E, int Vs, int d, float* logits, long ld,
                                   void* stream) {
    if (!h || !logits || Bg < 1 || Vs < 0 || d < 4 || (d & 3) || ld < Vs || (ld & 3) || ldh < d) return -10;
    if (Vs == 0) return 0;
    if (!E) return -10;
    GemmP g = gemm_defaults(Bg, (int)ld, d);
    g.Nb = Vs; g.lda = ldh; g.ldb = d;
    g.A[0] = h; g.B[0] = E;
    auto e = epi_linear<false, false, false>(logits, ld);
    return launch_gemm<64, 64, 2, 2, false, false, XF_NONE, XF_NONE, false>(g, no_xform(), e, nullptr, 1, (hipStream_t)stream);
}

extern "C" int bsarec_shard_ce_stats(const float* logits, long ld, int Bg, int Vs, const int64_t* answers, long lo, long V,
                                     float* stats, void* stream) {
    if (!logits || !answers || !stats || Bg < 1 || Vs < 0 || ld < Vs || (ld & 3)) return -10;
    LAUNCH(shard_ce_stats_kernel, dim3(Bg), dim3(ROW_THREADS), 0, (hipStream_t)stream, logits, ld, Vs, answers, lo, V, stats, Bg);
    return (int)hipGetLastError();
}

extern "C" int bsarec_shard_ce_grad(float* logits, long ld, int Bg, int Vs, const int64_t* answers, long lo, long V,
                                    const float* stats_all, int world, float* loss_rows, float* loss, void* stream) {
    if (!logits || !answers || !stats_all || !loss_rows || Bg < 1 || Vs < 0 || ld < Vs || (ld & 3) || world < 1 || world > 8)
        return -10;
    hipStream_t s = (hipStream_t)stream;
    LAUNCH(shard_ce_grad_kernel, dim3(Bg), dim3(ROW_THREADS), 0, s, logits, ld, Vs, answers, lo, V, stats_all, world, Bg,
           1.0f / (float)Bg, loss_rows);
    HIPCHK(hipGetLastError());
    if (loss) LAUNCH(loss_mean_kernel, dim3(1), dim3(ROW_THREADS), 0, s, loss_rows, Bg, loss);
    return (int)hipGetLastError();
}

// split-K over the owned rows for d h_last: enough slices to fill the chip at small Bg, chunks of >= 4096 rows
static void shard_split(int Bg, int Vs, int d, int* nsplit, int* kchunk) {
    const long tiles = (long)cdiv(Bg, 64) * cdiv(d, 64);
    long want = (1024 + tiles - 1) / tiles;
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    long ch = rup(cdiv(Vs > 0 ? Vs : 1, want), GEMM_BK);
    if (ch < 256) ch = 256;
    *kchunk = (int)ch;
    *nsplit = cdiv(Vs > 0 ? Vs : 1, ch);
}

extern "C" long bsarec_shard_head_bwd_scratch_floats(int Bg, int Vs, int d) {
    if (Bg < 1 || Vs < 0 || d < 4) return -10;
    int ns, kc;
    shard_split(Bg, Vs, d, &ns, &kc);
    return (long)ns * Bg * d;
}

extern "C" int bsarec_shard_head_bwd(const float* dlogits, long ld, int Bg, int Vs, const float* h, long ldh, const float* E,
                                     int d, float* dE, float* dh, float* scratch, void* stream) {
    if (!dlogits || !h || !dh || !scratch || Bg < 1 || Vs < 0 || d < 4 || (d & 3) || ld < Vs || (ld & 3) || ldh < d) return -10;
    hipStream_t s = (hipStream_t)stream;
    if (Vs == 0) { if (!g_dry) HIPCHK(hipMemsetAsync(dh, 0, (size_t)Bg * d * sizeof(float), s)); return 0; }
    if (!E || !dE) return -10;
    const XformP nox = no_xform();
    {       // dE[v, :] = sum_b dlogits[b, v] h[b, :]
        GemmP g = gemm_defaults(Vs, d, Bg);
        g.lda = ld; g.ldb = ldh; g.A[0] = dlogits; g.B[0] = h;
        auto e = epi_linear<false, false, false>(dE, d);
        RET((launch_gemm<64, 64, 2, 2, true, true, XF_NONE, XF_NONE, false>(g, nox, e, nullptr, 1, s)));
    }
    int ns, kc;
    shard_split(Bg, Vs, d, &ns, &kc);
    {       // dh[b, :] = sum_v dlogits[b, v] E[v, :]
        GemmP g = gemm_defaults(Bg, d, (int)ld);
        g.Kv = Vs; g.lda = ld; g.ldb = d; g.A[0] = dlogits; g.B[0] = E;
        g.nsplit = ns; g.kchunk = kc;
        auto e = epi_linear<false, false, false>(scratch, d);
        e.c_split = (long)Bg * d;
        RET((launch_gemm<64, 64, 2, 2, false, true, XF_NONE, XF_NONE, false>(g, nox, e, nullptr, 1, s)));
    }
    const long n4 = (long)Bg * d / 4;
    LAUNCH(shard_slab_sum_kernel, dim3(cdiv(n4, ROW_THREADS)), dim3(ROW_THREADS), 0, s, scratch, ns, n4, dh);
    return (int)hipGetLastError();
}

extern "C" int bsarec_shard_scatter_rows(const int64_t* ids_all, long n, int world, const float* const* stage_grads, long lo,
                                         long Vs, long V, int d, float* dE, void* stream) {
    if (!ids_all || n < 1 || Vs < 0 || d < 4 || (d & 3)) return -10;
    if (Vs == 0) return 0;
    if (!dE) return -10;
    ShardPtrs G;
    RET(shard_ptrs(G, stage_grads, world));
    const int d4 = d / 4;
    LAUNCH(shard_scatter_rows_kernel, dim3(cdiv(n * world * d4, 256)), dim3(256), 0, (hipStream_t)stream, ids_all, n, world, G, lo,
           Vs, V, d4, dE);
    return (int)hipGetLastError();
}
