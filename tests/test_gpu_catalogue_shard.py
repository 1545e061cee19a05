"""Catalogue-sharded head (include/bsarec_shard.h, bsarec_amd/catalogue.py; SURVEY 8e, the C5 variant).

1. The stand-alone entry points against the oracle's arithmetic restated with torch on the same inputs, with the W shards of one
   table held by ONE process (the kernels only see pointers): the W partial heads together must reproduce the unsharded
   logits / CrossEntropyLoss / gradients of src/model/bsarec.py:32-35.
2. Two ranks on one GPU (gloo control plane, hipIpc mappings between the two processes -- the real data path): three
   sharded steps must equal three steps of ONE process that holds the full table and trains on the global batch.
"""
import argparse
import ctypes as C
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _ptrs8(L, tensors):
    return L.PTRS8(*([t.data_ptr() for t in tensors] + [None] * (8 - len(tensors))))


@pytest.mark.parametrize("V,W,Bg,d", [(301, 2, 96, 64), (1000, 3, 40, 128), (37, 8, 16, 64), (5, 8, 8, 64)])
def test_sharded_head_pieces_equal_the_unsharded_head(V, W, Bg, d):
    from bsarec_amd import _lib as L
    lib = L.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(V * 7 + W)
    E = (torch.randn(V, d, generator=g) * 0.3).to(dev)
    h = torch.randn(Bg, d, generator=g).to(dev)
    ans = torch.randint(0, V, (Bg,), generator=g).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    rows_per = (V + W - 1) // W
    shards = []
    for r in range(W):
        t = torch.zeros(rows_per, d, device=dev)
        lo = r * rows_per
        vs = max(0, min(rows_per, V - lo))
        if vs:
            t[:vs] = E[lo:lo + vs]
        shards.append(t)
    # ---- lookup rows
    n = 50
    ids = torch.randint(0, V, (n,), generator=g).to(dev)
    ids[::7] = 0
    stage = torch.full((n + 1, d), 7.0, device=dev)
    local = torch.zeros(n, dtype=torch.int64, device=dev)
    p8 = _ptrs8(L, shards)
    L.check(lib.bsarec_shard_gather_rows(ids.data_ptr(), n, C.byref(p8), W, rows_per, V, d, stage.data_ptr(),
                                         local.data_ptr(), st), "gather")
    torch.testing.assert_close(stage[0], E[0], rtol=0, atol=0)
    nz = ids > 0
    torch.testing.assert_close(stage[1:][nz], E[ids[nz]], rtol=0, atol=0)
    assert torch.equal(local, torch.where(nz, torch.arange(1, n + 1, device=dev), torch.zeros_like(ids)))
    # ---- head forward: statistics of every shard, then the combined loss
    want_logits = h.double() @ E.double().T
    want_lsm = torch.log_softmax(want_logits, dim=1)
    want_rows = -want_lsm[torch.arange(Bg), ans]
    stats_all = torch.zeros(W, 3, Bg, device=dev)
    logits, lds = [], []
    for r in range(W):
        lo = r * rows_per
        vs = max(0, min(rows_per, V - lo))
        ld = (max(vs, 1) + 3) // 4 * 4
        lg = torch.full((Bg, ld), 3.0, device=dev)
        L.check(lib.bsarec_shard_logits(h.data_ptr(), d, Bg, shards[r].data_ptr(), vs, d, lg.data_ptr(), ld, st), "logits")
        if vs:
            torch.testing.assert_close(lg[:, :vs].double(), want_logits[:, lo:lo + vs], rtol=1e-5, atol=1e-5)
        L.check(lib.bsarec_shard_ce_stats(lg.data_ptr(), ld, Bg, vs, ans.data_ptr(), lo, V, stats_all[r].data_ptr(), st), "stats")
        logits.append(lg); lds.append(ld)
    loss_rows = torch.zeros(Bg, device=dev)
    loss = torch.zeros(1, device=dev)
    dE_want = ((torch.softmax(want_logits, 1) - torch.nn.functional.one_hot(ans, V)) / Bg).T @ h.double()
    dh_want = ((torch.softmax(want_logits, 1) - torch.nn.functional.one_hot(ans, V)) / Bg) @ E.double()
    dh_sum = torch.zeros(Bg, d, device=dev, dtype=torch.float64)
    for r in range(W):
        lo = r * rows_per
        vs = max(0, min(rows_per, V - lo))
        L.check(lib.bsarec_shard_ce_grad(logits[r].data_ptr(), lds[r], Bg, vs, ans.data_ptr(), lo, V, stats_all.data_ptr(), W,
                                         loss_rows.data_ptr(), loss.data_ptr(), st), "grad")
        torch.testing.assert_close(loss_rows.double(), want_rows, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(loss[0].double(), want_rows.mean(), rtol=1e-5, atol=1e-5)
        if lds[r] > vs:
            assert float(logits[r][:, vs:].abs().max()) == 0.0          # pad columns zeroed
        dE = torch.full((rows_per, d), 9.0, device=dev)
        dh = torch.full((Bg, d), 9.0, device=dev)
        scratch = torch.zeros(max(1, lib.bsarec_shard_head_bwd_scratch_floats(Bg, vs, d)), device=dev)
        L.check(lib.bsarec_shard_head_bwd(logits[r].data_ptr(), lds[r], Bg, vs, h.data_ptr(), d, shards[r].data_ptr(), d,
                                          dE.data_ptr(), dh.data_ptr(), scratch.data_ptr(), st), "head_bwd")
        if vs:
            torch.testing.assert_close(dE[:vs].double(), dE_want[lo:lo + vs], rtol=1e-4, atol=1e-6)
        dh_sum += dh.double()
    torch.testing.assert_close(dh_sum, dh_want, rtol=1e-4, atol=1e-6)
    # ---- lookup-path gradient rows pulled by the owners
    ids_all = torch.randint(0, V, (W, n), generator=g).to(dev)
    ids_all[:, ::5] = 0
    grads = [torch.randn(n + 1, d, generator=g).to(dev) for _ in range(W)]
    want = torch.zeros(V, d, device=dev, dtype=torch.float64)
    for r in range(W):
        want.index_add_(0, ids_all[r], grads[r][1:].double())
    want[0] = 0
    g8 = _ptrs8(L, grads)
    for r in range(W):
        lo = r * rows_per
        vs = max(0, min(rows_per, V - lo))
        dE = torch.zeros(rows_per, d, device=dev)
        L.check(lib.bsarec_shard_scatter_rows(ids_all.data_ptr(), n, W, C.byref(g8), lo, vs, V, d, dE.data_ptr(), st), "scatter")
        if vs:
            torch.testing.assert_close(dE[:vs].double(), want[lo:lo + vs], rtol=1e-5, atol=1e-5)
        if rows_per > vs:
            assert float(dE[vs:].abs().max()) == 0.0


def _ns(**kw):
    a = argparse.Namespace(item_size=301, hidden_size=64, max_seq_length=50, batch_size=32, hidden_dropout_prob=0.0,
                           attention_probs_dropout_prob=0.0, num_hidden_layers=2, num_attention_heads=2,
                           hidden_act="gelu", initializer_range=0.02, c=3, alpha=0.9, seed=42, lr=1e-3,
                           adam_beta1=0.9, adam_beta2=0.999, weight_decay=0.0, no_cuda=False, log_freq=1)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _batches(ns, steps, Bg):
    g = torch.Generator(device="cpu").manual_seed(5)
    V, Lq = ns.item_size, ns.max_seq_length
    out = []
    for _ in range(steps):
        ids = torch.randint(1, V, (Bg, Lq), generator=g)
        pad = torch.randint(0, Lq - 2, (Bg,), generator=g)
        ids[torch.arange(Lq)[None, :] < pad[:, None]] = 0            # left padding, as the reference's dataset
        out.append((ids, torch.randint(1, V, (Bg,), generator=g)))
    return out


def _seen(ns, Bg):
    g = torch.Generator(device="cpu").manual_seed(8)
    seen = torch.randint(1, ns.item_size, (Bg, 30), generator=g)
    seen[:, 25:] = -1
    return seen


def _full_model(ns):
    from bsarec_amd import BSARecModel
    torch.manual_seed(3)
    return BSARecModel(ns).cuda()


def _worker(rank, world, port, kw, out_dir):
    import torch.distributed as dist
    from bsarec_amd.catalogue import ShardedCatalogue
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        ns = _ns(**kw)
        B = ns.batch_size
        sc = ShardedCatalogue(ns, B, dist.group.WORLD, "cuda:0")
        sc.load_full_state_dict(_full_model(ns).state_dict())
        losses = []
        for ids, ans in _batches(ns, 3, world * B):
            losses.append(float(sc.train_step(ids[rank * B:(rank + 1) * B], ans[rank * B:(rank + 1) * B])))
        assert not sc.px.timed_out()
        # evaluation over the sharded table: top-20 with the "seen" items zeroed
        ids, _ = _batches(ns, 1, world * B)[0]
        seen = _seen(ns, world * B)
        tv, ti = sc.topk(ids[rank * B:(rank + 1) * B], 20, seen[rank * B:(rank + 1) * B])
        # HR / NDCG bookkeeping over the sharded table (answers = each sequence's own best unseen item -> HR@5 = 1, and a second
        # set of answers that nobody ranks: item 0 is seen-or-low): every rank must report the GLOBAL metrics
        mine = slice(rank * B, (rank + 1) * B)
        ans_hit = ti[:, 2].clone()                                   # rank 2 of every list: HR@5/10/20 = 1, NDCG = 1/log2(4) = 0.5
        vals, txt = sc.full_sort_scores([(ids[mine], ans_hit, seen[mine])], epoch=3)
        assert vals[0] == 1.0 and vals[2] == 1.0 and vals[4] == 1.0 and abs(vals[1] - 0.5) < 1e-12 and abs(vals[5] - 0.5) < 1e-12, vals
        assert txt.startswith("{'Epoch': 3, 'HR@5': '1.0000', 'NDCG@5': '0.5000'")
        ans_mixed = torch.where(torch.arange(B, device=ti.device) % 2 == 0, ti[:, 0], ti[:, 19])      # even rows best item, odd rows the 20th
        if rank % 2 == 1:
            ans_mixed = ti[:, 7]                                     # odd ranks: 8th place -> in HR@10 / HR@20 only
        vals2, _ = sc.full_sort_scores([(ids[mine], ans_mixed, seen[mine])])
        np.savez(os.path.join(out_dir, f"topk{rank}.npz"), v=tv.cpu().numpy(), i=ti.cpu().numpy(), vals2=np.asarray(vals2),
                 ans_mixed=ans_mixed.cpu().numpy())
        sd = {k: v.detach().cpu().numpy() for k, v in sc.full_state_dict().items()}
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), losses=np.asarray(losses), **sd)
        sc.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kw", [(2, dict()), (2, dict(hidden_size=128, max_seq_length=64, num_attention_heads=4, item_size=1003, c=9)),
                                      (3, dict(item_size=302, batch_size=16))],
                         ids=["W2_fused_d64_L50", "W2_generic_d128_L64", "W3_uneven_shards"])
def test_ranks_sharded_catalogue_equals_the_full_table_step(world, kw, tmp_path):
    """(W3: 302 rows over 3 ranks = 101 + 101 + 100 -- a shorter last shard.)"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(world, port, kw, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    for r in range(1, world):
        rr = np.load(tmp_path / f"rank{r}.npz")
        for k in r0.files:
            np.testing.assert_array_equal(r0[k], rr[k], err_msg=k)      # encoder replicas bit-identical; same gathered table
    ns = _ns(**kw)
    model = _full_model(ns)
    model.configure_adam(lr=ns.lr, betas=(ns.adam_beta1, ns.adam_beta2), weight_decay=ns.weight_decay)
    model.train()
    losses = [float(model.train_step(ids.cuda(), ans.cuda())) for ids, ans in _batches(ns, 3, world * ns.batch_size)]
    np.testing.assert_allclose(r0["losses"], losses, atol=2e-4)
    # the sharded top-20 (taken after the three steps) against the full table of the same run: the gathered table of rank 0
    # IS the sharded model's table, so load it into a full model and score there
    B = ns.batch_size
    full = _full_model(ns)
    full.load_state_dict({k: torch.from_numpy(r0[k]) for k in r0.files if k != "losses"})
    full.eval()
    ids, _ = _batches(ns, 1, world * B)[0]
    seen = _seen(ns, world * B)
    with torch.no_grad():
        scores = full.full_logits(ids.cuda()).clone()
    rows = torch.arange(world * B).view(-1, 1).expand_as(seen)
    ok = seen >= 0
    scores[rows[ok].cuda(), seen[ok].cuda()] = 0.0
    want_v, want_i = torch.topk(scores, 20, dim=1)
    for r in range(world):
        t = np.load(tmp_path / f"topk{r}.npz")
        wv, wi = want_v[r * B:(r + 1) * B].cpu().numpy(), want_i[r * B:(r + 1) * B].cpu().numpy()
        np.testing.assert_allclose(t["v"], wv, rtol=1e-5, atol=1e-6)
        assert (t["i"] == wi).mean() > 0.99                       # (equal scores may swap places)
    # the metric bookkeeping: every rank reported the same numbers, and they are the reference's formulas (src/metrics.py:3-31)
    # over ALL ranks' sequences
    allv = [np.load(tmp_path / f"topk{r}.npz")["vals2"] for r in range(world)]
    for v in allv[1:]:
        np.testing.assert_array_equal(allv[0], v)
    hits = np.concatenate([np.load(tmp_path / f"topk{r}.npz")["i"] == np.load(tmp_path / f"topk{r}.npz")["ans_mixed"][:, None] for r in range(world)])
    want_m = []
    for kk in (5, 10, 20):
        want_m += [hits[:, :kk].any(1).mean(), (hits[:, :kk] / np.log2(np.arange(kk) + 2.0)).sum(1).mean()]
    np.testing.assert_allclose(allv[0], want_m, rtol=1e-12, atol=1e-12)
    sd = model.state_dict()
    assert set(sd) == set(r0.files) - {"losses"}
    for k in sd:
        got, want = r0[k], sd[k].detach().cpu().numpy()
        assert got.shape == want.shape, k
        bad = np.abs(got - want) > 2e-5
        assert bad.mean() <= 2e-3, (k, bad.mean(), np.abs(got - want).max())


def test_one_shard_of_C5_at_its_own_shape():
    """BASELINE.json C5 sharded 8 ways: one rank's head at ITS shape -- 1,250,001 owned rows, d = 256, the node's 8 x 256
    sequences (10.2 GB of partial logits).  Too big for an element-wise oracle: sampled columns / rows against torch in
    float64, the loss against a chunked logsumexp, and the split-K d h_last against a chunked matmul."""
    from bsarec_amd import _lib as L
    lib = L.load()
    dev = torch.device("cuda:0")
    V, W, d, Bg = 10_000_001, 8, 256, 2048
    rows_per = (V + W - 1) // W
    lo, vs = 0, rows_per
    ld = (vs + 3) // 4 * 4
    g = torch.Generator(device="cuda").manual_seed(1)
    E = torch.randn(rows_per, d, device=dev, generator=g) * 0.05
    h = torch.randn(Bg, d, device=dev, generator=g)
    ans = torch.randint(0, V, (Bg,), device=dev, generator=g)
    ans[::2] = torch.randint(0, vs, (Bg // 2,), device=dev, generator=g)        # half of the answers owned by this rank
    st = torch.cuda.current_stream().cuda_stream
    logits = torch.empty(Bg, ld, device=dev)
    stats_all = torch.zeros(W, 3, Bg, device=dev)
    stats_all[1:, 0] = -float("inf")                   # the seven other ranks: empty statistics
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    ev[0].record()
    L.check(lib.bsarec_shard_logits(h.data_ptr(), d, Bg, E.data_ptr(), vs, d, logits.data_ptr(), ld, st), "logits")
    ev[1].record()
    cols = torch.randint(0, vs, (512,), device=dev, generator=g)
    torch.testing.assert_close(logits[:, cols].double(), h.double() @ E[cols].double().T, rtol=1e-5, atol=1e-5)
    L.check(lib.bsarec_shard_ce_stats(logits.data_ptr(), ld, Bg, vs, ans.data_ptr(), lo, V, stats_all[0].data_ptr(), st), "stats")
    lse = torch.cat([torch.logsumexp(logits[i:i + 128, :vs].double(), 1) for i in range(0, Bg, 128)])
    owned = ans < vs
    tgt = torch.where(owned, logits[torch.arange(Bg, device=dev), ans.clamp(max=vs - 1)].double(), torch.zeros_like(lse))
    sample = logits[:, cols].clone()
    loss_rows = torch.zeros(Bg, device=dev)
    loss = torch.zeros(1, device=dev)
    ev[2].record()
    L.check(lib.bsarec_shard_ce_grad(logits.data_ptr(), ld, Bg, vs, ans.data_ptr(), lo, V, stats_all.data_ptr(), W,
                                     loss_rows.data_ptr(), loss.data_ptr(), st), "grad")
    ev[3].record()
    torch.testing.assert_close(loss_rows.double(), lse - tgt, rtol=1e-5, atol=1e-5)
    want = (torch.exp(sample.double() - lse[:, None]) - (cols[None, :] == ans[:, None]).double()) / Bg
    torch.testing.assert_close(logits[:, cols].double(), want, rtol=1e-4, atol=1e-9)
    dE = torch.empty(rows_per, d, device=dev)
    dh = torch.empty(Bg, d, device=dev)
    scratch = torch.zeros(lib.bsarec_shard_head_bwd_scratch_floats(Bg, vs, d), device=dev)
    L.check(lib.bsarec_shard_head_bwd(logits.data_ptr(), ld, Bg, vs, h.data_ptr(), d, E.data_ptr(), d, dE.data_ptr(),
                                      dh.data_ptr(), scratch.data_ptr(), st), "head_bwd")
    ev[4].record()
    torch.cuda.synchronize()
    torch.testing.assert_close(dE[cols].double(), logits[:, cols].double().T @ h.double(), rtol=1e-4, atol=1e-9)
    dh_want = torch.zeros(Bg, d, device=dev, dtype=torch.float64)
    for i in range(0, vs, 131072):
        dh_want += logits[:, i:min(i + 131072, vs)].double() @ E[i:i + 131072].double()
    torch.testing.assert_close(dh.double(), dh_want, rtol=1e-4, atol=1e-8)
    print("C5 shard head, ms: logits %.2f  ce_grad %.2f  head_bwd %.2f" %
          (ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3]), ev[3].elapsed_time(ev[4])))


def _graph_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from bsarec_amd.catalogue import ShardedCatalogue
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            device_id=torch.device("cuda", 0))
    try:
        ns = _ns(hidden_dropout_prob=0.3, attention_probs_dropout_prob=0.2)
        B = ns.batch_size
        res = {}
        for mode in ("eager", "graph"):
            torch.manual_seed(7)                                     # the encoder replica draws its weights from torch's generator
            sc = ShardedCatalogue(ns, B, dist.group.WORLD, "cuda:0")
            losses = []
            for ids, ans in _batches(ns, 4, B):
                step = sc.train_step_graph if mode == "graph" else sc.train_step
                losses.append(float(step(ids.cuda(), ans.cuda()).item()))
            if mode == "graph":
                assert sc.graph_captured, getattr(sc, "_graph_error", "no capture attempted")
            sc.check_exchange()
            res[mode] = (losses, {k: v.detach().cpu().numpy() for k, v in sc.full_state_dict().items()})
            sc.close()
        np.savez(os.path.join(out_dir, "graph.npz"), le=np.asarray(res["eager"][0]), lg=np.asarray(res["graph"][0]),
                 **{"e/" + k: v for k, v in res["eager"][1].items()}, **{"g/" + k: v for k, v in res["graph"][1].items()})
    finally:
        dist.destroy_process_group()


def test_sharded_step_replays_from_one_graph_with_its_collectives_rccl_one_rank(tmp_path):
    """ShardedCatalogue.train_step_graph: the whole catalogue-sharded step -- ~25 kernels and the five collectives between
    them -- captured in ONE hipGraph over RCCL (a 1-rank group: the only RCCL group a one-GPU box can form; the collectives are
    real RCCL calls on the capture stream) and replayed: same losses and parameters as the eager step (same dropout stream)."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_graph_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    z = np.load(tmp_path / "graph.npz")
    np.testing.assert_allclose(z["lg"], z["le"], rtol=1e-5)
    for k in [k[2:] for k in z.files if k.startswith("e/")]:
        a, b = z["g/" + k], z["e/" + k]
        bad = np.abs(a - b) > 2e-5
        assert bad.mean() <= 2e-3, (k, bad.mean(), np.abs(a - b).max())
