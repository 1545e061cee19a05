"""Two data-parallel ranks on ONE GPU (gloo for the control plane, so that no second device is needed): the sharded
step -- same permutation on both ranks, rank slice of every global batch, ONE gradient exchange, Adam on sum / world --
must leave both replicas bit-identical and equal to a single process that trains on the global batch of 2B (dropout
off: the Philox streams are rank-decorrelated by design).  Every form of the exchange (bsarec_amd/dp.py):
  rccl           one all-reduce of the flat arena (gloo stands in for RCCL here),
  rccl_bucketed  dense item-table bucket started by the library's hook right after the logits backward + second bucket
                 (encoder gradients + lookup rows) after the backward,
  p2p            IPC-mapped peer arenas, one barrier kernel, the fused Adam reads both ranks' arenas -- the real data
                 path (hipIpc mappings, flags, step-parity arenas), with both ranks on the one GPU of the box."""
import argparse
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _ns():
    return argparse.Namespace(item_size=301, hidden_size=64, max_seq_length=50, batch_size=64, hidden_dropout_prob=0.0,
                              attention_probs_dropout_prob=0.0, num_hidden_layers=2, num_attention_heads=2,
                              hidden_act="gelu", initializer_range=0.02, c=3, alpha=0.9, seed=42, lr=1e-3,
                              adam_beta1=0.9, adam_beta2=0.999, weight_decay=0.0, no_cuda=False, log_freq=1)


def _table():
    from bsarec_amd import data as D
    seqs = D.synth_ml1m_like(seed=3, n_users=40, n_items=300)
    u, x, a_ = D.train_table(seqs, 50)
    return u[:1024], x[:1024], a_[:1024]


def _worker(rank, world, port, graph, exchange, out_dir):
    import torch.distributed as dist
    from bsarec_amd import BSARecModel, data as D
    from bsarec_amd.trainer import Trainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        u, x, a_ = _table()
        torch.manual_seed(1)
        model = BSARecModel(_ns()).cuda()
        model.set_seed(5, rank)
        dl = D.DeviceBatches(u, x, a_, 64, "cuda", shuffle=True, seed=11, rank=rank, world=world)
        tr = Trainer(model, dl, None, None, _ns(), None, use_graph=graph, process_group=dist.group.WORLD, exchange=exchange)
        assert tr.exchange == exchange, (tr.exchange, exchange)       # no silent fallback in this test
        tr.steps_per_graph = 4                   # 8 steps per epoch and rank: groups of 4 replay as ONE graph launch (p2p)
        if exchange != "p2p":
            tr.dp_graph = "two"                  # gloo's all-reduce cannot be captured: grad graph + eager exchange + Adam graph
        losses = [float(tr.train(e)["rec_loss"]) for e in range(2)]
        if exchange == "p2p":
            assert not tr._px.timed_out()
            assert tr.dp_graph == "one" or not graph
            if graph:
                assert any(k[0] == "indexed_multi" for k in tr._graphs if isinstance(k, tuple)), "multi-step graph not exercised"
        sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), losses=np.asarray(losses), **sd)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["rccl", "rccl_bucketed", "p2p"])
@pytest.mark.parametrize("graph", [False, True])
def test_two_ranks_equal_each_other_and_the_global_batch(graph, exchange, tmp_path):
    import torch.multiprocessing as mp
    from bsarec_amd import BSARecModel, data as D
    from bsarec_amd.trainer import Trainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, graph, exchange, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in r0.files:
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)          # replicas stay bit-identical
    # single process, global batch = 2 x 64 over the same permutation
    u, x, a_ = _table()
    torch.manual_seed(1)
    model = BSARecModel(_ns()).cuda()
    model.set_seed(5)
    dl = D.DeviceBatches(u, x, a_, 128, "cuda", shuffle=True, seed=11)
    tr = Trainer(model, dl, None, None, _ns(), None, use_graph=False)
    losses = [float(tr.train(e)["rec_loss"]) for e in range(2)]
    np.testing.assert_allclose(r0["losses"], losses, atol=2e-4)
    sd = model.state_dict()
    for k in sd:
        got, want = r0[k], sd[k].detach().cpu().numpy()
        bad = np.abs(got - want) > 2e-5
        assert bad.mean() <= 2e-3, (k, bad.mean(), np.abs(got - want).max())


def _prepare_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from bsarec_amd import BSARecModel, data as D
    from bsarec_amd.trainer import Trainer, graph_sizes
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        u, x, a_ = _table()
        u, x, a_ = (np.concatenate([t] * 16) for t in (u, x, a_))          # 128 steps per rank and epoch
        torch.manual_seed(1)
        model = BSARecModel(_ns()).cuda()
        model.set_seed(5, rank)
        dl = D.DeviceBatches(u, x, a_, 64, "cuda", shuffle=True, seed=11, rank=rank, world=world)
        tr = Trainer(model, dl, None, None, _ns(), None, use_graph=True, process_group=dist.group.WORLD, exchange="p2p")
        tr.steps_per_graph = 4
        perm = dl.local_permutation()
        cur = torch.zeros(1, dtype=torch.int64, device="cuda")
        ran = tr.prepare_indexed(dl, perm, cur, None)
        n0 = tr.graphs_built()
        # single-step graphs for both parities + one graph per (group size, parity)
        assert n0 == 2 + 2 * (len(graph_sizes(4)) - 1), (n0, sorted(map(str, tr._graphs)))
        for n in (7, 4, 1, 2, 9):                       # every mix of group sizes, starting at either parity
            tr.indexed_steps(dl, perm, cur, None, n)
            ran += n
        torch.cuda.synchronize()
        assert tr.graphs_built() == n0, "indexed_steps captured a graph after prepare_indexed"
        assert int(cur.item()) == ran * 64
        tr.check_exchange()
        np.savez(os.path.join(out_dir, f"prep{rank}.npz"), arena=model._arena.cpu().numpy(), ran=ran)
    finally:
        dist.destroy_process_group()


def test_prepare_indexed_builds_every_graph_up_front_p2p(tmp_path):
    """Trainer.prepare_indexed under the peer-to-peer exchange (two step parities): afterwards indexed_steps() of any
    length captures nothing, the device cursor advanced by exactly the steps run, and the replicas are bit-identical."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_prepare_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "prep0.npz"), np.load(tmp_path / "prep1.npz")
    assert int(r0["ran"]) == int(r1["ran"])
    np.testing.assert_array_equal(r0["arena"], r1["arena"])


# ---- data parallel for the sibling models (SASRec / FMLPRec: fused step split around the exchange; DuoRec: torch.optim loop
#      with one coalesced all-reduce of p.grad) ------------------------------------------------------------------------------

def _sib_batches(kind, B, steps, V=301, L=50):
    rng = np.random.default_rng(17)
    out = []
    for _ in range(steps):
        ids = np.zeros((B, L), dtype=np.int64)
        for b in range(B):
            n = int(rng.integers(1, L + 1))
            ids[b, L - n:] = rng.integers(1, V, size=n)
        ans = rng.integers(1, V, size=B).astype(np.int64)
        neg = rng.integers(1, V, size=B).astype(np.int64)
        same = np.zeros((B, L), dtype=np.int64)
        for b in range(B):
            n = int(rng.integers(1, L + 1))
            same[b, L - n:] = rng.integers(1, V, size=n)
        out.append((np.arange(B, dtype=np.int64), ids, ans, neg, same))
    return out


def _sib_ns(kind):
    a = _ns()
    a.contrast, a.tau, a.lmd, a.lmd_sem, a.ssl, a.sim = "us_x", 1.0, 0.1, 0.1, "us_x", "dot"
    return a


def _sib_worker(rank, world, port, kind, out_dir):
    import torch.distributed as dist
    from bsarec_amd import MODEL_DICT
    from bsarec_amd.trainer import Trainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        torch.manual_seed(1)
        model = MODEL_DICT[kind](args=_sib_ns(kind)).cuda()
        model.set_seed(5, rank)
        Bl = 16
        batches = [tuple(torch.from_numpy(t[rank * Bl:(rank + 1) * Bl]).cuda() for t in bt) for bt in _sib_batches(kind, 2 * Bl, 3)]
        tr = Trainer(model, batches, None, None, _sib_ns(kind), None, use_graph=False, process_group=dist.group.WORLD, exchange="rccl")
        losses = [float(tr.train(e)["rec_loss"]) for e in range(2)]
        sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        np.savez(os.path.join(out_dir, f"sib{rank}.npz"), losses=np.asarray(losses), **sd)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["sasrec", "fmlprec", "duorec"])
def test_two_ranks_sibling_models(kind, tmp_path):
    """Trainer(process_group=...) for the sibling models: replicas stay bit-identical; for the pos / neg heads (SASRec,
    FMLPRec: the loss is a mean over sequences) two ranks equal ONE process on the global batch; DuoRec's in-batch
    contrastive terms differ between a 2 x 16 and a 32 batch by construction, so there only the replicas are compared."""
    import torch.multiprocessing as mp
    from bsarec_amd import MODEL_DICT
    from bsarec_amd.trainer import Trainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_sib_worker, args=(2, port, kind, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "sib0.npz"), np.load(tmp_path / "sib1.npz")
    for k in r0.files:
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)
    assert np.isfinite(r0["losses"]).all()
    if kind == "duorec":
        return
    torch.manual_seed(1)
    model = MODEL_DICT[kind](args=_sib_ns(kind)).cuda()
    model.set_seed(5)
    batches = [tuple(torch.from_numpy(t).cuda() for t in bt) for bt in _sib_batches(kind, 32, 3)]
    tr = Trainer(model, batches, None, None, _sib_ns(kind), None, use_graph=False)
    losses = [float(tr.train(e)["rec_loss"]) for e in range(2)]
    np.testing.assert_allclose(r0["losses"], losses, atol=2e-4)
    sd = model.state_dict()
    for k in sd:
        got, want = r0[k], sd[k].detach().cpu().numpy()
        bad = np.abs(got - want) > 2e-5
        assert bad.mean() <= 2e-3, (k, bad.mean(), np.abs(got - want).max())
