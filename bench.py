#!/usr/bin/env python3
"""Training-throughput bench of the HIP BSARec path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N rank processes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      # or under a launcher

One "step" = Trainer.iteration's per-batch body (src/trainers.py:100-107) on one batch of the
ML-1M-shaped synthetic workload (C1 of SURVEY 8d: V=3417, L=50, d=64, 2 layers, 2 heads, c=3,
alpha=0.9, dropout 0.5, Adam lr 1e-3, B=256 sequences per GPU): device-side batch gather,
forward, full-catalogue CE, backward, (gradient all-reduce for N > 1), fused Adam.  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, spec
BF16_MFMA_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: bf16 MFMA, dense (not the 2:1-sparse headline)
PMC_PROFILE = os.path.join("profiles", "r03_pmc_C1.csv")     # committed rocprofv3 --pmc passes of this round's library


def model_args(a):
    ns = argparse.Namespace(
        item_size=a.item_size, hidden_size=a.hidden, max_seq_length=a.seq_len, batch_size=a.batch,
        hidden_dropout_prob=0.5, attention_probs_dropout_prob=0.5, num_hidden_layers=a.layers,
        num_attention_heads=a.heads, hidden_act="gelu", initializer_range=0.02, c=3, alpha=0.9, seed=42,
        lr=1e-3, adam_beta1=0.9, adam_beta2=0.999, weight_decay=0.0, no_cuda=False, log_freq=1)
    if getattr(a, "dtype", "f32") == "bf16":
        ns.storage = "bf16"
    return ns


def train_flops_per_seq(a, cb=2):
    d, L, N, V = a.hidden, a.seq_len, a.layers, a.item_size
    return 3.0 * (N * L * (24 * d * d + 4 * L * d + 8 * cb * d) + 2 * d * V)


def cpu_baseline(a, budget_s=12.0):
    """The CPU oracle (numpy restatement, validated against the imported reference) timed on this
    box's host cores on a bounded sample of the same workload: whole training steps at C1 shape."""
    import numpy as np
    from oracle import bsarec_oracle as O
    cfg = O.Config(item_size=a.item_size, hidden_size=a.hidden, max_seq_length=a.seq_len, num_hidden_layers=a.layers,
                   num_attention_heads=a.heads, c=3, alpha=0.9)
    P = O.init_params(cfg, 0)
    rng = np.random.default_rng(0)
    ids = rng.integers(1, a.item_size, size=(a.batch, a.seq_len))
    for b in range(a.batch):
        ids[b, :rng.integers(0, a.seq_len)] = 0
    ans = rng.integers(1, a.item_size, size=a.batch)
    st = O.AdamState()
    _, _, G, _ = O.loss_and_grads(P, cfg, ids, ans, O.DropoutSpec(True, 1, 1))      # warm-up
    O.adam_step(P, G, st)
    n, t0 = 0, time.time()
    while time.time() - t0 < budget_s and n < 64:
        _, _, G, _ = O.loss_and_grads(P, cfg, ids, ans, O.DropoutSpec(True, 1, n + 2))
        O.adam_step(P, G, st)
        n += 1
    dt = time.time() - t0
    out = {"value": round(n * a.batch / dt, 1), "unit": "sequences/s", "cores": os.cpu_count(), "kind": "port",
           "sample": f"{n} full training steps (fwd+bwd+Adam, dropout on) of B={a.batch} at the C1 shape, "
                     f"numpy oracle with BLAS threads on all host cores, {dt:.1f} s"}
    # how the port compares with the reference's own CPU path (imported PyTorch reference, same shape, same host):
    # measured in the build container by tools/cpu_ref_ratio.py -- the reference cannot travel to the GPU box
    path = os.path.join(ROOT, "profiles", "cpu_ref_ratio.json")
    if os.path.exists(path):
        r = json.load(open(path))
        out["ref_ratio"] = r["port_over_reference_throughput"]
        out["ref_ratio_source"] = ("profiles/cpu_ref_ratio.json (tools/cpu_ref_ratio.py in the build container, "
                                   f"{r['cores']} cores): port {r['port_seq_per_s']} vs imported reference {r['reference_seq_per_s']} seq/s; "
                                   "reference-equivalent CPU rate on this host ~= value / ref_ratio")
    return out


def fused_shape(a):
    return a.hidden == 64 and a.seq_len <= 64


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(a):
    """`python bench.py --gpus N` run bare: this process starts the N rank processes (one per GPU) and relays rank 0's
    JSON line.  It has touched no GPU (no HIP call, no torch.cuda call) -- the children are fresh interpreters."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, BSAREC_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = None if r == 0 else sys.stderr            # only rank 0 owns stdout (it prints the one JSON line)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    rc = 0
    deadline = time.time() + float(os.environ.get("BSAREC_BENCH_TIMEOUT", "1500"))
    for p in procs:
        try:
            rc = max(rc, abs(p.wait(timeout=max(1.0, deadline - time.time()))))
        except subprocess.TimeoutExpired:
            rc = max(rc, 124)
    if rc:
        for p in procs:                                  # exact PIDs we started, nothing by pattern
            if p.poll() is None:
                p.kill()
    return rc


def p2p_probe_child():
    """A throw-away process per rank (own process group on MASTER_PORT + 17) that exercises the peer-to-peer exchange on
    THIS node's GPUs before the measuring processes commit to it: IPC export / import of the arenas and flags, the
    barrier kernel, torch reads through every mapping (PeerExchange.create -> self_test) and one fused Adam step that
    sums every rank's arena with the library's system-scope loads.  The peer-to-peer path reads other GPUs' memory from
    kernels: were a mapping wrong the process would die of a memory fault -- here, not in the benchmark.  Exit 0 = usable."""
    import ctypes as C
    import torch
    if os.environ.get("BSAREC_P2P_PROBE_FAIL") == "1":   # rehearsal of the fallback
        return 5
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("BSAREC_DIST_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29533")) + 17)
    if backend == "nccl":
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        torch.distributed.init_process_group(backend, rank=rank, world_size=world)
    from bsarec_amd import _lib as Lb
    from bsarec_amd.dp import PeerExchange
    n = 322368                                          # the C1 arena
    px = PeerExchange.create(n, torch.distributed.group.WORLD, dev)
    if px is None:
        return 3
    px.arenas[0].fill_(float(rank + 1))
    st = torch.cuda.current_stream().cuda_stream
    px.barrier(st)
    w, m, v = (torch.zeros(n, device=dev) for _ in range(3))
    state = torch.zeros(8, dtype=torch.int64, device=dev)
    ad = Lb.Adam(w.data_ptr(), px.arenas[0].data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0 / world, None, 0)
    ad.n_grad_srcs = world
    for i, q in enumerate(px.grad_srcs(0)):
        ad.grad_srcs[i] = q
    Lb.check(Lb.load().bsarec_adam_step(C.byref(ad), state.data_ptr(), st), "bsarec_adam_step")
    px.barrier(st)
    torch.cuda.synchronize()
    g = (world + 1) / 2.0                               # mean over ranks of (rank + 1)
    ok = bool(torch.allclose(m, torch.full_like(m, 0.1 * g), rtol=1e-5)) and bool(torch.allclose(w, torch.full_like(w, -1e-3), rtol=1e-3))
    ok = ok and not px.timed_out()
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
    px.close()
    torch.distributed.destroy_process_group()
    return 0 if flag.item() == 1.0 else 4


def p2p_probe(a, world, backend):
    """Run p2p_probe_child in a child of this (not yet GPU-touching) rank; True if it exited 0 in time."""
    env = dict(os.environ, BSAREC_P2P_PROBE_CHILD="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # under torch.distributed.run the agent hosts the rendezvous store of the MAIN group; the probe group (MASTER_PORT + 17)
    # must host its own (its rank 0), so it must not be told to look for an agent store
    for k in [k for k in env if k.startswith("TORCHELASTIC_")]:
        env.pop(k)
    try:
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=sys.stderr)
        try:
            return p.wait(timeout=float(os.environ.get("BSAREC_P2P_PROBE_TIMEOUT", "240"))) == 0
        except subprocess.TimeoutExpired:
            p.kill()                                    # the exact PID started above
            return False
    except OSError:
        return False


class Feed:
    """Steps straight off the device-resident sample table: per step ONE C call (gather + fwd + CE + bwd + Adam), replayed
    from hipGraphs (Trainer.indexed_steps); keeps the host-side position so that a run never crosses the end of an epoch's
    permutation inside a graph."""

    def __init__(self, tr, bt, dev):
        import torch
        self.tr, self.bt, self.B = tr, bt, bt.batch_size
        self.cursor = torch.zeros(1, dtype=torch.int64, device=dev)
        self.perm_buf = torch.zeros(bt.answers.shape[0], dtype=torch.int64, device=dev)
        self.n_local, self.pos = 0, 0

    def new_epoch(self):
        perm = self.bt.local_permutation()
        self.bt.epoch += 1
        self.n_local = (perm.shape[0] // self.B) * self.B       # full batches only inside the timed region
        self.perm_buf[:perm.shape[0]].copy_(perm)
        self.cursor.zero_()
        self.pos = 0

    def prepare(self):
        """Build (capture + instantiate + replay once) every graph run() can launch -- BEFORE any clock starts."""
        need = 8 * self.tr.steps_per_graph + 16
        if self.pos + need * self.B > self.n_local:
            self.new_epoch()
        assert need * self.B <= self.n_local, "sample table too small to build the step graphs"
        ran = self.tr.prepare_indexed(self.bt, self.perm_buf, self.cursor, None)
        assert ran <= need
        self.pos += ran * self.B
        return ran

    def run(self, nsteps):
        """Exactly ``nsteps`` optimisation steps (groups of them replay as one graph launch: trainer.graph_schedule)."""
        loss = None
        while nsteps > 0:
            if self.pos + self.B > self.n_local:
                self.new_epoch()
            k = min(nsteps, (self.n_local - self.pos) // self.B)
            loss = self.tr.indexed_steps(self.bt, self.perm_buf, self.cursor, None, k)
            self.pos += k * self.B
            nsteps -= k
        return loss


def timed_steps(fd, steps, warmup, barrier):
    """``warmup`` untimed steps (after every graph of the run has been built and replayed once, untimed too), then EXACTLY
    ``steps`` steps between barrier + synchronize on both sides.  Asserts that the timed region built no graph (round-2
    VERDICT: under ``--steps 20 --warmup 5`` a 16-step graph used to be captured on the clock)."""
    fd.prepare()
    loss = fd.run(max(warmup, 1))
    barrier()
    g0 = fd.tr.graphs_built()
    t0 = time.perf_counter()
    loss = fd.run(steps)
    barrier()
    dt = time.perf_counter() - t0
    assert fd.tr.graphs_built() == g0, "a hipGraph was captured inside the timed region"
    return dt, loss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="sequences per GPU per step (reference default 256)")
    ap.add_argument("--item_size", type=int, default=3417)
    ap.add_argument("--seq_len", type=int, default=50)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--heads", type=int, default=2)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32: the reference's arithmetic (headline).  bf16: config C2's storage -- bf16 saved activations + "
                         "bf16 weight shadow + bf16 MFMA, fp32 accumulate / master weights / Adam; a separate, labelled line")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra C3-shape / bf16 measurements of the N = 1 line")
    ap.add_argument("--dp", action="store_true", help="take the data-parallel step (RCCL all-reduce) even with one rank")
    ap.add_argument("--exchange", choices=["auto", "rccl", "rccl_bucketed", "p2p"], default="auto",
                    help="gradient exchange of the data-parallel step: RCCL all-reduce(s), or the one-shot peer-to-peer read-reduce "
                         "fused into Adam (IPC-mapped peer arenas over xGMI)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world == 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a))                        # before anything here touches a GPU
    if world != a.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {a.gpus}")
    if os.environ.get("BSAREC_P2P_PROBE_CHILD") == "1":
        sys.exit(p2p_probe_child())
    # N > 1 with the default exchange: try the peer-to-peer path in a throw-away child process first (before THIS process
    # touches a GPU); a crash or a hang there costs the probe, not the benchmark, which then runs on RCCL
    probe_ok = None
    if world > 1 and a.exchange == "auto" and (os.environ.get("BSAREC_DIST_BACKEND", "nccl") == "nccl" or
                                              os.environ.get("BSAREC_P2P_PROBE") == "1"):
        probe_ok = p2p_probe(a, world, os.environ.get("BSAREC_DIST_BACKEND", "nccl"))

    import numpy as np
    import torch

    # stdout carries exactly ONE line (the JSON): everything else a library prints there -- RCCL's version banner at
    # communicator creation, for one -- is sent to stderr by pointing fd 1 at fd 2 for the duration of the run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # BSAREC_DIST_BACKEND=gloo: rehearsal of the N > 1 path with all ranks on ONE GPU (the build box has one); the
    # exchange then goes through the host, so the step uses grad graph + eager all-reduce + Adam graph
    backend = os.environ.get("BSAREC_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local = 0                 # (the Trainer itself falls back to grad graph + eager exchange + Adam graph where the
                                  #  exchange cannot be captured, i.e. for gloo collectives)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1 or a.dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)
        pg = torch.distributed.group.WORLD

    from bsarec_amd import BSARecModel, _lib as Lb
    from bsarec_amd import data as D
    from bsarec_amd.trainer import Trainer

    margs = model_args(a)
    # ML-1M-shaped synthetic interactions -> device-resident sample table (identical on every rank)
    seqs = D.synth_ml1m_like(seed=42, n_items=a.item_size - 1)
    users, inputs, answers = D.train_table(seqs, a.seq_len)

    def build(exchange):
        torch.manual_seed(42)                           # identical replicas on every rank (the Trainer also broadcasts rank 0's)
        model = BSARecModel(margs).to(dev)
        model.set_seed(42, rank)
        batches = D.DeviceBatches(users, inputs, answers, a.batch, dev, shuffle=True, seed=42, rank=rank, world=world)
        trainer = Trainer(model, batches, None, None, margs, None, use_graph=not a.no_graph, process_group=pg, exchange=exchange)
        return model, batches, trainer

    exchange = a.exchange
    if probe_ok is not None:                            # every rank's probe must have passed
        f = torch.tensor([1.0 if probe_ok else 0.0], device=dev)
        torch.distributed.all_reduce(f, op=torch.distributed.ReduceOp.MIN, group=pg)
        probe_ok = bool(f.item() == 1.0)
        if not probe_ok:
            exchange = "rccl"
            if rank == 0:
                print("bench: the peer-to-peer probe failed on some rank -- gradient exchange through RCCL", file=sys.stderr)
    model, batches, trainer = build(exchange)
    use_graph = trainer.use_graph

    # steps come straight off the device-resident table: per step ONE C call (gather + fwd + CE + bwd + Adam),
    # replayed from a hipGraph at N = 1; gather + fwd/bwd + exchange + Adam for N > 1
    B = a.batch

    feed = Feed(trainer, batches, dev)

    def stream_batches():
        while True:
            for bt in batches:
                if bt[1].shape[0] == a.batch:
                    yield bt
    stream = stream_batches()

    def barrier():
        if pg is not None:
            if backend == "nccl":
                torch.distributed.barrier(device_ids=[local])
            else:
                torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(fd, steps, warmup):
        return timed_steps(fd, steps, warmup, barrier)

    def replicas_state(mdl, tr):
        """Only gradients are exchanged: after any number of steps the parameter checksum must be the same on every rank
        (and no peer-to-peer wait may have given up).  -> (agree, lowest checksum, highest checksum, a wait timed out)"""
        cs = mdl._arena.double().sum().view(1)
        lo, hi = cs.clone(), cs.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN, group=pg)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX, group=pg)
        bad = torch.tensor([1.0 if (tr.exchange == "p2p" and tr._px.timed_out()) else 0.0], device=dev)
        torch.distributed.all_reduce(bad, op=torch.distributed.ReduceOp.MAX, group=pg)
        return bool(lo.item() == hi.item()) and bad.item() == 0.0, float(lo.item()), float(hi.item()), bool(bad.item() != 0.0)

    def replicas_agree(mdl, tr):
        return replicas_state(mdl, tr)[0]

    model.train()
    dt, loss = timed(feed, a.steps, a.warmup)
    p2p_rerun = None
    if world > 1 and a.exchange == "auto" and trainer.exchange == "p2p":
        agree, cs_lo, cs_hi, timed_out = replicas_state(model, trainer)
        forced = os.environ.get("BSAREC_P2P_FORCE_RERUN") == "1"        # rehearsal of this path on a healthy run
        if not agree or forced:
            # never seen on the hardware this was written on (one GPU); on a node where the peer-to-peer exchange misbehaves the
            # measurement is repeated through RCCL rather than lost -- and the line says WHY it was repeated
            if not agree:
                why = ("a cross-GPU barrier wait timed out" if timed_out else "the replicas' parameter checksums differ") + \
                      f" (lowest {cs_lo!r}, highest {cs_hi!r} over the ranks)"
                p2p_rerun = {"reason": "divergence", "detail": why, "forced": False}
            else:
                p2p_rerun = {"reason": "forced", "detail": "BSAREC_P2P_FORCE_RERUN=1 (rehearsal; the replicas agreed: "
                                                           f"checksum {cs_lo!r} on every rank)", "forced": True}
            p2p_rerun["action"] = "measured again through RCCL"
            if rank == 0:
                print(f"bench: peer-to-peer run repeated through RCCL -- {p2p_rerun['reason']}: {p2p_rerun['detail']}", file=sys.stderr)
            del feed, trainer, model, batches
            model, batches, trainer = build("rccl")
            use_graph = trainer.use_graph
            feed = Feed(trainer, batches, dev)
            stream = stream_batches()
            model.train()
            dt, loss = timed(feed, a.steps, a.warmup)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())
    assert np.isfinite(final_loss), "training diverged"

    pruned = fused_shape(a) and a.layers >= 2 and not model._plan(B).options["no_prune_top"]
    out = {
        "metric": "train sequences/sec, ML-1M L=50 d=64 2-layer", "value": round(a.batch * world * a.steps / dt, 1),
        "unit": "sequences/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"C1: ML-1M-shaped synthetic (6040 users, V={a.item_size}), L={a.seq_len} d={a.hidden} "
                               f"{a.layers} BSARec layers, {a.heads} heads, c=3 alpha=0.9 dropout=0.5, Adam lr=1e-3; "
                               "fwd + full-catalogue CE + bwd + Adam per step",
                   "batch_per_gpu": a.batch, "global_batch": a.batch * world, "seq_len": a.seq_len,
                   "parallelism": f"dp{world}", "launch": ((f"hipGraph replay, {trainer.steps_per_graph} steps per graph launch" if trainer.steps_per_graph > 1 else "hipGraph replay") if use_graph else "eager") +
                             (f"; data-parallel exchange: {trainer.exchange_desc()}" if pg is not None else ""),
                   "final_loss": round(final_loss, 4)},
    }
    if a.dtype == "bf16":
        out["config"]["precision"] = ("bf16 storage of the saved activations and of a bf16 shadow of the Linear weights, bf16 MFMA "
                                      "with fp32 accumulation; fp32 master weights, LayerNorm, softmax, loss and Adam (config C2's "
                                      "storage; NOT the headline: the reference computes in fp32)")
    out["config"]["top_block"] = ("loss path evaluates the top BSARecBlock on position L-1 only (it still attends to every "
                                  "position); exact, same loss and gradients (SURVEY C.6); algorithmic_* figures below use the "
                                  "un-pruned counts, executed_* the work actually launched") if pruned else "full"
    if pg is not None:
        # evidence that the collective really spans N ranks, and what one exchange of the gradient arena costs on its own
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones, group=pg)
        out["ranks_seen"] = int(round(float(ones.item())))
        g = model._garena
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            torch.distributed.all_reduce(g, group=pg)
        barrier()
        ev0.record()
        for _ in range(50):
            torch.distributed.all_reduce(g, group=pg)
        ev1.record()
        torch.cuda.synchronize()
        out["allreduce_us"] = round(ev0.elapsed_time(ev1) * 1e3 / 50, 2)
        out["allreduce_bytes"] = int(g.numel() * 4)
        out["exchange"] = trainer.exchange_report()
        if probe_ok is not None:
            out["exchange"]["p2p_probe"] = "passed (separate process per rank, before the run)" if probe_ok else \
                "failed: this run exchanges gradients through RCCL"
        if p2p_rerun:
            out["exchange"]["p2p_rerun"] = p2p_rerun
        # the replicas must still be bit-identical after the timed steps (only gradients are exchanged)
        out["replicas_identical"] = replicas_agree(model, trainer)
        assert out["replicas_identical"], "data-parallel replicas diverged"

    flops_seq = train_flops_per_seq(a)
    peak = FP32_MFMA_PEAK_TFLOPS if a.dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
    out["algorithmic_step_mfma_frac"] = round(flops_seq * a.batch * world * a.steps / dt / (peak * 1e12 * world), 5)

    solo = rank == 0 and world == 1
    if solo and pruned and not a.no_roofline:
        # the same step with the FULL top-block kernels (nothing uses the one-row structure), measured in this run:
        # every algorithmic FLOP is executed, so this is the executed-work fraction of the MFMA peak
        old = Lb.set_default_options(no_prune_top=1)
        try:
            torch.manual_seed(42)
            model2 = BSARecModel(margs).to(dev)
            model2.set_seed(42, rank)
            model2.train()
            batches2 = D.DeviceBatches(users, inputs, answers, a.batch, dev, shuffle=True, seed=42, rank=rank, world=world)
            trainer2 = Trainer(model2, batches2, None, None, margs, None, use_graph=not a.no_graph, process_group=pg)
            feed2 = Feed(trainer2, batches2, dev)
            nst = min(a.steps, 200)
            dt2, l2 = timed(feed2, nst, a.warmup)
            assert np.isfinite(float(l2.item()))
            out["full_top_block"] = {"value": round(a.batch * nst / dt2, 1), "ms_per_step": round(1e3 * dt2 / nst, 4), "steps": nst}
            out["executed_step_mfma_frac"] = round(flops_seq * a.batch * nst / dt2 / (peak * 1e12), 5)
            del trainer2, model2, batches2, feed2
        finally:
            Lb.set_default_options(**old)
    elif solo and not pruned:
        out["executed_step_mfma_frac"] = out["algorithmic_step_mfma_frac"]

    if solo and not a.no_roofline:
        # Per-kernel roofline: hipEvent pairs on the launch stream around every launch of one kernel class
        # inside real (eager) training steps; the class with the largest time per step is the dominant kernel.
        import ctypes as C
        lib = Lb.load()
        d, L, N, cb, h = a.hidden, a.seq_len, a.layers, 2, a.heads
        T = B * L
        fused = fused_shape(a)
        if fused:
            dw_exec = 24.0 * T * d * d                                   # the bottom block's six products
            dw_alg = dw_exec
            if pruned:                                                  # + the one-row top block's: 4 products over B rows, 2 over B*h
                dw_exec += 20.0 * B * d * d + 4.0 * B * h * d * d
                dw_alg = 24.0 * T * d * d * N / max(N - 1, 1)
            fwd_blk = B * L * (24 * d * d + 4 * L * d + 8 * cb * d)       # one full BSARecBlock forward / backward (input-gradient chain)
            bwd_blk = B * L * (24 * d * d + 8 * L * d + 16 * cb * d)
            in_block = pruned and N >= 2 and not model._plan(B).options.get("separate_top", 0)
            if in_block:
                # the one-row top block rides in these launches (tail of the forward, head of the backward): executed = the block
                # below + the top block's K / V projections of all rows and its one-row vector products; algorithmic = two blocks
                fwd_exec = fwd_blk + B * L * 4 * d * d + B * (20 * d * d + 4 * L * d)
                bwd_exec = bwd_blk + B * (26 * d * d + 8 * L * d * h)
                cands = [(Lb.K_FUSED_BWD, "fused_layer_bwd_kernel<head = top block> (input-gradient chain of a BSARecBlock per sequence; "
                                          "the one-row top block's backward runs first inside the same launch)", bwd_exec, 2 * bwd_blk),
                         (Lb.K_FUSED_FWD, "fused_layer_fwd_kernel<tail = top block> (BSARecBlock forward per sequence; the one-row top "
                                          "block's forward runs as the tail of the same launch)", fwd_exec, 2 * fwd_blk)]
            else:
                cands = [(Lb.K_FUSED_BWD, "fused_layer_bwd_kernel (whole BSARecBlock input-gradient chain per sequence)", bwd_blk, None),
                         (Lb.K_FUSED_FWD, "fused_layer_fwd_kernel (whole BSARecBlock forward per sequence)", fwd_blk, None)]
            cands.append((Lb.K_DW1, "dw_direct_kernel (weight + bias gradients, direct split-K; the one-row top block's products ride in "
                                    "the next block's launch)", dw_exec, dw_alg))
        else:
            cands = [(Lb.K_FFN1, "gemm_kernel<NT, EpiLinear<bias>> (FFN dense_1)", 2.0 * T * d * 4 * d, None),
                     (Lb.K_DW1, "gemm_grouped_tn_kernel (6 weight + bias gradients of a block, split-K)", 24.0 * T * d * d, None)]
        rows = []
        ovh = C.c_double()
        lib.bsarec_profile_event_overhead(C.c_void_p(torch.cuda.current_stream().cuda_stream), 200, C.byref(ovh))
        ovh_s = ovh.value * 1e-3          # an empty event bracket: what the two marker packets themselves cost
        plan = model._plan(B)
        for kclass, name, fl, fl_alg in cands:
            lib.bsarec_profile_select(plan.handle, kclass)
            nprof = 10
            for _ in range(nprof):
                _, ids, ans, _, _ = next(stream)
                trainer._step_eager(ids, ans)
            torch.cuda.synchronize()
            ms, n = C.c_double(), C.c_int()
            lib.bsarec_profile_read(plan.handle, C.byref(ms), C.byref(n))
            if n.value == 0:
                continue
            avg_s = ms.value * 1e-3 / n.value - ovh_s
            row = {"kernel": name, "launches_per_step": n.value / nprof, "avg_us": round(avg_s * 1e6, 3),
                   "us_per_step": round(avg_s * 1e6 * n.value / nprof, 2), "flops_per_launch": float(fl),
                   "achieved": round(fl / avg_s / 1e12, 3), "frac": round(fl / avg_s / 1e12 / peak, 5)}
            if fl_alg is not None and fl_alg != fl:
                row["algorithmic_flops_per_launch"] = float(fl_alg)
                row["algorithmic_achieved"] = round(fl_alg / avg_s / 1e12, 3)
            rows.append(row)
        lib.bsarec_profile_select(plan.handle, Lb.K_NONE)
        rows.sort(key=lambda r: -r["us_per_step"])
        top = rows[0]

        def pmc_traffic(kernel_prefix):
            # HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/): WRITE_SIZE + 2 * FETCH_SIZE KiB
            # (gfx950 tallies wide streaming reads at half their bytes, MI355X guide, HBM section).  An OFFLINE pass:
            # only reported when the profile's recorded library hash is this build's
            path = os.path.join(ROOT, PMC_PROFILE)
            if not os.path.exists(path) or not fused or a.batch != 256 or a.dtype != "f32":
                return None, None
            vals, sha = {}, None
            for line in open(path):
                if line.startswith("# library_sha16="):
                    sha = line.strip().split("=", 1)[1]
                if line.startswith('"' + kernel_prefix):
                    _, counter, avg, _ = line.rsplit(",", 3)
                    vals[counter] = float(avg)
            if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
                src = f"{PMC_PROFILE} (offline rocprofv3 --pmc passes, tools/profile_round.sh; library_sha16={sha}"
                src += ", this build)" if sha == Lb.source_sha16() else f", this build is {Lb.source_sha16()}: kernels changed since)"
                return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, src
            return None, None
        traffic, tsrc = pmc_traffic(top["kernel"].split(" ")[0].split("<")[0])

        def fp32_alu_share(kernel_prefix, avg_us):
            """On the fp32 path an MFMA occupies the SIMD's vector ALU: while v_mfma_f32_{16x16x4,32x32x2}_f32 executes, no
            vector instruction of either resident wave issues (tools/micro/mfma_valu_mix.hip -> profiles/r03_micro_mfma_valu_mix.txt;
            a bf16 MFMA does not do that).  So the fp32 kernels' ALU time is MFMA cycles + VALU cycles, ADDITIVE, and the MFMA
            peak alone is not reachable by a kernel that also has vector work.  From the committed PMC passes: cycles per
            SIMD spent in MFMAs (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) + in other vector instructions
            ((SQ_INSTS_VALU - SQ_INSTS_MFMA) / 1024 x 2.2 cycles, the measured two-waves-per-SIMD issue rate), against the
            launch's duration at the 2.4 GHz maximum clock (a lower bound of the busy fraction: the chip clocks lower under load)."""
            path = os.path.join(ROOT, PMC_PROFILE)
            if not os.path.exists(path) or not fused or a.batch != 256 or a.dtype != "f32":
                return None
            vals = {}
            for line in open(path):
                if line.startswith('"' + kernel_prefix):
                    _, counter, avg, _ = line.rsplit(",", 3)
                    vals[counter] = float(avg)
            need = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA")
            if not all(k in vals for k in need):
                return None
            mfma_cyc = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0
            valu_cyc = (vals["SQ_INSTS_VALU"] - vals["SQ_INSTS_MFMA"]) / 1024.0 * 2.2
            avail = avg_us * 1e-6 * 2.4e9
            return {"mfma_cycles_per_simd": round(mfma_cyc), "valu_cycles_per_simd": round(valu_cyc),
                    "launch_cycles_at_2.4GHz": round(avail), "busy_frac": round((mfma_cyc + valu_cyc) / avail, 4),
                    "mfma_only_frac": round(mfma_cyc / avail, 4),
                    "note": "fp32 MFMA and VALU share the SIMD's ALU (micro-benchmark in tools/micro/): busy = MFMA + VALU cycles"}
        alu = fp32_alu_share(top["kernel"].split(" ")[0].split("<")[0], top["avg_us"])
        out["roofline"] = {"bound": "mfma", "kernel": top["kernel"], "achieved": top["achieved"],
                           "peak": peak, "unit": "TFLOP/s",
                           "frac": round(top["achieved"] / peak, 5),
                           "flops_counted": "executed (the FLOPs this launch really issues); algorithmic_* = the un-pruned count of "
                                            "the blocks the launch stands for",
                           "algorithmic_achieved": top.get("algorithmic_achieved"),
                           "algorithmic_frac": round(top["algorithmic_achieved"] / peak, 5) if top.get("algorithmic_achieved") else None,
                           "traffic": traffic, "traffic_source": tsrc, "fp32_alu": alu,
                           "avg_us": top["avg_us"], "event_overhead_us": round(ovh_s * 1e6, 3), "launches_per_step": top["launches_per_step"],
                           "flops_per_launch": top["flops_per_launch"], "other_kernels": rows[1:]}

    if solo and not a.no_secondary and a.dtype == "f32" and fused_shape(a):
        # non-headline workloads measured in the same run (each its own model / plan; a few seconds)
        sec = {}
        try:
            sec["C3"] = secondary_c3(dev, seqs, D, BSARecModel, Trainer)
        except Exception as e:                                      # never lose the headline line to a secondary measurement
            sec["C3"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            sec["C3_bf16"] = secondary_c3(dev, seqs, D, BSARecModel, Trainer, dtype="bf16")
        except Exception as e:  # noqa: BLE001
            sec["C3_bf16"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            sec["C2_storage_bf16"] = secondary_bf16(a, dev, users, inputs, answers, D, BSARecModel, Trainer)
        except Exception as e:
            sec["C2_storage_bf16"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            sec["C2_beauty_bf16"] = secondary_beauty_bf16(dev, D, BSARecModel, Trainer)
        except Exception as e:
            sec["C2_beauty_bf16"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            sec["C4_per_rank_shape"] = secondary_c4(dev, D, BSARecModel, Trainer)
        except Exception as e:
            sec["C4_per_rank_shape"] = {"error": f"{type(e).__name__}: {e}"}
        out["secondary"] = sec
    if solo and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a)
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)
    if pg is not None:
        torch.distributed.destroy_process_group()


def secondary_bf16(a, dev, users, inputs, answers, D, BSARecModel, Trainer, steps=100, warmup=10):
    """The headline workload (C1) through config C2's storage: bf16 saved activations + bf16 weight shadow + bf16 MFMA,
    fp32 accumulation, master weights and Adam.  NOT the headline (the reference computes in fp32): a labelled extra."""
    import copy
    import numpy as np
    import torch
    a2 = copy.copy(a)
    a2.dtype = "bf16"
    m2 = model_args(a2)
    torch.manual_seed(42)
    model = BSARecModel(m2).to(dev)
    model.set_seed(42, 0)
    model.train()
    bt = D.DeviceBatches(users, inputs, answers, a.batch, dev, shuffle=True, seed=42)
    tr = Trainer(model, bt, None, None, m2, None, use_graph=True)
    dt, loss = timed_steps(Feed(tr, bt, dev), steps, warmup, torch.cuda.synchronize)
    assert np.isfinite(float(loss.item()))
    return {"workload": "C1 shape with bf16 storage (config C2's dtype): bf16 activations / weight shadow / MFMA, fp32 accumulate + masters + Adam",
            "dtype": "bf16", "value": round(a.batch * steps / dt, 1), "unit": "sequences/s", "ms_per_step": round(1e3 * dt / steps, 4),
            "steps": steps, "parity_gates": "tests/test_gpu_bf16.py: logits <= 5e-3 rel-Linf, loss <= 5e-4 rel, grads <= 2e-2 rel-L2 vs the fp32 oracle; "
                                            "KAT-1 Beauty metrics equal to 4 decimals"}


def secondary_beauty_bf16(dev, D, BSARecModel, Trainer, steps=100, warmup=10):
    """BASELINE config 2 at ITS OWN shape: Amazon-Beauty (the real interaction sequences, carried by the committed fixture
    tests/golden/kat_Beauty.npz -- the reference mount does not exist on the GPU box), V = 12,102, L = 50, d = 64, 2 layers,
    1 head, c = 5, alpha = 0.7, lr 5e-4 (reference README.md:43-48), B = 256, bf16 storage."""
    import numpy as np
    import torch
    path = os.path.join(ROOT, "tests", "golden", "kat_Beauty.npz")
    z = np.load(path)
    cfg = json.loads(str(z["cfg"]))
    off, items = z["seq_offsets"], z["seq_items"]
    seqs = [items[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]
    a2 = argparse.Namespace(item_size=cfg["item_size"], hidden=64, seq_len=50, batch=256, layers=2, heads=cfg["num_attention_heads"], dtype="bf16")
    m2 = model_args(a2)
    m2.c, m2.alpha, m2.lr = cfg["c"], cfg["alpha"], 5e-4
    torch.manual_seed(42)
    model = BSARecModel(m2).to(dev)
    model.set_seed(42, 0)
    model.train()
    u, x, y = D.train_table(seqs, 50)
    bt = D.DeviceBatches(u, x, y, 256, dev, shuffle=True, seed=42)
    tr = Trainer(model, bt, None, None, m2, None, use_graph=True)
    dt, loss = timed_steps(Feed(tr, bt, dev), steps, warmup, torch.cuda.synchronize)
    assert np.isfinite(float(loss.item()))
    fl = train_flops_per_seq(a2, cb=cfg["c"] // 2 + 1)
    return {"workload": f"C2: Amazon-Beauty (real sequences, {len(y)} training samples, {float((x == 0).mean()):.2f} padding), V={cfg['item_size']} "
                        f"L=50 d=64 2 layers 1 head c={cfg['c']} alpha={cfg['alpha']}, B=256, bf16 storage",
            "dtype": "bf16", "value": round(256 * steps / dt, 1), "unit": "sequences/s", "ms_per_step": round(1e3 * dt / steps, 4), "steps": steps,
            "step_bf16_mfma_frac": round(fl * 256 * steps / dt / (BF16_MFMA_PEAK_TFLOPS * 1e12), 5),
            "parity": "tests/test_gpu_config_shapes.py::test_c2_beauty_shape_bf16_training_step_vs_fp32_oracle"}


def secondary_c4(dev, D, BSARecModel, Trainer, steps=40, warmup=5):
    """BASELINE config 4's PER-RANK step (Yelp shape: V = 20,034, L = 50, d = 64, 2 layers, 2 heads; global batch 8,192 over
    8 ranks = 1,024 sequences per rank), fp32, on one GPU -- the 8-way shard itself needs a node; synthetic Yelp-shaped data
    (the real file is not on the GPU box)."""
    import numpy as np
    import torch
    a4 = argparse.Namespace(item_size=20034, hidden=64, seq_len=50, batch=1024, layers=2, heads=2, dtype="f32")
    m4 = model_args(a4)
    torch.manual_seed(42)
    model = BSARecModel(m4).to(dev)
    model.set_seed(42, 0)
    model.train()
    seqs = D.synth_ml1m_like(seed=7, n_users=3000, n_items=20033)
    u, x, y = D.train_table(seqs, 50)
    bt = D.DeviceBatches(u, x, y, 1024, dev, shuffle=True, seed=42)
    tr = Trainer(model, bt, None, None, m4, None, use_graph=True)
    tr.steps_per_graph = 4
    dt, loss = timed_steps(Feed(tr, bt, dev), steps, warmup, torch.cuda.synchronize)
    assert np.isfinite(float(loss.item()))
    fl = train_flops_per_seq(a4, cb=2)
    return {"workload": "C4 per-rank step: Yelp shape V=20034 L=50 d=64 2 layers 2 heads, B=1024 on ONE GPU (1/8 of the global batch 8192), fp32",
            "value": round(1024 * steps / dt, 1), "unit": "sequences/s", "ms_per_step": round(1e3 * dt / steps, 4), "steps": steps,
            "step_mfma_frac": round(fl * 1024 * steps / dt / (FP32_MFMA_PEAK_TFLOPS * 1e12), 5),
            "parity": "tests/test_gpu_config_shapes.py (oracle parity at V=20034; 2 ranks x 1024 == one process x 2048)"}


def secondary_c3(dev, seqs, D, BSARecModel, Trainer, steps=15, warmup=3, dtype="f32"):
    """BASELINE config 3: the C1 interactions re-cut with L = 200, hidden 256, 4 heads, 4 layers, B = 256 --
    generic tiled kernels.  Same step definition as the headline.  dtype "f32": the reference's arithmetic; "bf16": the
    products of the block stack on bf16 MFMAs (operands rounded while staged into LDS, fp32 tensors / accumulation / head /
    Adam -- csrc/gemm.h), judged at the bf16 gates of tests/test_gpu_bf16_generic.py."""
    import numpy as np
    import torch
    a3 = argparse.Namespace(item_size=3417, hidden=256, seq_len=200, batch=256, layers=4, heads=4, dtype=dtype)
    m3 = model_args(a3)
    torch.manual_seed(42)
    model = BSARecModel(m3).to(dev)
    model.set_seed(42, 0)
    model.train()
    u, x, y = D.train_table(seqs[:600], 200)                     # a slice of the users: enough full batches, quick to build
    bt = D.DeviceBatches(u, x, y, 256, dev, shuffle=True, seed=42)
    tr = Trainer(model, bt, None, None, m3, None, use_graph=True)
    tr.steps_per_graph = 1                                       # 17 ms per step: nothing to gain from grouping
    dt, loss = timed_steps(Feed(tr, bt, dev), steps, warmup, torch.cuda.synchronize)
    assert np.isfinite(float(loss.item()))
    fl = train_flops_per_seq(a3, cb=2)
    if dtype == "bf16":
        return {"workload": "C3 in bf16: L=200 d=256 4 heads 4 BSARec layers, B=256, generic tiled kernels with bf16 products "
                            "(fp32 tensors in HBM, operands rounded to bf16 in LDS, fp32 accumulate / LayerNorm / softmax / head / Adam)",
                "dtype": "bf16", "value": round(256 * steps / dt, 1), "unit": "sequences/s", "ms_per_step": round(1e3 * dt / steps, 3),
                "steps": steps, "step_bf16_mfma_frac": round(fl * 256 * steps / dt / (BF16_MFMA_PEAK_TFLOPS * 1e12), 5),
                "parity_gates": "tests/test_gpu_bf16_generic.py: logits <= 5e-3 rel-Linf, loss <= 5e-4 rel, grads <= 2e-2 rel-L2 "
                                "vs the reference's fp32 goldens and the fp32 oracle"}
    return {"workload": "C3: C1 interactions re-cut, L=200 d=256 4 heads 4 BSARec layers, B=256, fp32, generic tiled kernels",
            "value": round(256 * steps / dt, 1), "unit": "sequences/s", "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
            "step_mfma_frac": round(fl * 256 * steps / dt / (FP32_MFMA_PEAK_TFLOPS * 1e12), 5)}


if __name__ == "__main__":
    main()
