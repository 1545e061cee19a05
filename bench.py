#!/usr/bin/env python3
"""Training-throughput bench of the HIP BSARec path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = Trainer.iteration's per-batch body (src/trainers.py:100-107) on one batch of the
ML-1M-shaped synthetic workload (C1 of SURVEY 8d: V=3417, L=50, d=64, 2 layers, 2 heads, c=3,
alpha=0.9, dropout 0.5, Adam lr 1e-3, B=256 sequences per GPU): device-side batch gather,
forward, full-catalogue CE, backward, (gradient all-reduce for N > 1), fused Adam.  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, spec


def model_args(a):
    return argparse.Namespace(
        item_size=a.item_size, hidden_size=a.hidden, max_seq_length=a.seq_len, batch_size=a.batch,
        hidden_dropout_prob=0.5, attention_probs_dropout_prob=0.5, num_hidden_layers=a.layers,
        num_attention_heads=a.heads, hidden_act="gelu", initializer_range=0.02, c=3, alpha=0.9, seed=42,
        lr=1e-3, adam_beta1=0.9, adam_beta2=0.999, weight_decay=0.0, no_cuda=False, log_freq=1)


def train_flops_per_seq(a, cb=2):
    d, L, N, V = a.hidden, a.seq_len, a.layers, a.item_size
    return 3.0 * (N * L * (24 * d * d + 4 * L * d + 8 * cb * d) + 2 * d * V)


def cpu_baseline(a, budget_s=12.0):
    """The CPU oracle (numpy restatement, validated against the imported reference) timed on this
    box's host cores on a bounded sample of the same workload: whole training steps at C1 shape."""
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
    return {"value": round(n * a.batch / dt, 1), "unit": "sequences/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{n} full training steps (fwd+bwd+Adam, dropout on) of B={a.batch} at the C1 shape, "
                      f"numpy oracle with BLAS threads on all host cores, {dt:.1f} s"}


def fused_shape(a):
    return a.hidden == 64 and a.seq_len <= 64


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
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dp", action="store_true", help="take the data-parallel step (RCCL all-reduce) even with one rank")
    a = ap.parse_args()

    # stdout carries exactly ONE line (the JSON): everything else a library prints there -- RCCL's version banner at
    # communicator creation, for one -- is sent to stderr by pointing fd 1 at fd 2 for the duration of the run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # BSAREC_DIST_BACKEND=gloo: rehearsal of the N > 1 path with all ranks on ONE GPU (the build box has one); the
    # exchange then goes through the host, so the step uses grad graph + eager all-reduce + Adam graph
    backend = os.environ.get("BSAREC_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local = 0
        os.environ.setdefault("BSAREC_DP_GRAPH", "two")
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
    torch.manual_seed(42)                               # identical replicas on every rank
    model = BSARecModel(margs).to(dev)
    model.set_seed(42, rank)
    # ML-1M-shaped synthetic interactions -> device-resident sample table (identical on every rank)
    seqs = D.synth_ml1m_like(seed=42, n_items=a.item_size - 1)
    users, inputs, answers = D.train_table(seqs, a.seq_len)
    batches = D.DeviceBatches(users, inputs, answers, a.batch, dev, shuffle=True, seed=42, rank=rank, world=world)
    trainer = Trainer(model, batches, None, None, margs, None, use_graph=not a.no_graph, process_group=pg)
    use_graph = trainer.use_graph

    # steps come straight off the device-resident table: per step ONE C call (gather + fwd + CE + bwd + Adam),
    # replayed from a hipGraph at N = 1; eager gather + fwd/bwd + all-reduce + Adam for N > 1
    perm_state = {"perm": None, "pos": 0}
    B = a.batch
    cursor = torch.zeros(1, dtype=torch.int64, device=dev)
    perm_buf = torch.zeros(len(answers), dtype=torch.int64, device=dev)
    n_local = [0]

    def new_epoch():
        perm = batches.local_permutation()
        batches.epoch += 1
        n_local[0] = (perm.shape[0] // B) * B               # full batches only inside the timed region
        perm_buf[:perm.shape[0]].copy_(perm)
        cursor.zero_()
        perm_state["pos"] = 0

    def one_step():
        if perm_state["pos"] + B > n_local[0]:
            new_epoch()
        perm_state["pos"] += B
        return trainer.indexed_step(batches, perm_buf, cursor, None)

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

    model.train()
    for _ in range(max(a.warmup, 1)):
        loss = one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = one_step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())
    assert np.isfinite(final_loss), "training diverged"

    out = {
        "metric": "train sequences/sec, ML-1M L=50 d=64 2-layer", "value": round(a.batch * world * a.steps / dt, 1),
        "unit": "sequences/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"C1: ML-1M-shaped synthetic (6040 users, V={a.item_size}), L={a.seq_len} d={a.hidden} "
                               f"{a.layers} BSARec layers, {a.heads} heads, c=3 alpha=0.9 dropout=0.5, Adam lr=1e-3; "
                               "fwd + full-catalogue CE + bwd + Adam per step",
                   "batch_per_gpu": a.batch, "global_batch": a.batch * world, "seq_len": a.seq_len,
                   "parallelism": f"dp{world}", "launch": ("hipGraph replay" if use_graph else "eager") +
                             (f" ({trainer.dp_graph} graph per step incl. RCCL all-reduce)" if pg is not None and use_graph and trainer.dp_graph == "one"
                              else " (grad graph + eager RCCL all-reduce + Adam graph)" if pg is not None and use_graph else ""),
                   "final_loss": round(final_loss, 4)},
    }
    out["config"]["top_block"] = ("loss path evaluates the top BSARecBlock on position L-1 only (it still attends to every "
                                  "position); exact, same loss and gradients (SURVEY C.6); FLOP-based fractions below use the "
                                  "un-pruned counts") if getattr(model, "_plans", None) is not None and fused_shape(a) and a.layers >= 2 and \
        os.environ.get("BSAREC_PRUNE_TOP", "1") != "0" else "full"
    if rank == 0 and world == 1 and out["config"]["top_block"] != "full" and not a.no_roofline:
        # the same step with the FULL top-block kernels (nothing uses the one-row structure), measured in this run
        Lb.load().bsarec_set_prune_top(0)
        try:
            torch.manual_seed(42)
            model2 = BSARecModel(margs).to(dev)
            model2.set_seed(42, rank)
            model2.train()
            batches2 = D.DeviceBatches(users, inputs, answers, a.batch, dev, shuffle=True, seed=42, rank=rank, world=world)
            trainer2 = Trainer(model2, batches2, None, None, margs, None, use_graph=not a.no_graph, process_group=pg)
            perm2 = batches2.local_permutation()
            nfull = (perm2.shape[0] // B) * B
            pbuf2 = torch.zeros(len(answers), dtype=torch.int64, device=dev)
            pbuf2[:perm2.shape[0]].copy_(perm2)
            cur2 = torch.zeros(1, dtype=torch.int64, device=dev)
            nst = min(a.steps, nfull // B - max(a.warmup, 1) - 1)
            for _ in range(max(a.warmup, 1)):
                trainer2.indexed_step(batches2, pbuf2, cur2, None)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(nst):
                l2 = trainer2.indexed_step(batches2, pbuf2, cur2, None)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            assert np.isfinite(float(l2.item()))
            out["full_top_block"] = {"value": round(a.batch * nst / dt2, 1), "ms_per_step": round(1e3 * dt2 / nst, 4), "steps": nst}
            del trainer2, model2, batches2
        finally:
            Lb.load().bsarec_set_prune_top(1)
    flops_seq = train_flops_per_seq(a)
    out["step_mfma_frac"] = round(flops_seq * a.batch * world * a.steps / dt / (FP32_MFMA_PEAK_TFLOPS * 1e12 * world), 5)

    if rank == 0 and world == 1 and not a.no_roofline:
        # Per-kernel roofline: hipEvent pairs on the launch stream around every launch of one kernel class
        # inside real (eager) training steps; the class with the largest time per step is the dominant kernel.
        import ctypes as C
        lib = Lb.load()
        d, L, B, N, cb = a.hidden, a.seq_len, a.batch, a.layers, 2
        T = B * L
        fused = (d == 64 and L <= 64)
        if fused:
            cands = [(Lb.K_FUSED_BWD, "fused_layer_bwd_kernel (whole BSARecBlock input-gradient chain per sequence)",
                      B * L * (24 * d * d + 8 * L * d + 16 * cb * d)),
                     (Lb.K_FUSED_FWD, "fused_layer_fwd_kernel (whole BSARecBlock forward per sequence)",
                      B * L * (24 * d * d + 4 * L * d + 8 * cb * d)),
                     (Lb.K_DW1, "dw_direct_kernel (weight + bias gradients, direct split-K; un-pruned FLOP count of all N blocks over "
                                "its N-1 launches: the top block's products use the one-row structure of their gradient and ride "
                                "in the next block's launch)",
                      24.0 * T * d * d * (N / max(N - 1, 1) if out["config"]["top_block"] != "full" else 1.0))]
        else:
            cands = [(Lb.K_FFN1, "gemm_kernel<NT, EpiLinear<bias>> (FFN dense_1)", 2.0 * T * d * 4 * d),
                     (Lb.K_DW1, "gemm_grouped_tn_kernel (6 weight + bias gradients of a block, split-K)", 24.0 * T * d * d)]
        rows = []
        ovh = C.c_double()
        lib.bsarec_profile_event_overhead(C.c_void_p(torch.cuda.current_stream().cuda_stream), 200, C.byref(ovh))
        ovh_s = ovh.value * 1e-3          # an empty event bracket: what the two marker packets themselves cost
        for kclass, name, fl in cands:
            lib.bsarec_profile_select(kclass)
            nprof = 10
            for _ in range(nprof):
                _, ids, ans, _, _ = next(stream)
                trainer._step_eager(ids, ans)
            torch.cuda.synchronize()
            ms, n = C.c_double(), C.c_int()
            lib.bsarec_profile_read(C.byref(ms), C.byref(n))
            if n.value == 0:
                continue
            avg_s = ms.value * 1e-3 / n.value - ovh_s
            rows.append({"kernel": name, "launches_per_step": n.value / nprof, "avg_us": round(avg_s * 1e6, 3),
                         "us_per_step": round(avg_s * 1e6 * n.value / nprof, 2), "flops_per_launch": float(fl),
                         "achieved": round(fl / avg_s / 1e12, 3)})
        lib.bsarec_profile_select(Lb.K_NONE)
        rows.sort(key=lambda r: -r["us_per_step"])
        top = rows[0]

        def pmc_traffic(kernel_prefix):
            # HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/): WRITE_SIZE + 2 * FETCH_SIZE KiB
            # (gfx950 tallies wide streaming reads at half their bytes, MI355X guide, HBM section)
            path = os.path.join(ROOT, "profiles", "r01_f_pmc_C1.csv")
            if not os.path.exists(path) or not fused or a.batch != 256:
                return None
            vals = {}
            for line in open(path):
                if line.startswith('"' + kernel_prefix):
                    _, counter, avg, _ = line.rsplit(",", 3)
                    vals[counter] = float(avg)
            if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
                return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
            return None
        out["roofline"] = {"bound": "mfma", "kernel": top["kernel"], "achieved": top["achieved"],
                           "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(top["achieved"] / FP32_MFMA_PEAK_TFLOPS, 5),
                           "traffic": pmc_traffic(top["kernel"].split(" ")[0]),
                           "avg_us": top["avg_us"], "event_overhead_us": round(ovh_s * 1e6, 3), "launches_per_step": top["launches_per_step"],
                           "flops_per_launch": top["flops_per_launch"], "other_kernels": rows[1:]}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a)
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)
    if pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
