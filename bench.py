#!/usr/bin/env python3
"""SpectralNet training throughput on a synthetic Amazon2M-shape graph (SPARC hot path).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (launched by torch.distributed.run for N > 1).  Every rank generates
the same 2.45 M-node planted-cluster graph in HBM, extracts and prepares ITS share of the
50-cluster training batches with the device kernels (sub-block, dual-value CSR, affinity
vectors), then runs W untimed and K timed optimisation steps (orthogonality step on one
local batch, gradient step on another, flat-gradient all-reduce, Adam).  Per-GPU work is
fixed as N grows (weak scaling).  Rank 0 prints one JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
F64_MFMA_PEAK_TFLOPS = 78.6  # MI355X dense FP64 matrix peak (vendor data sheet)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def cpu_baseline(graph, batches_nodes, bounds, shape, arch, sample_nodes, threads, full_nodes=0):
    """The dense torch restatement of the reference step (the oracle) on the host CPU,
    on a bounded sample: the first `sample_nodes` nodes of two training batches."""
    from oracle import sparc_oracle as O
    from sparc_amd import synth
    torch.set_num_threads(threads)
    lam = shape["diag_lambda"]
    dims = [shape["din"], arch["hidden_dim1"], arch["hidden_dim2"], arch["hidden_dim3"], arch["output_dim"]]

    def one_step(n_nodes):
        pt1 = batches_nodes[0][:n_nodes]
        pt2 = batches_nodes[1 % len(batches_nodes)][:n_nodes]
        coords, values, shp, _ = graph.subblock(pt1, mode=(0 if lam == -1 else 2), binarize=False, diag_lambda=lam)
        support = (coords.cpu().numpy(), values.cpu().numpy(), shp)
        x1 = synth.block_features(pt1, bounds, shape["din"], 47)
        x2 = synth.block_features(pt2, bounds, shape["din"], 47)
        torch.manual_seed(0)
        params = []
        for i in range(4):
            lin = torch.nn.Linear(dims[i], dims[i + 1])
            params += [lin.weight.detach(), lin.bias.detach()]
        t0 = time.perf_counter()
        out = O.train_step_dense(params, x1, support, x2, dtype=torch.float32)
        O.adam_update(params, out["grads"], {})
        return time.perf_counter() - t0, len(pt1), len(pt2)

    one_step(min(1024, sample_nodes))      # thread pools and allocator warm before the timed step
    t, n1, n2 = one_step(sample_nodes)
    pt1 = batches_nodes[0][:sample_nodes]
    out = {"value": n1 / t, "unit": "nodes/s", "cores": threads, "kind": "port",
           "sample": f"dense reference-step restatement (oracle), float32, one step on the first {n1} nodes of a "
                     f"training batch (+{n2}-node orthonormalisation batch), timed once after a 1024-node warm-up, "
                     f"{t:.2f} s/step"}
    if full_nodes > len(pt1):
        # the dense step is dominated by three m x m x m products: time grows with the cube of the batch size
        t_full = t * (full_nodes / len(pt1)) ** 3
        out["full_batch_extrapolation"] = {"nodes": full_nodes, "seconds_per_step": t_full, "nodes_per_s": full_nodes / t_full,
                                           "rule": "sample time x (m / sample)^3: the dense affinity costs 3 m^3 multiply-adds"}
    return out


def inner_pass_bytes(b, k):
    """(algorithmic, moved, algorithmic with mirrored values) bytes of ONE inner pass of the Laplacian chain on batch b
    at spectral dimension k.  Algorithmic: every stored entry once as index + value (8 bytes), the row pointers, and
    three dense operands of width 2k (gathered operand, own-row operand, result) at the real k.  Moved: what the
    kernel that runs actually reads and writes (its column padding, its own entry format).  The third figure counts
    4 more bytes per entry for supports whose mirrored values differ (A[j][i] stored beside A[i][j])."""
    entry = b.nnz * (4 + 4) + (b.m + 1) * 8
    mirrored = b.nnz * 4 if b.valT is not None else 0
    path = b.pass_kernel(k, True)
    t = b.tiling(k, True) if path == "tiles" else None
    kp = t.kp if t is not None else (16 if k <= 16 else 32 if k <= 32 else 64)
    moved = entry + mirrored + 3 * b.m * 2 * kp * 4
    if t is not None and getattr(t, "pattern", False):   # row-factor form: addresses only, plus 16 bytes of row factors per row
        moved = b.nnz * 4 + (b.m + 1) * 8 + b.m * 16 + 3 * b.m * 2 * kp * 4
    if path == "blocks":   # the blocks as dense arrays, the other entries as CSR, and the intermediate result written and re-read
        bl = b.blocks(k, True)
        per = 4 + 4 + (4 if b.valT is not None else 0)
        n_blk = sum(int(d[2]) * 32 for d in bl["keep"][0].cpu().numpy()) * (2 if b.valT is not None else 1)
        moved = n_blk * 4 + bl["r_nnz"] * per + (b.m + 1) * 8 + 5 * b.m * 2 * kp * 4
    dense = 3 * b.m * 2 * k * 4
    return entry + dense, moved, entry + mirrored + dense


def recorded_traffic(workload, b, k):
    """(HBM bytes per launch of the inner pass, where that figure was recorded) from the rocprofv3 counter passes
    (tools/run_pmc.sh, tools/collect_traffic.py): a recorded figure, not a measurement of this run, and only given
    if it was taken on this workload's batch shape; else (None, None)."""
    path = os.path.join(ROOT, "profiles", "spmm_hbm_traffic.json")
    try:
        rec = json.load(open(path))
    except Exception:
        return None, None
    same = rec.get("workload") == workload and rec.get("k") == k and abs(rec.get("m", 0) - b.m) <= 0.05 * b.m and \
        abs(rec.get("nnz", 0) - b.nnz) <= 0.05 * b.nnz
    if not same:
        return None, None
    return rec.get("bytes_per_launch"), (f"profiles/spmm_hbm_traffic.json: rocprofv3 --pmc passes over an inner pass on a {rec.get('workload')}-shape "
                                         f"batch of {rec.get('m')} rows, {rec.get('nnz')} entries, k = {rec.get('k')} (not this run)")


def side_measurements(trainer, lib, check, k, dev, reps=30):
    """Outside the timed region: average duration of the last pass of the chain (loss + gradient epilogue)
    and of the Gram product of the orthonormalisation, each on batch 0, with events on the launch stream."""
    import ctypes
    from sparc_amd.ops import gram
    b = trainer.batches[0]
    Y = torch.randn(b.m, k, device=dev)
    for _ in range(3):
        b.lap(Y, True)
    torch.cuda.synchronize()
    check(lib.sparc_profile_begin(4, reps), "sparc_profile_begin")
    for _ in range(reps):
        b.lap(Y, True)
    torch.cuda.synchronize()
    tot, n = ctypes.c_double(0.0), ctypes.c_int64(0)
    check(lib.sparc_profile_end(ctypes.byref(tot), ctypes.byref(n)), "sparc_profile_end")
    lap_us = tot.value / max(n.value, 1) * 1e3
    alg, _, _ = inner_pass_bytes(b, k)
    lap_alg = alg - b.m * 2 * k * 4 + b.m * k * 4 + b.m * 32     # writes the gradient [m][k] instead of [m][2k]; reads the row records
    for _ in range(3):
        gram(Y, Y)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        gram(Y, Y)
    ev[1].record()
    torch.cuda.synchronize()
    gram_us = ev[0].elapsed_time(ev[1]) / reps * 1e3
    flops = 2.0 * b.m * k * k

    def timed(fn):
        for _ in range(3):
            fn()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for _ in range(reps):
            fn()
        e[1].record()
        torch.cuda.synchronize()
        return e[0].elapsed_time(e[1]) / reps * 1e3

    # in a training step the Gram product of the orthogonality batch rides in the output layer's forward launch and the
    # one of the backward shares its launch with the product by the orthonormalisation weights
    from sparc_amd.ops import gram_apply, linear_narrow_fwd, orth_apply
    layer = trainer.net.layers[-1][0]
    W, bias = layer.weight.detach(), layer.bias.detach()
    h = torch.randn(2 * b.m, W.shape[1], device=dev)
    fused = {}
    if linear_narrow_fwd(h, W, bias, 1) is not None and linear_narrow_fwd(h, W, bias, 1, gram_rows=b.m)[1] is not None:
        fused["output_layer_forward_us"] = timed(lambda: linear_narrow_fwd(h, W, bias, 1))
        fused["output_layer_forward_with_gram_us"] = timed(lambda: linear_narrow_fwd(h, W, bias, 1, gram_rows=b.m))
    Wo = torch.randn(k, k, device=dev)
    Z = torch.empty_like(Y)
    fused["apply_us"] = timed(lambda: orth_apply(Y, Wo, transpose=True, out=Z))
    fused["gram_with_apply_us"] = timed(lambda: gram_apply(Y, Y, Wo, True, Z))
    return {
        "lap_last_pass": {"avg_kernel_us": lap_us, "algorithmic_bytes": lap_alg,
                          "achieved_GBs": lap_alg / (lap_us * 1e-6) / 1e9, "frac_of_hbm_peak": lap_alg / (lap_us * 1e-6) / 1e9 / HBM_PEAK_GBS},
        "gram": {"avg_us_two_launches": gram_us, "flop": flops, "achieved_TFLOPs_f64": flops / (gram_us * 1e-6) / 1e12,
                 "f64_mfma_peak_TFLOPs": F64_MFMA_PEAK_TFLOPS, "mfma_utilisation": flops / (gram_us * 1e-6) / 1e12 / F64_MFMA_PEAK_TFLOPS,
                 "achieved_GBs": 2.0 * b.m * k * 4 / (gram_us * 1e-6) / 1e9,
                 "note": "gram_partial + gram_reduce back to back; the product reads 2*m*k*4 bytes for 2*m*k^2 flop, i.e. it is "
                         "HBM/launch bound long before the matrix cores matter",
                 "in_a_training_step": fused},
    }


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes through
    torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) with this command line, let them print to this
    process's stdout / stderr, and return their exit status.  Must run before anything initialises the GPU here."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs between the ranks on these hosts
    print(f"[bench] starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def keep_stdout_for_the_result():
    """Libraries under the ranks write to file descriptor 1 themselves (RCCL prints its version banner there when the
    first communicator forms).  Point descriptor 1 at stderr for the rest of the process and return a file on the
    original stdout: the result line written to it is then the only thing stdout carries."""
    sys.stdout.flush()
    result = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return result


def check_launch(world: int, rank: int, local_rank: int, backend: str, result=sys.stdout) -> None:
    """--check-launch: the ranks meet, sum their rank numbers and rank 0 reports what it saw; no kernel runs.  On a
    host without a GPU the group is formed over gloo."""
    has_gpu = torch.cuda.is_available()
    if not has_gpu:
        backend = "gloo"
    dev = torch.device("cuda", int(os.environ.get("SPARC_BENCH_DEVICE", local_rank))) if has_gpu else torch.device("cpu")
    total = torch.tensor([float(rank)], device=dev if backend == "nccl" else "cpu")
    if world > 1:
        if has_gpu:
            torch.cuda.set_device(dev)
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
        dist.all_reduce(total)
        seen = dist.get_world_size()
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen = 1
    if rank == 0:
        print(json.dumps({"launch_check": {"world": seen, "env_world": world, "backend": backend if world > 1 else None,
                                           "rank_sum": float(total[0]), "device": str(dev)}}), file=result, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="amazon2m")
    ap.add_argument("--batches-per-rank", type=int, default=12,
                    help="training batches each rank prepares and cycles through (100 batches / 8 GPUs)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the graph (testing only)")
    ap.add_argument("--cpu-sample", type=int, default=16384)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hip-graph", action="store_true", help="replay whole steps as HIP graphs (one GPU; small batches)")
    ap.add_argument("--kernel-choice", choices=["rule", "measure"], default="rule",
                    help="rule: a batch's kernel follows from its structure (runs repeat bit for bit, ranks agree); "
                         "measure: time the candidates on every batch and keep the fastest")
    ap.add_argument("--check-launch", action="store_true",
                    help="only start the ranks, form the process group, all-reduce one number and report the world")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))     # nothing has touched the GPU in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start the ranks with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}, or leave WORLD_SIZE unset and bench.py starts them itself")
    # rehearsal knobs for a one-GPU box: several ranks on one card over gloo (never used by the driver)
    backend = os.environ.get("SPARC_BENCH_BACKEND", "nccl")
    result = keep_stdout_for_the_result()
    if args.check_launch:
        return check_launch(world, rank, local_rank, backend, result)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    if "SPARC_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["SPARC_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        print(f"[bench] rank {dist.get_rank()} of {dist.get_world_size()} ({dist.get_backend()}): {dev}, "
              f"{torch.cuda.get_device_name(dev)}", file=sys.stderr, flush=True)

    from sparc_amd import synth
    from sparc_amd._lib import lib, check
    from sparc_amd.graph import ClusterBatch
    from sparc_amd.parallel import ShardedSpectralTrainer, shard_batches

    shape = dict(synth.SHAPES[args.workload])
    n = max(int(shape["n"] * args.scale), 2000)
    n_edges = max(int(shape["edges"] * args.scale), 4000)
    n_clusters = max(int(shape["clusters"] * args.scale), shape["bsize"]) if shape["clusters"] > 1 else 1
    k, din, lam = shape["k"], shape["din"], shape["diag_lambda"]
    arch = {"n_layers": 6, "hidden_dim1": 1024, "hidden_dim2": 1024, "hidden_dim3": 512, "output_dim": k}

    t0 = time.perf_counter()
    graph, bounds = synth.planted_cluster_graph(n, n_edges, n_clusters, dev, seed=0)
    torch.cuda.synchronize()
    nodes_of, cluster_ids = synth.cluster_batches(bounds, shape["bsize"], seed=0, return_clusters=True)
    t_graph = time.perf_counter() - t0
    log(f"[bench] graph: {n} nodes, {int(graph.indices.numel())} stored entries, {len(nodes_of)} batches, {t_graph:.1f}s")

    mine = shard_batches(len(nodes_of), world, rank)[:args.batches_per_rank]
    if not mine:
        raise SystemExit(f"rank {rank} is left without a batch: the graph has {len(nodes_of)} batches for {world} ranks "
                         f"(raise --scale)")
    # SPARC_BENCH_REDUCE_ALONE=1 (rehearsal knob, never used by the driver): one rank forms an RCCL group of its own and
    # runs the data-parallel step's bucket copies and collectives, which shows their cost on a one-GPU box
    alone = world == 1 and os.environ.get("SPARC_BENCH_REDUCE_ALONE") == "1"
    if alone:
        import socket
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    trainer = ShardedSpectralTrainer(arch, din, dev, lr=1e-3, seed=0, hip_graph=args.hip_graph, reduce_in_a_world_of_one=alone)
    t0 = time.perf_counter()
    for b in mine:
        pt = nodes_of[b]
        coords, values, shp, _ = graph.subblock(pt, mode=(0 if lam == -1 else 2), binarize=False, diag_lambda=lam)
        csz = np.diff(bounds.cpu().numpy())[cluster_ids[b]]
        cb = ClusterBatch.from_device_coo(shp[0], coords, values, cluster_sizes=csz).warm(k, measure=args.kernel_choice == "measure")
        trainer.add_batch(synth.block_features(pt, bounds, din, 47), cb)
    torch.cuda.synchronize()
    t_prep = time.perf_counter() - t0
    nb = len(trainer.batches)
    ms = [b.m for b in trainer.batches]
    nnzs = [b.nnz for b in trainer.batches]
    log(f"[bench] rank {rank}: prepared {nb} batches in {t_prep:.1f}s (m ~ {int(np.mean(ms))}, nnz ~ {int(np.mean(nnzs))})")

    rng = np.random.RandomState(1234 + rank)

    def one_step(s):
        pid = s % nb
        pid_2 = rng.choice(nb)
        rng.permutation(ms[pid])                      # the reference draws it (src/SpectralTrainer.py:211)
        head = rng.permutation(ms[pid_2])[:k]         # (:218) only the first k entries matter
        return pid, trainer.step(pid, pid_2, head)

    pass_bytes = [inner_pass_bytes(b, k) for b in trainer.batches]   # (reads descriptors back: not inside the timed loop)
    for s in range(args.warmup):
        one_step(s)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # HIP events around the inner row-gather passes (epilogue 1), on the launch stream
    if not args.hip_graph:   # (events recorded inside a captured graph carry no time: measured after the run instead)
        # one event pair per step around both inner passes, on the first 32 steps of the timed region (a pair costs the
        # stream about as much as one of the passes it brackets: event_pair_with_nothing_between_us)
        check(lib.sparc_profile_begin(1, min(args.steps, 32)), "sparc_profile_begin")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nodes = 0
    alg_bytes = moved_bytes = alg12_bytes = 0.0
    losses = []
    for s in range(args.warmup, args.warmup + args.steps):
        pid, loss = one_step(s)
        nodes += ms[pid]
        a_b, m_b, a12_b = pass_bytes[pid]
        alg_bytes += 2 * a_b
        moved_bytes += 2 * m_b
        alg12_bytes += 2 * a12_b
        losses.append(loss)
    t_host = time.perf_counter() - t0     # the host's share: how long it took to enqueue the steps
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    log(f"[bench] rank {rank}: host enqueued {args.steps} steps in {t_host * 1e3 / args.steps:.2f} ms/step, "
        f"device finished them in {elapsed * 1e3 / args.steps:.2f} ms/step")
    tot_ms, launches = ctypes.c_double(0.0), ctypes.c_int64(0)
    if args.hip_graph:       # the same event bracket around the inner passes of eager calls on batch 0
        b0 = trainer.batches[0]
        Y0 = torch.randn(b0.m, k, device=dev)
        for _ in range(3):
            b0.lap(Y0, True)
        check(lib.sparc_profile_begin(1, args.steps), "sparc_profile_begin")
        for _ in range(args.steps):
            b0.lap(Y0, True)
        torch.cuda.synchronize()
    check(lib.sparc_profile_end(ctypes.byref(tot_ms), ctypes.byref(launches)), "sparc_profile_end")
    # what an event pair measures with nothing between its two records (the bracket's own share of avg_kernel_us)
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    for e0, e1 in pairs:
        e0.record()
        e1.record()
    torch.cuda.synchronize()
    empty_pair_us = float(np.median([e0.elapsed_time(e1) for e0, e1 in pairs])) * 1e3

    stats = torch.tensor([elapsed, float(nodes)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, nodes = float(mx[0]), float(sm[1])
    final_loss = float(torch.stack(losses).double().mean())
    if not np.isfinite(final_loss):
        raise SystemExit("non-finite loss in the timed region")

    if rank == 0:
        kern_ms = tot_ms.value / max(launches.value, 1)
        per_launch = alg_bytes / max(2 * args.steps, 1)   # (the average over all steps' batches; the batches cycle)
        achieved = per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        b0 = trainer.batches[0]
        path = b0.pass_kernel(k, True)
        t0_ = b0.tiling(k, True) if path == "tiles" else None
        row_factor_form = bool(t0_ is not None and getattr(t0_, "pattern", False))
        traffic, traffic_src = recorded_traffic(args.workload, b0, k)
        kernel = (f"spmm_win<vec {t0_.vec}, dual, add-x0{', row-factor form' if row_factor_form else ''}> (kp {t0_.kp})" if path == "tiles" else
                  "dense_blocks<add-x0> (matrix cores, the clusters' diagonal blocks) + spmm_rows<dual, add-x0> on the other entries (kp 64)"
                  if path == "blocks" else f"spmm_rows<dual, add-x0> (kp {16 if k <= 16 else 32 if k <= 32 else 64})") + \
            ": the two inner passes of the Laplacian chain"
        out = {
            "metric": "SpectralNet train nodes/sec + SpMM HBM GB/s (% roofline), Amazon2M shape",
            "value": nodes / elapsed, "unit": "nodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}-shape planted-cluster graph: {n} nodes, "
                                   f"{(int(graph.indices.numel()) - n) // 2} undirected edges, {n_clusters} clusters, "
                                   f"{shape['bsize']} clusters/batch, diag_lambda={lam}, MLP {din}-1024-1024-512-{k}",
                       "batch_nodes": int(np.mean(ms)), "batch_nnz": int(np.mean(nnzs)),
                       "batches_per_gpu": nb, "prep_seconds_per_batch": t_prep / max(nb, 1),
                       "parallelism": f"dp{world} (cluster batches sharded, flat-gradient all-reduce)" + (", steps replayed as HIP graphs" if args.hip_graph else ""),
                       "mean_loss": final_loss},
            "roofline": {"bound": "hbm", "kernel": kernel,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_recorded_on": traffic_src, "avg_kernel_us": kern_ms * 1e3,
                         "launches_timed": int(launches.value),
                         "launches_per_event_pair": 2, "event_pair_with_nothing_between_us": empty_pair_us,
                         "algorithmic_bytes_per_launch": per_launch,
                         "algorithmic_bytes_formula": "nnz*(4+4) + 8*(m+1) + 3*m*(2k)*4 (gathered operand, own-row operand, "
                                                      "result; width 2k = [A half | A^T half])",
                         "algorithmic_bytes_per_launch_with_mirrored_values": alg12_bytes / max(2 * args.steps, 1),
                         "bytes_moved_per_launch": moved_bytes / max(2 * args.steps, 1),
                         "note": ("row-factor form of the support: the value streams that the algorithmic bytes count are "
                                  "not read, so measured traffic lies below them") if row_factor_form else None},
        }
        if world == 1:
            out["other_kernels"] = side_measurements(trainer, lib, check, k, dev)
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(graph, nodes_of, bounds, shape, arch, args.cpu_sample, threads,
                                               full_nodes=int(np.mean(ms)))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), file=result, flush=True)
    if world > 1:
        dist.barrier()
    if world > 1 or alone:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
