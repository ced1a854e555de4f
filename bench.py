#!/usr/bin/env python
"""Headline benchmark: edges/s aggregated (fwd+bwd) per MI355X + HBM roofline of the
dominant kernel (BASELINE.json metric).

A "step" = ONE pass of the sparse hot path over the whole synthetic graph: GENConv
softmax_sg aggregation forward (dgcn_gen_aggr_fwd_f32) + its backward w.r.t. x
(dgcn_gen_aggr_bwd_f32) on an ogbn-products-shaped random graph
(N=2,449,029, E=126,167,309, C=128, t=0.1), inputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--shape products|arxiv|...]

N>1 (launched by torch.distributed.run, one rank per GPU): the SAME graph is partitioned by
destination range across ranks (strong scaling); every step all-gathers the feature shards
over RCCL and each rank aggregates its own destination rows (deep_gcns_torch_amd/dist.py).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL peer access)
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def fwd_bytes(E, N, C, saved=0):  # E*(4C+4) + N*4C + 4(N+1), + N*4C per array saved for backward
    return E * (4 * C + 4) + N * 4 * C * (1 + saved) + 4 * (N + 1)


def bwd_bytes(E, N, C):  # E*(8C+4) + N*12C
    return E * (8 * C + 4) + N * 12 * C


def cpu_baseline(shape_name: str, channels: int, t: float, budget_s: float = 20.0):
    """Oracle (CPU restatement of the reference path) timed on this host's cores on a bounded,
    uniformly down-scaled sample of the same workload (same average degree)."""
    from deep_gcns_torch_amd import synth
    from oracle import sparse_ref  # baseline leg only
    ncores = os.cpu_count() or 1
    torch.set_num_threads(ncores)
    s = synth.SHAPES[shape_name]
    scale = 128 if shape_name == "products" else 1
    n = s["n"] // scale
    nu = s["n_undirected"] // scale
    ei = synth.undirected_random_graph(n, nu, seed=s["seed"])
    E = ei.size(1)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, channels, generator=g).requires_grad_(True)
    go = torch.randn(n, channels, generator=g)

    def step():
        out = sparse_ref.gen_propagate(x, ei, aggr="softmax_sg", t=t)
        torch.autograd.grad(out, x, go)

    step()  # warm-up
    t0 = time.perf_counter()
    reps = 0
    while True:
        step()
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 3:
            break
    return dict(value=E * reps / el, unit="edges/s", cores=ncores, kind="port",
                sample=f"{shape_name}-shaped uniform graph scaled 1/{scale}: N={n}, E={E}, C={channels}, "
                       f"softmax_sg t={t}, fwd+bwd, {reps} reps in {el:.1f}s (oracle/sparse_ref.py, torch CPU)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shape", default="products")
    ap.add_argument("--graph", default="uniform", choices=["uniform", "powerlaw"])
    ap.add_argument("--channels", type=int, default=0)
    ap.add_argument("--aggr", default="softmax_sg")
    ap.add_argument("--t", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fwd-only", action="store_true", help="profiling aid: skip the backward")
    ap.add_argument("--force-partitioned", action="store_true",
                    help="run the RCCL multi-rank path even with one rank (sanity check)")
    ap.add_argument("--scheme", default="auto", choices=["auto", "transposed", "allgather", "halo"],
                    help="multi-rank exchange: channel-transposed all-to-all or destination-partitioned all-gather")
    ap.add_argument("--pipeline-chunks", type=int, default=0, help="0 = library default")
    ap.add_argument("--node-groups", type=int, default=0, help="transposed scheme: node groups (0 = library default)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from deep_gcns_torch_amd import ops, synth
    from deep_gcns_torch_amd.graph import Graph
    from deep_gcns_torch_amd import _lib
    _lib.load()

    dist = None
    partitioned = world > 1 or args.force_partitioned
    if partitioned:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    s = synth.SHAPES[args.shape]
    C = args.channels or s["channels"]
    n = s["n"]
    gen = synth.undirected_random_graph if args.graph == "uniform" else synth.powerlaw_graph
    ei = gen(n, s["n_undirected"], seed=s["seed"], device=dev)
    E = ei.size(1)

    gx = torch.Generator(device=dev).manual_seed(1234)
    x_full = torch.randn(n, C, device=dev, generator=gx)
    g_full = torch.randn(n, C, device=dev, generator=gx)

    if not partitioned:
        graph = Graph.from_edge_index(ei, n)
        del ei
        x = x_full.requires_grad_(True)

        def fwd():
            return ops.gen_aggregate(x, graph, aggr=args.aggr, t=args.t)

        def step():
            out = fwd()
            if not args.fwd_only:
                torch.autograd.grad(out, x, g_full)
            return out
    else:
        from deep_gcns_torch_amd import dist as ddist
        x = g_loc = None
        extra = dict(pipeline_chunks=args.pipeline_chunks) if args.pipeline_chunks else {}

        def make(scheme, node_groups):
            """(partition, x_local, g_local, fwd) for one exchange scheme; rows are re-sliced per scheme because the
            row ownership (bounds) differs between them."""
            part = ddist.build_partition(ei, n, C, rank, world, scheme=scheme, node_groups=node_groups)
            xl = x_full[part.lo:part.hi].clone().requires_grad_(True)
            gl = g_full[part.lo:part.hi].clone()
            return part, xl, gl, (lambda: ddist.aggregate(xl, part, aggr=args.aggr, t=args.t, **extra))

        def timed(fwd_fn, xl, gl, reps):
            torch.cuda.synchronize(dev); dist.barrier(); torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(reps):
                o = fwd_fn()
                if not args.fwd_only:
                    torch.autograd.grad(o, xl, gl)
            torch.cuda.synchronize(dev); dist.barrier(); torch.cuda.synchronize(dev)
            tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item()) / reps * 1e3

        wn_default = ddist.default_node_groups(C, world)
        if args.scheme == "auto":
            # the exchange is bound by the xGMI links and RCCL's per-collective efficiency: try the applicable schemes
            # for a few untimed steps and keep the fastest (same choice on every rank: max-over-ranks timings)
            cands = [("allgather", 1), ("halo", 1)]
            if world > 1 or args.force_partitioned:
                if ddist.transposed_supported(C, world, None, 1):
                    cands.append(("transposed", 1))
                if wn_default > 1 and ddist.transposed_supported(C, world, None, wn_default):
                    cands.append(("transposed", wn_default))
        else:
            cands = [(args.scheme, args.node_groups or (wn_default if args.scheme == "transposed" else 1))]
        tuned = {}
        best = None
        for sch, wn in cands:
            if len(cands) == 1:
                best = (0.0, sch, wn, make(sch, wn))
                break
            # a candidate that fails on any rank is dropped on all of them (the flag is agreed on with a MIN-reduce;
            # a failure in the middle of a collective cannot be recovered from and surfaces as the NCCL timeout)
            cand, ms, err = None, float("inf"), None
            try:
                cand = make(sch, wn)
                timed(cand[3], cand[1], cand[2], 2)
                ms = timed(cand[3], cand[1], cand[2], 3)
            except Exception as exc:   # noqa: BLE001 -- reported below, never silently
                err = repr(exc)[:200]
            okf = torch.tensor([0 if err else 1], device=dev, dtype=torch.int32)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if int(okf.item()) == 0:
                tuned[f"{sch}/node_groups={wn}"] = f"failed: {err or 'on another rank'}"
                cand = None
            else:
                tuned[f"{sch}/node_groups={wn}"] = round(ms, 3)
                if best is None or ms < best[0]:
                    best = (ms, sch, wn, cand)
                cand = None
            torch.cuda.empty_cache()
        if best is None:
            raise SystemExit(f"no exchange scheme ran: {tuned}")
        _, scheme, node_groups, (part, x, g_loc, fwd) = best
        transposed = isinstance(part, ddist.TransposedGraph)
        del ei, x_full, g_full

        def step():
            out = fwd()
            if not args.fwd_only:
                torch.autograd.grad(out, x, g_loc)
            return out

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    if dist is not None:          # RCCL prints a banner through C stdio at communicator creation: flush it now so
        import ctypes             # the JSON line stays the LAST line of stdout
        ctypes.CDLL(None).fflush(None)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # dominant kernel (forward aggregation) timed alone with events on the launch stream
    stream = torch.cuda.current_stream(dev)
    evs = []
    # same launch as inside the timed steps (training-mode forward: it also writes the array the backward needs --
    # log-sum-exp / pre-clamp mean / arg-max ids -- counted below), so rocprofv3's per-kernel average agrees
    for _ in range(max(5, min(args.steps, 20))):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        fwd()
        b.record(stream)
        evs.append((a, b))
    torch.cuda.synchronize(dev)
    fwd_ms = sorted(a.elapsed_time(b) for a, b in evs)
    fwd_ms_avg = sum(fwd_ms) / len(fwd_ms)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        saved = 1 if args.aggr.split("_")[0] in ("softmax", "power", "max") else 0
        if not partitioned:
            algo = fwd_bytes(E, n, C, saved)
        else:
            if transposed:
                algo = fwd_bytes(part.n_edges, n // part.node_groups, C // part.channel_groups, saved)
            else:
                algo = fwd_bytes(part.n_local_edges, part.hi - part.lo, C, saved)
        achieved = algo / (fwd_ms_avg * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tfile) and not partitioned:
            try:
                tj = json.load(open(tfile))
                if tj.get("shape") == args.shape and tj.get("graph") == args.graph and tj.get("channels") == C:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "edges/sec aggregated (fwd+bwd) per MI355X; achieved HBM GB/s vs roofline",
            "value": E * args.steps / elapsed if not args.fwd_only else E * args.steps / elapsed,
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"GENConv {args.aggr} aggregation (t={args.t}) fwd+bwd, ogbn-{args.shape}-shaped "
                            f"{args.graph} random graph N={n} E={E} C={C}"
                            + (" [fwd only]" if args.fwd_only else ""),
                "parallelism": ("single GPU" if not partitioned else
                                (f"node-partitioned rows x{world}, channel-transposed exchange (RCCL all-to-all in/out; "
                                 f"{part.node_groups} node group(s) x {part.channel_groups} channel group(s): each rank "
                                 f"aggregates {part.n_edges} edges for {C // part.channel_groups} channels)" if transposed else
                                 (f"destination-partitioned x{world}, halo rows only (RCCL all-to-all, {part.n_halo} halo rows "
                                  f"on rank 0)" if isinstance(part, ddist.HaloGraph) else
                                  f"destination-partitioned x{world}, RCCL all-gather fwd / reduce-scatter bwd"))),
            },
            "roofline": {
                "kernel": "gen_aggr_fwd_kernel<SOFTMAX> (dgcn_gen_aggr_fwd_f32)"
                          + (" + exchange (whole forward of one rank)" if partitioned else ""),
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": algo,
                "launch_ms_avg": fwd_ms_avg,
                "launch_ms_min": fwd_ms[0],
                "frac_of_measured_copy_6290GBs": achieved / 6290.0,
            },
            "fwd_edges_per_s": (E if not partitioned else (part.n_edges if transposed else part.n_local_edges)) / (fwd_ms_avg * 1e-3),
        }
        if partitioned and tuned:
            res["config"]["autotuned_ms_per_step"] = tuned
        if world == 1 and not partitioned and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.shape, C, args.t)
        print(json.dumps(res), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
