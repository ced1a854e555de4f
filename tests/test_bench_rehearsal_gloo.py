"""CPU, EIGHT ranks over gloo: ``bench.py --gpus 8 --rehearsal`` -- the driver's own multi-GPU command line with the
backend swapped for gloo, CPU tensors and the oracle as the rank-local kernel, on a 1/256-scale ogbn-products-shaped
graph.  What an 8-GPU node would run for the first time otherwise: the destination partition of every scheme, the
candidate agreement of the scheme autotuner (a MIN-reduce over ranks per candidate), the barrier-bracketed timed loop
with the MAX over ranks, the per-phase breakdown and the one JSON line of rank 0.  No timing claim is attached to it."""
import json
import os
import subprocess
import sys

import pytest

from test_dist_gloo import _free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,graph", [(8, "uniform"), (4, "powerlaw")])
def test_bench_command_line_at_world_8_on_cpu(world, graph):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--rehearsal", "--scale-div", "256", "--channels", "16", "--steps", "2", "--warmup", "1",
           "--graph", graph]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                   # ONE line, from rank 0
    res = json.loads(lines[0])
    assert res["rehearsal"] is True and res["n_gpus"] == world and res["steps"] == 2 and res["warmup"] == 1
    assert res["metric"].startswith("edges/sec aggregated") and res["unit"] == "edges/s" and res["scaling"] == "strong"
    assert res["value"] > 0 and res["ms_per_step"] > 0 and res["roofline"] is None
    cfg = res["config"]
    assert "REHEARSAL" in cfg["workload"] and f"x{world}" in cfg["parallelism"]
    tuned = cfg["autotuned_ms_per_step"]
    assert {"allgather/node_groups=1", "halo/node_groups=1"} <= set(tuned)            # every applicable scheme ran ...
    assert all(isinstance(v, float) for v in tuned.values()), tuned                   # ... on every rank
    phase = cfg["phase_ms"]
    assert "error" not in phase and phase["step"] > 0 and phase["kernels"] > 0 and phase["local_edges"] > 0
    # whole-job value: all edges of the graph per step, not one rank's share
    n_edges = int(cfg["workload"].split(" E=")[1].split()[0])
    assert abs(res["value"] - n_edges * res["steps"] / (res["ms_per_step"] * 1e-3 * res["steps"])) <= 1e-6 * res["value"]
