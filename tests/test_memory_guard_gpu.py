"""Memory safety of the library's kernels under a guard-page allocator.

``tests/guard_alloc/guard_alloc.cpp`` replaces torch's caching allocator for a child process: every allocation -- inputs,
outputs, saved-for-backward arrays, workspaces -- sits in its own HIP virtual-memory mapping that ENDS where the buffer
ends, with unmapped pages on both sides, and is unmapped when freed.  A kernel that reads or writes one element past any
buffer it was handed (or touches a freed one) takes a GPU page fault and the child dies; ``tests/guard_alloc/run.py`` then
names the buffer.  The children run parts of this very suite: the sparse aggregation kernels (row walk, per-edge encoder
walk incl. hub merge kernels, arg-max winner kernels, fused edge GEMM), the reversible RevGCN on the golden cases, the
node-wise kernels, the graph builder, and a captured + replayed training step (allocations of a capture stay mapped, as
torch's graph pools keep them)."""
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("identity_dropout_mask")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = os.path.join(ROOT, "tests", "guard_alloc", "run.py")


def _guarded(args, timeout, no_blocking=False):
    if not os.path.exists(os.path.join(ROOT, "tests", "guard_alloc", "libguard_alloc.so")):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__._build_guard_allocator()
    cmd = [sys.executable, RUN, "--timeout", str(timeout), "--log", f"/tmp/dgcn_guard_{os.getpid()}.log"]
    if no_blocking:
        cmd.append("--no-blocking")
    r = subprocess.run(cmd + ["--"] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout + 60)
    tail = (r.stdout or "")[-3000:] + (r.stderr or "")[-1000:]
    assert r.returncode == 0, f"guarded run failed (rc {r.returncode}):\n{tail}"
    return r.stdout


def test_guard_allocator_selftest():
    """The harness itself: torch ops and the device graph builder give the host's answers in a guarded process."""
    out = _guarded([sys.executable, os.path.join(ROOT, "tests", "guard_alloc", "selftest.py")], 300)
    assert "selftest ok" in out


@pytest.mark.parametrize("target,select", [
    ("tests/test_gen_aggr_gpu.py", "not products_shape and not arxiv_shape"),
    ("tests/test_egemm_gpu.py", "not cluster_shape"),
    ("tests/test_revgcn.py", ""),
    ("tests/test_node_ops_gpu.py", "layer or batch"),
    ("tests/test_graph_build_gpu.py", ""),
])
def test_kernels_stay_inside_their_buffers(target, select):
    args = [sys.executable, "-m", "pytest", target, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"]
    if select:
        args += ["-k", select]
    out = _guarded(args, 900)
    assert " passed" in out and " failed" not in out


def test_captured_and_replayed_step_under_the_guard():
    """Eager rows, then the same model captured and replayed (the sequence in which bench.py faulted in round 4)."""
    out = _guarded([sys.executable, os.path.join(ROOT, "tests", "guard_alloc", "revgcn_sequence.py"), "--winner", "1",
                    "--rows", "composed,composed_graph", "--steps", "1", "--replays", "2"], 900, no_blocking=True)
    assert "sequence ok" in out
