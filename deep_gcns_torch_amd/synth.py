"""Seeded synthetic graphs / point clouds of the shapes named in BASELINE.md (no datasets
offline).  Pure index generation with torch; used by tests, bench.py and the golden script."""
from __future__ import annotations

import torch


def tricky_graph(n: int = 257, e: int = 4099, hub_deg: int = 2100, seed: int = 0) -> torch.Tensor:
    """Small adversarial edge list: isolated destinations, duplicate edges,
    self-loops, one hub destination with >= 2048 in-edges, one hub source, unsorted order."""
    g = torch.Generator().manual_seed(seed)
    assert e > hub_deg + 200 and n > 40
    hub_dst, hub_src = 5, 7
    n_rand = e - hub_deg - 150
    src = torch.randint(0, n, (n_rand,), generator=g)
    dst = torch.randint(16, n, (n_rand,), generator=g)         # nodes 0..15 get no random in-edges
    hub = torch.stack([torch.randint(0, n, (hub_deg,), generator=g), torch.full((hub_deg,), hub_dst)])
    out_hub = torch.stack([torch.full((100,), hub_src), torch.randint(16, n, (100,), generator=g)])
    loops = torch.arange(20, 45).repeat(2, 1)                   # 25 self-loops
    dup = torch.stack([src[:25], dst[:25]])                     # 25 duplicated edges
    ei = torch.cat([torch.stack([src, dst]), hub, out_hub, loops, dup], dim=1)
    assert ei.size(1) == e
    perm = torch.randperm(e, generator=g)
    return ei[:, perm].contiguous()


def undirected_random_graph(n: int, n_undirected: int, seed: int, device="cpu",
                            self_loops: bool = True) -> torch.Tensor:
    """Uniform random graph, symmetrised (both directions kept, duplicates allowed -- they are
    rare and legal) plus one self-loop per node appended at the end, as the reference prepares
    ogbn-arxiv (examples/ogb/ogbn_arxiv/main.py:72-75: to_undirected + add_self_loops)."""
    g = torch.Generator(device=device).manual_seed(seed)
    a = torch.randint(0, n, (n_undirected,), generator=g, device=device)
    b = torch.randint(0, n, (n_undirected,), generator=g, device=device)
    src = torch.cat([a, b])
    dst = torch.cat([b, a])
    if self_loops:
        loop = torch.arange(n, device=device)
        src = torch.cat([src, loop])
        dst = torch.cat([dst, loop])
    return torch.stack([src, dst])


def powerlaw_graph(n: int, n_undirected: int, seed: int, exponent: float = 2.5, device="cpu",
                   self_loops: bool = True) -> torch.Tensor:
    """Chung-Lu style graph: endpoints drawn proportionally to a power-law weight sequence."""
    g = torch.Generator(device=device).manual_seed(seed)
    ranks = torch.arange(1, n + 1, device=device, dtype=torch.float64)
    w = ranks.pow(-1.0 / (exponent - 1.0))
    cdf = torch.cumsum(w / w.sum(), 0)
    u = torch.rand(2 * n_undirected, generator=g, device=device, dtype=torch.float64)
    ends = torch.searchsorted(cdf, u).clamp_(max=n - 1)
    scramble = torch.randperm(n, generator=g, device=device)   # hubs are not the low ids
    ends = scramble[ends]
    a, b = ends[:n_undirected], ends[n_undirected:]
    src = torch.cat([a, b])
    dst = torch.cat([b, a])
    if self_loops:
        loop = torch.arange(n, device=device)
        src = torch.cat([src, loop])
        dst = torch.cat([dst, loop])
    return torch.stack([src, dst])


def local_graph(n: int, n_undirected: int, seed: int, window: int = 0, far_fraction: float = 0.02, device="cpu",
                self_loops: bool = True) -> torch.Tensor:
    """Locality-ordered graph: the node ids are a good ordering (what METIS / RCM give a real graph such as
    ogbn-products, whose co-purchase neighbourhoods are tight): an edge joins node a to a node within ``window`` ids
    of it, except a small ``far_fraction`` of uniformly random long-range edges.  Destination partitions of such a graph
    reference few remote source rows, which is the regime the halo exchange scheme (dist.HaloGraph) is built for; the
    uniform graph above is the opposite extreme (every partition references every row)."""
    g = torch.Generator(device=device).manual_seed(seed)
    if window <= 0:
        window = max(64, n // 256)
    a = torch.randint(0, n, (n_undirected,), generator=g, device=device)
    off = torch.randint(1, window + 1, (n_undirected,), generator=g, device=device)
    sign = torch.randint(0, 2, (n_undirected,), generator=g, device=device) * 2 - 1
    b = (a + sign * off).clamp_(0, n - 1)
    far = torch.rand(n_undirected, generator=g, device=device) < far_fraction
    b = torch.where(far, torch.randint(0, n, (n_undirected,), generator=g, device=device), b)
    src = torch.cat([a, b])
    dst = torch.cat([b, a])
    if self_loops:
        loop = torch.arange(n, device=device)
        src = torch.cat([src, loop])
        dst = torch.cat([dst, loop])
    return torch.stack([src, dst])


# named shapes
SHAPES = {
    "arxiv": dict(n=169_343, n_undirected=1_157_799, channels=128, seed=3),       # E = 2,484,941
    "products": dict(n=2_449_029, n_undirected=61_859_140, channels=128, seed=4),  # E = 126,167,309
    "proteins_cluster": dict(n=13_253, n_undirected=388_986, channels=64, seed=5),  # E = 791,225
    "ppi": dict(n=2_245, e=61_318, channels=50, seed=1),
}


def lattice_cloud(B: int, C: int, N: int, seed: int, span: int = 0) -> torch.Tensor:
    """Point cloud (B,C,N,1) whose coordinates are small integers * 2^-11: every product and
    partial sum of the squared-distance computation is an exactly representable fp32 value
    under ANY summation order (needs 2*C*span^2 <= 2^24), so the kNN index comparison is not at
    the mercy of the GEMM reduction order (SURVEY.md §7 hard-part 1).  Exact ties remain
    possible; tests compare tied ranks by distance value."""
    if span <= 0:
        span = 1
        while 2 * C * (2 * span) ** 2 <= (1 << 24):
            span *= 2
    assert 2 * C * span * span <= (1 << 24), "distances would not be exact in fp32"
    g = torch.Generator().manual_seed(seed)
    q = torch.randint(0, span, (B, C, N, 1), generator=g)
    return q.to(torch.float32) * (2.0 ** -11)
