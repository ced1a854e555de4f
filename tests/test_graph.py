"""CPU: host logic of the graph-structure builder (CSR/CSC, stable order, hub work lists)."""
import torch

from deep_gcns_torch_amd import synth
from deep_gcns_torch_amd.graph import Graph, graph_of


def _check_csr(rowptr, col, eperm, keys, vals, n):
    E = keys.numel()
    assert rowptr[0] == 0 and rowptr[-1] == E and rowptr.numel() == n + 1
    perm = torch.arange(E) if eperm is None else eperm.long()
    for r in range(n):
        seg = perm[rowptr[r]:rowptr[r + 1]]
        assert torch.all(keys[seg] == r)
        assert torch.all(seg[1:] > seg[:-1]), "stable order violated"
    assert torch.equal(col.long(), vals[perm])


def test_csr_csc_of_tricky_graph():
    ei = synth.tricky_graph()
    g = Graph.from_edge_index(ei, 257)
    _check_csr(g.rowptr, g.col, g.eperm, ei[1], ei[0], 257)
    _check_csr(g.t_rowptr, g.t_col, g.t_eperm, ei[0], ei[1], 257)
    assert torch.equal(g.deg, torch.bincount(ei[1], minlength=257).float())
    assert g.deg[:16].sum() >= 0 and int(g.deg[5]) >= 2048


def test_sorted_input_needs_no_permutation():
    ei = synth.tricky_graph()
    order = torch.sort(ei[1], stable=True).indices
    g = Graph.from_edge_index(ei[:, order].contiguous(), 257)
    assert g.eperm is None


def test_work_list_splits_hubs_only():
    ei = synth.tricky_graph()
    g = Graph.from_edge_index(ei, 257, hub_chunk=256)
    n_work, n_slots, row, beg, end, slot, split_first = g.work
    assert split_first.tolist() == [i for i in range(n_work) if int(slot[i]) >= 0 and (i == 0 or int(row[i - 1]) != int(row[i]))]
    deg = (g.rowptr[1:] - g.rowptr[:-1]).long()
    covered = torch.zeros(257, dtype=torch.long)
    for i in range(n_work):
        r = int(row[i])
        assert int(g.rowptr[r]) <= int(beg[i]) <= int(end[i]) <= int(g.rowptr[r + 1])
        covered[r] += int(end[i] - beg[i])
        if deg[r] > 512:
            assert int(slot[i]) >= 0 and int(end[i] - beg[i]) <= 256
        else:
            assert int(slot[i]) == -1 and int(end[i] - beg[i]) == int(deg[r])
    assert torch.equal(covered, deg)
    used = slot[slot >= 0]
    assert torch.equal(torch.sort(used).values, torch.arange(n_slots, dtype=torch.int32))
    # items of one row are contiguous
    assert torch.all(row[1:] >= row[:-1])
    # a graph without hubs gets no work list
    small = torch.randint(0, 50, (2, 300), generator=torch.Generator().manual_seed(1))
    assert Graph.from_edge_index(small, 50).work is None


def test_cache_is_keyed_by_tensor_object_and_version():
    ei = torch.randint(0, 20, (2, 64), generator=torch.Generator().manual_seed(2))
    g1 = graph_of(ei, 20)
    assert graph_of(ei, 20) is g1
    ei[0, 0] = (ei[0, 0] + 1) % 20          # in-place edit bumps _version -> rebuild
    g2 = graph_of(ei, 20)
    assert g2 is not g1
    clone = ei.clone()
    assert graph_of(clone, 20) is not g2     # different object, different storage -> different entry
    assert graph_of(g2) is g2
    # an alias of a LIVE cached tensor (what re-entrant checkpointing passes: edge_index.detach()) hits
    assert graph_of(ei.detach(), 20) is g2
    assert graph_of(ei.view(2, -1), 20) is g2
    assert graph_of(ei[:, :32], 20) is not g2            # different shape: not an alias of the whole list
    # once the owner dies a recycled allocation must not resurrect the entry
    import gc
    from deep_gcns_torch_amd import graph as G
    n_before = len(G._cache)
    del clone
    gc.collect()
    assert len(G._cache) == n_before - 1


def test_empty_and_rectangular():
    g = Graph(torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long), 5, 3)
    assert g.rowptr.tolist() == [0, 0, 0, 0] and g.t_rowptr.tolist() == [0] * 6
    src = torch.tensor([4, 0, 4, 2])
    dst = torch.tensor([1, 1, 0, 2])
    g = Graph(src, dst, 5, 3)
    assert g.rowptr.tolist() == [0, 1, 3, 4] and g.col.tolist() == [4, 4, 0, 2]
    assert g.t_rowptr.tolist() == [0, 1, 1, 2, 2, 4] and g.t_col.tolist() == [1, 2, 1, 0]


def test_locality_ordered_synthetic_graph_keeps_partitions_mostly_local():
    """bench.py --graph local: symmetric, in range, and a destination partition references few remote rows
    (the uniform graph references nearly all of them)."""
    from deep_gcns_torch_amd import synth
    n = 20000
    ei = synth.local_graph(n, 200_000, seed=3)
    assert ei.shape == (2, 2 * 200_000 + n) and int(ei.min()) >= 0 and int(ei.max()) < n
    fwd = set(map(tuple, ei.t()[:1000].tolist()))
    rev = set(map(tuple, ei.flip(0).t().tolist()))
    assert fwd <= rev                                                   # every edge has its reverse
    lo, hi = n // 4, n // 2

    def remote_rows(e):
        s = e[0][(e[1] >= lo) & (e[1] < hi)]
        return torch.unique(s[(s < lo) | (s >= hi)]).numel()

    uniform = synth.undirected_random_graph(n, 200_000, seed=3)
    assert remote_rows(ei) < 0.5 * (hi - lo)
    assert remote_rows(uniform) > 2 * (hi - lo)


def test_exchange_bytes_and_choose_scheme_follow_the_graphs_locality():
    """dist.exchange_bytes counts what each multi-GPU scheme makes a rank receive from the edge list alone (no process
    group); dist.choose_scheme turns the counts into the scheme build_partition("auto") takes: the halo exchange on a
    locality-ordered graph, the channel-transposed scheme on a graph whose partitions reference every row, the
    local-first / all-gather pair where the channel count rules the transposed scheme out."""
    import torch
    from deep_gcns_torch_amd import dist as ddist, synth
    n, C, W = 20_000, 128, 8
    local = synth.local_graph(n, 150_000, seed=1)
    uniform = synth.undirected_random_graph(n, 150_000, seed=1)
    for r in (0, 3, 7):
        cl = ddist.exchange_bytes(local, n, C, r, W)
        cu = ddist.exchange_bytes(uniform, n, C, r, W)
        assert cl["edges"] == cl["local_source_edges"] + cl["remote_source_edges"]
        assert cl["halo"] == cl["halo_rows"] * C * 4 and cl["allgather"] == cl["split"]
        assert cl["halo_rows"] < 0.5 * (n - cl["rows"]) <= cu["halo_rows"] / 0.9       # few remote rows vs nearly all
        assert cl["local_source_edges"] > 0.8 * cl["edges"] and cu["local_source_edges"] < 0.25 * cu["edges"]
        assert ddist.choose_scheme(cl, "softmax_sg", {"t": 0.1}) == "halo"
        assert ddist.choose_scheme(cu, "softmax_sg", {"t": 0.1}) == "transposed"
    # 100 channels at W = 8: no channel split -> the all-gather volume; as the local-first scheme only where enough edges
    # have a local source and the aggregator has an associative partial state
    c100 = ddist.exchange_bytes(uniform, n, 100, 0, W)
    assert "transposed" not in c100 and ddist.choose_scheme(c100, "softmax_sg", {"t": 0.1}) == "allgather"
    half = dict(c100, local_source_edges=c100["edges"] // 2)
    assert ddist.choose_scheme(half, "max", {}) == "split" and ddist.choose_scheme(half, "power", {"p": 2.0}) == "split"
    assert ddist.choose_scheme(half, "softmax", {"learn_t": True}) == "allgather"
    assert ddist.choose_scheme(half, "power", {"p": 2.0, "learn_p": True}) == "allgather"
