"""GPU: the device graph builder (csrc/graph_build.hip) against the torch/scipy host constructions.

* `Graph` on CUDA tensors (dgcn_graph_csr_build: histogram + scan + 32-bit radix sort, one status read) must equal,
  array for array, the structure the torch composition builds from the same edge list on the CPU (stable order:
  duplicate edges, self loops, hubs that need work lists, already-sorted lists, empty rows, E = 0).
* `graph_prep.induced_subgraph` on CUDA (dgcn_subgraph_extract) must equal scipy's `adj[nodes, :][:, nodes]`
  slicing -- what the reference's utils/data_util.generate_sub_graphs does per cluster -- including the edge ids that
  select the edge_attr rows (examples/ogb/ogbn_proteins/dataset.py:139-140 does that lookup in a python dict).
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from deep_gcns_torch_amd import graph_prep as gp
from deep_gcns_torch_amd import synth
from deep_gcns_torch_amd.graph import Graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _same(a, b, what):
    if a is None or b is None:
        assert a is None and b is None, what
        return
    assert torch.equal(a.cpu(), b.cpu()), what


def _compare(ei, n_src, n_dst):
    gc = Graph(ei[0], ei[1], n_src, n_dst)
    gd = Graph(ei[0].to(DEV), ei[1].to(DEV), n_src, n_dst)
    for name in ("rowptr", "col", "eperm", "t_rowptr", "t_col", "t_eperm", "deg", "out_deg"):
        _same(getattr(gd, name), getattr(gc, name), name)
    assert (gd.work is None) == (gc.work is None)
    if gc.work is not None:
        assert gd.work[0] == gc.work[0] and gd.work[1] == gc.work[1]
        for a, b in zip(gd.work[2:], gc.work[2:]):
            _same(a, b, "work list")
    _same(gd.erow, gc.erow, "erow")


def test_csr_csc_match_the_host_construction():
    torch.manual_seed(0)
    _compare(synth.tricky_graph(), 257, 257)                                   # hub of 2100, duplicates, loops
    _compare(synth.powerlaw_graph(5000, 60_000, seed=3, exponent=2.1), 5000, 5000)
    g = torch.Generator().manual_seed(1)
    ei = torch.stack([torch.randint(0, 300, (4000,), generator=g), torch.randint(100, 200, (4000,), generator=g)])
    _compare(ei, 300, 300)                                                      # empty rows at both ends
    srt = ei[:, torch.sort(ei[1], stable=True).indices]
    _compare(srt, 300, 300)                                                     # already destination-sorted: eperm None
    _compare(torch.zeros(2, 0, dtype=torch.long), 7, 7)                         # no edges
    _compare(torch.tensor([[0, 1, 2], [3, 3, 1]]), 5, 5)
    rect = torch.stack([torch.randint(0, 40, (500,), generator=g), torch.randint(0, 9, (500,), generator=g)])
    _compare(rect, 40, 9)                                                       # rectangular (rank-local) graph


def test_out_of_range_ids_are_reported():
    bad = torch.tensor([[0, 1, 9], [1, 2, 0]], device=DEV)
    with pytest.raises(ValueError, match="out of range"):
        Graph(bad[0], bad[1], 5, 5)
    with pytest.raises(ValueError, match="out of range"):
        Graph(torch.tensor([0, -1], device=DEV), torch.tensor([1, 1], device=DEV), 5, 5)


def test_arxiv_sized_graph_builds_and_aggregates():
    s = synth.SHAPES["arxiv"]
    ei = synth.undirected_random_graph(s["n"], s["n_undirected"], s["seed"], device=DEV)
    g = Graph.from_edge_index(ei, s["n"])
    assert int(g.rowptr[-1]) == ei.size(1) and int(g.t_rowptr[-1]) == ei.size(1)
    # rows are runs of equal destination, in original edge order
    dst_sorted = ei[1][g.eperm.long()]
    assert bool((dst_sorted[1:] >= dst_sorted[:-1]).all())
    same = dst_sorted[1:] == dst_sorted[:-1]
    assert bool((g.eperm[1:][same] > g.eperm[:-1][same]).all())
    assert torch.equal(g.col.long(), ei[0][g.eperm.long()])
    assert torch.equal(g.erow.long(), dst_sorted)


@pytest.mark.parametrize("clusters", [6, 10])
def test_induced_subgraph_matches_scipy_slicing(clusters):
    n = 3000
    ei = gp.to_undirected(torch.randint(0, n, (2, 40_000), generator=torch.Generator().manual_seed(5)), n)
    np.random.seed(7)
    parts = np.random.randint(clusters, size=n)
    E = ei.size(1)
    edge_attr = torch.arange(E, dtype=torch.float32).unsqueeze(1).repeat(1, 3)
    adj = sp.csr_matrix((np.arange(1, E + 1), (ei[0].numpy(), ei[1].numpy())), shape=(n, n))   # value = edge id + 1
    eid, pd = ei.to(DEV), torch.from_numpy(parts).to(DEV)
    for c in range(clusters):
        nodes = np.where(parts == c)[0]
        coo = adj[nodes, :][:, nodes].tocoo()
        nd, sub, attr, eids = gp.induced_subgraph(eid, pd, c, n, edge_attr=edge_attr.to(DEV))
        assert np.array_equal(nd.cpu().numpy(), nodes)
        # scipy returns the slice row-major; the device result keeps the ORIGINAL edge order: compare as sorted triples
        mine = sorted(zip(sub[0].tolist(), sub[1].tolist(), eids.tolist()))
        ref = sorted(zip(coo.row.tolist(), coo.col.tolist(), (coo.data - 1).tolist()))
        assert mine == ref
        assert bool((eids[1:] > eids[:-1]).all())                               # original order preserved
        assert torch.equal(attr.cpu(), edge_attr[eids.cpu()])
        assert torch.equal(nd[sub], eid[:, eids])
    # the CPU composition gives the same answer
    nd_c, sub_c, _, eids_c = gp.induced_subgraph(ei, torch.from_numpy(parts), 2, n)
    nd_d, sub_d, _, eids_d = gp.induced_subgraph(eid, pd, 2, n)
    assert torch.equal(nd_c, nd_d.cpu()) and torch.equal(sub_c, sub_d.cpu()) and torch.equal(eids_c, eids_d.cpu())


def test_coalesce_and_to_undirected_match_the_host_composition():
    """dgcn_graph_coalesce (radix sort of compact row * 2^b + col keys + first-of-run compaction) against the torch.unique
    composition on the CPU: PyG to_undirected / torch_sparse.coalesce semantics (examples/ogb/ogbn_arxiv/main.py:72-75)."""
    g = torch.Generator().manual_seed(3)
    for n, e in ((50, 400), (3000, 40_000), (100_000, 300_000), (7, 0), (1, 5)):
        ei = torch.randint(0, n, (2, e), generator=g)
        for fn in (gp.coalesce, gp.to_undirected):
            want = fn(ei, n)
            got = fn(ei.to(DEV), n)
            assert got.dtype == torch.long and torch.equal(got.cpu(), want), (fn.__name__, n, e)
    s = synth.SHAPES["arxiv"]
    raw = torch.randint(0, s["n"], (2, s["n_undirected"]), generator=g)
    und = gp.to_undirected(raw.to(DEV), s["n"])
    key = und[0] * s["n"] + und[1]
    assert bool((key[1:] > key[:-1]).all())                                      # sorted by (row, col), no duplicates
    assert torch.equal(torch.sort(und[1] * s["n"] + und[0]).values, key)          # symmetric
    with pytest.raises(ValueError, match="out of range"):
        gp.coalesce(torch.tensor([[0, 9], [1, 2]], device=DEV), 5)


def test_subgraph_extract_rejects_out_of_range_endpoints():
    parts = torch.zeros(5, dtype=torch.long, device=DEV)
    with pytest.raises(ValueError, match="out of range"):
        gp.induced_subgraph(torch.tensor([[0, 1, 7], [1, 2, 0]], device=DEV), parts, 0, 5)
    with pytest.raises(ValueError, match="out of range"):
        gp.induced_subgraph(torch.tensor([[0, -1], [1, 2]], device=DEV), parts, 0, 5)


def test_hub_work_lists_on_power_law_graphs():
    """The device work list (dgcn_graph_work_list) on the graphs it exists for: the ogbn-proteins cluster shape (hubs of
    thousands of in-edges) -- array for array the host construction, and consistent with rowptr."""
    s = synth.SHAPES["proteins_cluster"]
    ei = synth.powerlaw_graph(s["n"], s["n_undirected"], s["seed"])
    _compare(ei, s["n"], s["n"])
    gd = Graph(ei[0].to(DEV), ei[1].to(DEV), s["n"], s["n"])
    n_work, n_slots, row, beg, end, slot, split = gd.work
    assert n_work == row.numel() and int((slot >= 0).sum()) == n_slots
    assert bool((end > beg).all()) and bool((end - beg <= 2 * 256).all())
    rp = gd.rowptr.long()
    assert torch.equal(beg[split.long()].long(), rp[row[split.long()].long()])    # first item of a split row starts the row
