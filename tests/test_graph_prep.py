"""CPU: device-agnostic graph preprocessing helpers against the restated PyG semantics and scipy slicing."""
import numpy as np
import scipy.sparse as sp
import torch

from deep_gcns_torch_amd import graph_prep as gp
from deep_gcns_torch_amd.utils import data_util
from oracle import thirdparty as tp


def _rand_graph(n, e, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, n, (2, e), generator=g)


def test_to_undirected_and_self_loops():
    ei = _rand_graph(50, 400, 0)
    und = gp.to_undirected(ei, 50)
    pairs = set(map(tuple, ei.t().tolist())) | set(map(tuple, ei.flip(0).t().tolist()))
    assert set(map(tuple, und.t().tolist())) == pairs and und.size(1) == len(pairs)
    key = und[0] * 50 + und[1]
    assert bool((key[1:] > key[:-1]).all())                       # sorted by (row, col), no duplicates
    with_loops = gp.add_self_loops(und, 50)
    ref, _ = tp.add_self_loops(und, num_nodes=50)
    assert torch.equal(with_loops, ref)
    no_loops, _ = gp.remove_self_loops(with_loops)
    assert bool((no_loops[0] != no_loops[1]).all())


def test_induced_subgraphs_match_scipy_slicing():
    n, clusters = 300, 6
    ei = gp.to_undirected(_rand_graph(n, 3000, 1), n)
    np.random.seed(3)
    parts_np = data_util.random_partition_graph(n, clusters)
    adj = sp.csr_matrix((np.ones(ei.size(1)), (ei[0].numpy(), ei[1].numpy())), shape=(n, n))
    ref_nodes, ref_edges = data_util.generate_sub_graphs(adj, parts_np, clusters, batch_size=2)
    nodes, edges = gp.generate_sub_graphs(ei, torch.from_numpy(parts_np), n, clusters, batch_size=2)
    assert len(nodes) == len(ref_nodes) == 3
    for c, (a, b, ea, eb) in enumerate(zip(nodes, ref_nodes, edges, ref_edges)):
        assert np.array_equal(b, np.where(parts_np == c)[0])        # the reference's selection: parts == c
        assert np.array_equal(a.numpy(), b)
        assert torch.equal(ea, eb)                                  # same relabelling, same (row, col) order

    class _SparseTensorLike:                                        # what the reference's callers pass (torch_sparse)
        def to_scipy(self, layout="csr"):
            assert layout == "csr"
            return adj
    n2, e2 = data_util.generate_sub_graphs(_SparseTensorLike(), parts_np, clusters, batch_size=1)
    assert len(n2) == clusters and all(torch.equal(x, y) for x, y in zip(e2[:3], ref_edges))
    nd, sub, attr, eids = gp.induced_subgraph(ei, torch.from_numpy(parts_np), 4, n, edge_attr=torch.arange(ei.size(1)).float())
    assert torch.equal(attr, eids.float()) and bool((sub >= 0).all()) and int(sub.max()) < nd.numel()
    assert torch.equal(nd[sub], ei[:, eids])
