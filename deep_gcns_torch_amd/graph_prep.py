"""Device-side graph preprocessing.

The reference prepares graphs on the CPU every epoch: PyG ``to_undirected`` / ``add_self_loops``
(examples/ogb/ogbn_arxiv/main.py:72-75), a random node partition and scipy CSR slicing per cluster
(utils/data_util.py:43-61, examples/ogb/ogbn_products/main.py:120-126; ogbn_proteins/dataset.py:87-151 additionally
looks every edge up in a python dict to recover its edge_attr row), then an H2D copy per cluster.  Once the
aggregation runs at HBM speed that host work is the epoch bottleneck.  Here the same integer work runs on the
device: the induced sub-graph of a cluster (nodes, relabelled edges in original order, kept edge ids for
``edge_attr``) is ONE libdgcn call (``dgcn_subgraph_extract``: flag -> scan -> compact, csrc/graph_build.hip) with a
single host read of the two counts; ``to_undirected`` / coalescing use the device sort.  Everything returns COO
``edge_index`` tensors with the reference's conventions, ready for the modules (whose CSR/CSC is built by
``dgcn_graph_csr_build`` and cached by ``graph.graph_of``).  CPU tensors take the torch compositions (host-logic
tests).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


def _coalesce_device(edge_index: torch.Tensor, num_nodes: int, both: bool) -> torch.Tensor:
    """dgcn_graph_coalesce (csrc/graph_build.hip): one radix sort of compact row * 2^b + col keys, first-of-run flags,
    scan, compaction; one host read of (count, bad-id flag)."""
    from . import _lib
    lib = _lib.load()
    dev = edge_index.device
    ei = edge_index.long().contiguous()
    E = ei.size(1)
    n_keys = E * (2 if both else 1)
    out = torch.empty(2, max(n_keys, 1), device=dev, dtype=torch.long)
    counts = torch.empty(2, device=dev, dtype=torch.long)
    ws_bytes = lib.dgcn_graph_coalesce_workspace_bytes(E, num_nodes, 1 if both else 0)
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    with _lib.device_ctx(dev):
        rc = lib.dgcn_graph_coalesce(ei[0].data_ptr(), ei[1].data_ptr(), E, num_nodes, 1 if both else 0, out.data_ptr(),
                                     out.stride(0), counts.data_ptr(), ws.data_ptr(), ws_bytes,
                                     _lib.current_stream_handle(dev))
    _lib.check(rc, "dgcn_graph_coalesce")
    n_out, bad = counts.tolist()                          # the one host read
    if bad:
        raise ValueError("edge_index out of range")
    return out[:, :n_out].clone() if n_out * 2 < n_keys else out[:, :n_out]


def coalesce(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """Sort by (row, col) and drop duplicate edges (torch_sparse.coalesce on indices only)."""
    if edge_index.is_cuda and num_nodes > 0:
        return _coalesce_device(edge_index, num_nodes, both=False)
    key = edge_index[0] * num_nodes + edge_index[1]
    key = torch.unique(key, sorted=True)
    return torch.stack([key // num_nodes, key % num_nodes])


def to_undirected(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """Both directions of every edge, coalesced (sorted by source, duplicates removed): PyG ``to_undirected``
    (examples/ogb/ogbn_arxiv/main.py:72-75)."""
    if edge_index.is_cuda and num_nodes > 0:
        return _coalesce_device(edge_index, num_nodes, both=True)
    both = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    return coalesce(both, num_nodes)


def add_self_loops(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """Append one (i, i) edge per node at the END of the list (PyG ``add_self_loops`` keeps existing loops)."""
    loop = torch.arange(num_nodes, device=edge_index.device, dtype=edge_index.dtype)
    return torch.cat([edge_index, loop.unsqueeze(0).repeat(2, 1)], dim=1)


def remove_self_loops(edge_index: torch.Tensor, edge_attr: Optional[torch.Tensor] = None):
    keep = edge_index[0] != edge_index[1]
    return edge_index[:, keep], (None if edge_attr is None else edge_attr[keep])


def random_partition(num_nodes: int, cluster_number: int, generator: Optional[torch.Generator] = None,
                     device="cpu") -> torch.Tensor:
    """Uniform random cluster id per node (utils/data_util.py:43-45), generated on ``device``."""
    return torch.randint(0, cluster_number, (num_nodes,), generator=generator, device=device)


def induced_subgraph(edge_index: torch.Tensor, parts: torch.Tensor, cluster: int, num_nodes: int,
                     edge_attr: Optional[torch.Tensor] = None
                     ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], torch.Tensor]:
    """Sub-graph induced by the nodes with ``parts == cluster`` (what ``adj[nodes, :][:, nodes]`` does on the
    host, utils/data_util.py:55-60): returns (node ids ascending, relabelled edge_index in the original edge
    order, the matching rows of ``edge_attr``, ids of the kept edges)."""
    if edge_index.is_cuda:
        nodes, sub, eids = _induced_subgraph_device(edge_index, parts, cluster, num_nodes)
    else:
        mask = parts == cluster
        nodes = torch.nonzero(mask).flatten()
        new_id = torch.full((num_nodes,), -1, dtype=edge_index.dtype, device=edge_index.device)
        new_id[nodes] = torch.arange(nodes.numel(), device=edge_index.device, dtype=edge_index.dtype)
        keep = mask[edge_index[0]] & mask[edge_index[1]]
        eids = torch.nonzero(keep).flatten()
        sub = new_id[edge_index[:, eids]]
    return nodes, sub, (None if edge_attr is None else edge_attr[eids]), eids


def _induced_subgraph_device(edge_index, parts, cluster, num_nodes):
    from . import _lib
    lib = _lib.load()
    dev = edge_index.device
    ei = edge_index.long().contiguous()
    pt = parts.to(device=dev, dtype=torch.long).contiguous()
    E = ei.size(1)
    i64 = dict(device=dev, dtype=torch.long)
    nodes = torch.empty(num_nodes, **i64)
    sub = torch.empty(2, E, **i64)
    eids = torch.empty(E, **i64)
    counts = torch.empty(3, **i64)
    ws_bytes = lib.dgcn_subgraph_workspace_bytes(E, num_nodes)
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    with _lib.device_ctx(dev):
        rc = lib.dgcn_subgraph_extract(ei[0].data_ptr(), ei[1].data_ptr(), E, pt.data_ptr(), num_nodes, int(cluster),
                                       nodes.data_ptr(), sub[0].data_ptr(), sub[1].data_ptr(), eids.data_ptr(),
                                       counts.data_ptr(), ws.data_ptr(), ws_bytes, _lib.current_stream_handle(dev))
    _lib.check(rc, "dgcn_subgraph_extract")
    n_sub, e_sub, bad = counts.tolist()                  # the one host read
    if bad:
        raise ValueError("edge_index out of range")
    # trimmed COPIES: a caller that keeps one sub-graph per cluster must not pin the full-size (E) scratch arrays
    return nodes[:n_sub].clone(), sub[:, :e_sub].clone(), eids[:e_sub].clone()


def generate_sub_graphs(edge_index: torch.Tensor, parts: torch.Tensor, num_nodes: int, cluster_number: int = 10,
                        batch_size: int = 1) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """Device-side equivalent of utils/data_util.generate_sub_graphs (reference utils/data_util.py:48-61): batch
    ``c`` = the sub-graph induced by ``parts == c`` for c in range(cluster_number // batch_size), exactly the
    reference's selection.  Edge lists are sorted by (row, col) like the scipy CSR -> COO conversion."""
    sg_nodes, sg_edges = [], []
    for c in range(cluster_number // batch_size):
        group = parts == c
        nodes = torch.nonzero(group).flatten()
        new_id = torch.full((num_nodes,), -1, dtype=edge_index.dtype, device=edge_index.device)
        new_id[nodes] = torch.arange(nodes.numel(), device=edge_index.device, dtype=edge_index.dtype)
        keep = group[edge_index[0]] & group[edge_index[1]]
        sub = new_id[edge_index[:, keep]]
        sg_nodes.append(nodes)
        sg_edges.append(coalesce(sub, max(int(nodes.numel()), 1)) if sub.numel() else sub)
    return sg_nodes, sg_edges
