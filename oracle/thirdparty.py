"""ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU restatement (plain PyTorch ops) of the third-party primitives the reference's sparse
hot path calls but which are NOT vendored in /root/reference and NOT installed here:

  torch-scatter   (unpinned in deepgcn_env_install.sh:27; 2.0.7-2.0.9 match torch 1.9)
      scatter / scatter_sum / scatter_mean / scatter_max / scatter_min / scatter_softmax
      call sites: gcn_lib/sparse/torch_message.py:52,55,57,71 ; utils/pyg_util.py:26-27
  torch-geometric (unpinned, README says >=1.6.0; 1.7.x matches torch 1.9)
      MessagePassing.propagate (flow source_to_target), utils.degree,
      utils.add_self_loops/remove_self_loops, nn.EdgeConv
      call sites: gcn_lib/sparse/torch_message.py:3,5,8,47,62 ; torch_vertex.py:4,9,68,106-114
  torch-cluster   (unpinned; 1.5.9)   knn_graph
      call sites: gcn_lib/dense/torch_edge.py:3,97 ; gcn_lib/sparse/torch_edge.py:3,46

Their published algorithms are restated below; because the packages are absent, the
third-party layer itself is "parity unpinned" (see DESIGN.md).  Everything above that layer
is pinned by running the reference's OWN files on top of these functions
(oracle/make_golden.py) and by the hand-computed known answers in tests/.
"""
from __future__ import annotations

import inspect

import torch


def _broadcast(index: torch.Tensor, src: torch.Tensor, dim: int) -> torch.Tensor:
    """torch_scatter.utils.broadcast: expand a 1-D index along `dim` to src's shape."""
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() == 1:
        for _ in range(dim):
            index = index.unsqueeze(0)
    for _ in range(index.dim(), src.dim()):
        index = index.unsqueeze(-1)
    return index.expand(src.size())


def _out_size(src, index, dim, dim_size):
    size = list(src.size())
    if dim_size is not None:
        size[dim] = dim_size
    elif index.numel() == 0:
        size[dim] = 0
    else:
        size[dim] = int(index.max()) + 1
    return size


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    index = _broadcast(index, src, dim)
    if out is None:
        out = torch.zeros(_out_size(src, index, dim, dim_size), dtype=src.dtype, device=src.device)
    return out.scatter_add_(dim, index, src)


scatter_add = scatter_sum


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    out = scatter_sum(src, index, dim, out, dim_size)
    dim_size = out.size(dim)
    index_dim = dim
    if index_dim < 0:
        index_dim = index_dim + src.dim()
    if index.dim() <= index_dim:
        index_dim = index.dim() - 1
    ones = torch.ones(index.size(), dtype=src.dtype, device=src.device)
    count = scatter_sum(ones, index, index_dim, None, dim_size)
    count.clamp_(1)
    count = _broadcast(count, out, dim)
    out.div_(count)
    return out


def _scatter_extreme(src, index, dim, dim_size, mode):
    """scatter_max / scatter_min: (values, arg).  Segments that receive nothing yield 0 and
    arg = src.size(dim); among equal maxima the FIRST edge wins (CPU kernel loops in order
    with a strict comparison)."""
    if dim < 0:
        dim = src.dim() + dim
    idx = _broadcast(index, src, dim)
    size = _out_size(src, idx, dim, dim_size)
    n = src.size(dim)
    with torch.no_grad():
        ext = torch.zeros(size, dtype=src.dtype, device=src.device)
        ext.scatter_reduce_(dim, idx, src.detach(), "amax" if mode == "max" else "amin", include_self=False)
        # arg = first position attaining the extreme
        pos_shape = [1] * src.dim()
        pos_shape[dim] = n
        pos = torch.arange(n, device=src.device).view(pos_shape).expand(src.size())
        hit = src.detach() == ext.gather(dim, idx)
        cand = torch.where(hit, pos, torch.full_like(pos, n))
        arg = torch.full(size, n, dtype=torch.long, device=src.device)
        arg.scatter_reduce_(dim, idx, cand, "amin", include_self=True)
    # values through a gather so that autograd sends the gradient to the arg edge only
    # (torch_scatter's backward), not split among ties
    if n == 0:
        return torch.zeros(size, dtype=src.dtype, device=src.device), arg
    picked = src.gather(dim, arg.clamp(max=n - 1))
    out = torch.where(arg < n, picked, torch.zeros_like(picked))
    return out, arg


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    return _scatter_extreme(src, index, dim, dim_size, "max")


def scatter_min(src, index, dim=-1, out=None, dim_size=None):
    return _scatter_extreme(src, index, dim, dim_size, "min")


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    if reduce in ("sum", "add"):
        return scatter_sum(src, index, dim, out, dim_size)
    if reduce == "mean":
        return scatter_mean(src, index, dim, out, dim_size)
    if reduce == "max":
        return scatter_max(src, index, dim, out, dim_size)[0]
    if reduce == "min":
        return scatter_min(src, index, dim, out, dim_size)[0]
    raise ValueError(reduce)


def scatter_softmax(src, index, dim=-1, dim_size=None):
    """torch_scatter.composite.scatter_softmax (>=2.0.6: no epsilon in the denominator)."""
    if not torch.is_floating_point(src):
        raise ValueError("`scatter_softmax` can only be computed over tensors with floating point data types.")
    index = _broadcast(index, src, dim)
    max_value_per_index = scatter_max(src, index, dim=dim, dim_size=dim_size)[0]
    max_per_src_element = max_value_per_index.gather(dim, index)
    recentered = src - max_per_src_element
    recentered_exp = recentered.exp()
    sum_per_index = scatter_sum(recentered_exp, index, dim, dim_size=dim_size)
    normalizing = sum_per_index.gather(dim, index)
    return recentered_exp.div(normalizing)


# ------------------------------------------------------------------ torch_geometric
def degree(index, num_nodes=None, dtype=None):
    N = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros((N,), dtype=dtype, device=index.device)
    one = torch.ones((index.size(0),), dtype=out.dtype, device=out.device)
    return out.scatter_add_(0, index, one)


def remove_self_loops(edge_index, edge_attr=None):
    mask = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, mask]
    return edge_index, (None if edge_attr is None else edge_attr[mask])


def add_self_loops(edge_index, edge_weight=None, fill_value=1.0, num_nodes=None):
    N = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    loop = torch.arange(0, N, dtype=torch.long, device=edge_index.device).unsqueeze(0).repeat(2, 1)
    if edge_weight is not None:
        loop_w = edge_weight.new_full((N,), fill_value)
        edge_weight = torch.cat([edge_weight, loop_w], dim=0)
    return torch.cat([edge_index, loop], dim=1), edge_weight


class MessagePassing(torch.nn.Module):
    """PyG MessagePassing, flow='source_to_target', node_dim=-2 (1.6+): for an argument named
    ``foo_j`` gather ``foo[edge_index[0]]``, for ``foo_i`` gather ``foo[edge_index[1]]``; other
    message arguments are passed through; aggregate over ``edge_index[1]``."""

    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
        super().__init__()
        self.aggr = aggr
        self.flow = flow
        self.node_dim = node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        assert self.flow == "source_to_target"
        dim_size = None
        for v in kwargs.values():
            if isinstance(v, torch.Tensor) and v.dim() >= 2:
                dim_size = v.size(self.node_dim)
                break
        if size is not None:
            dim_size = size[1] if isinstance(size, (tuple, list)) else size
        msg_kwargs = {}
        for name, prm in inspect.signature(self.message).parameters.items():
            if name.endswith("_j") or name.endswith("_i"):
                data = kwargs.get(name[:-2])
                if data is None:
                    msg_kwargs[name] = None
                else:
                    sel = edge_index[0] if name.endswith("_j") else edge_index[1]
                    msg_kwargs[name] = data.index_select(self.node_dim, sel)
            elif name in kwargs:
                msg_kwargs[name] = kwargs[name]
            elif prm.default is not inspect.Parameter.empty:
                msg_kwargs[name] = prm.default
        out = self.message(**msg_kwargs)
        out = self.aggregate(out, index=edge_index[1], ptr=None, dim_size=dim_size)
        return self.update(out)

    def message(self, x_j):
        return x_j

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        return scatter(inputs, index, dim=self.node_dim, dim_size=dim_size, reduce=self.aggr)

    def update(self, inputs):
        return inputs


class EdgeConv(MessagePassing):
    """torch_geometric.nn.EdgeConv: max_j nn([x_i, x_j - x_i])."""

    def __init__(self, nn, aggr="max", **kwargs):
        super().__init__(aggr=aggr, **kwargs)
        self.nn = nn

    def forward(self, x, edge_index):
        return self.propagate(edge_index, x=x)

    def message(self, x_i, x_j):
        return self.nn(torch.cat([x_i, x_j - x_i], dim=-1))


# ------------------------------------------------------------------ torch_cluster
def knn_graph(x, k, batch=None, loop=False, flow="source_to_target"):
    """Brute-force restatement of torch_cluster.knn_graph: k nearest neighbours of every point
    within its batch segment, self excluded unless loop=True; row 0 = neighbour, row 1 = centre,
    grouped by centre.  (Ties are broken by index, an arbitrary but fixed choice.)"""
    n = x.size(0)
    if batch is None:
        batch = torch.zeros(n, dtype=torch.long, device=x.device)
    d = torch.cdist(x.double(), x.double())
    d = d.masked_fill(batch.view(-1, 1) != batch.view(1, -1), float("inf"))
    if not loop:
        d.fill_diagonal_(float("inf"))
    nbr = torch.topk(d, k, dim=1, largest=False, sorted=True).indices
    centre = torch.arange(n, device=x.device).view(-1, 1).expand(n, k)
    return torch.stack([nbr.reshape(-1), centre.reshape(-1)], dim=0)
