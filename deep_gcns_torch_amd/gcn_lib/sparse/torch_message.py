"""Generalized message aggregation (SoftMax / PowerMean / add / mean / max) -- the public
surface of the reference's gcn_lib/sparse/torch_message.py (GenMessagePassing :8-85,
MsgNorm :88-99) without PyG: gather + message + aggregate run as ONE fused HIP kernel
(deep_gcns_torch_amd.ops.gen_aggregate), so no (E, C) tensor ever exists.
"""
import torch
import torch.nn.functional as F

from ... import node_ops, ops
from ...graph import Graph, graph_of, scatter_graph_of

__all__ = ["GenMessagePassing", "MsgNorm"]


def _active_partition():
    import sys
    d = sys.modules.get("deep_gcns_torch_amd.dist")      # only a process that imported dist can have an active partition
    return d.active_partition() if d is not None else None

_SOFTMAX = ("softmax_sg", "softmax", "softmax_sum")
_POWER = ("power", "power_sum")


class GenMessagePassing(torch.nn.Module):
    """Holds the aggregator configuration exactly like the reference (:9-42):

    * ``t`` is a Parameter of shape [1] only when ``learn_t`` and aggr in {softmax, softmax_sum},
      otherwise the python float; ``self.learn_t`` exists for the softmax family only;
    * ``p`` is a Parameter when ``learn_p`` (power family), otherwise the float;
    * ``y`` is always a Parameter for the ``*_sum`` variants (``requires_grad=learn_y``).
    """

    node_dim = -2  # PyG >= 1.6 convention the reference relies on (dim=self.node_dim)

    def __init__(self, aggr="softmax", t=1.0, learn_t=False, p=1.0, learn_p=False, y=0.0, learn_y=False):
        super().__init__()
        self.aggr = aggr
        if aggr in _SOFTMAX:
            if learn_t and aggr in ("softmax", "softmax_sum"):
                self.learn_t = True
                self.t = torch.nn.Parameter(torch.Tensor([t]), requires_grad=True)
            else:
                self.learn_t = False
                self.t = t
            if aggr == "softmax_sum":
                self.y = torch.nn.Parameter(torch.Tensor([y]), requires_grad=learn_y)
        elif aggr in _POWER:
            self.p = torch.nn.Parameter(torch.Tensor([p]), requires_grad=True) if learn_p else p
            if aggr == "power_sum":
                self.y = torch.nn.Parameter(torch.Tensor([y]), requires_grad=learn_y)

    # -- fused path ---------------------------------------------------------------------
    def fusable_root(self) -> bool:
        """Whether ``x + aggregate`` can come out of the aggregation kernel itself (no degree scaling, no learnable
        temperature / exponent: their gradients read the bare aggregate)."""
        aggr = self.aggr
        if aggr in ("softmax_sum", "power_sum"):
            return False
        if aggr in _SOFTMAX and self.learn_t:
            return False
        if aggr in _POWER and isinstance(self.p, torch.nn.Parameter):
            return False
        return True

    def _aggregate_fused(self, x, graph: Graph, edge_attr=None, relu_eps=True, eps=1e-7, add_root=False,
                         edge_encoder=None):
        """AGGR_i over relu(x_src + edge_attr) + eps (or the raw rows), then the optional
        degree scaling deg^sigmoid(y) of the ``*_sum`` variants (torch_message.py:60-63,77-80)."""
        aggr = self.aggr
        if aggr is None or aggr in ("add", "mean", "max"):
            out = ops.gen_aggregate(x, graph, edge_attr, aggr=aggr or "add", relu_eps=relu_eps, eps=eps,
                                    add_root=add_root, edge_encoder=edge_encoder)
        elif aggr in _SOFTMAX:
            out = ops.gen_aggregate(x, graph, edge_attr, aggr=aggr, t=self.t, learn_t=self.learn_t,
                                    relu_eps=relu_eps, eps=eps, add_root=add_root, edge_encoder=edge_encoder)
        elif aggr in _POWER:
            out = ops.gen_aggregate(x, graph, edge_attr, aggr=aggr, p=self.p,
                                    learn_p=isinstance(self.p, torch.nn.Parameter), relu_eps=relu_eps, eps=eps,
                                    add_root=add_root, edge_encoder=edge_encoder)
        else:
            raise NotImplementedError("To be implemented")
        if aggr in ("softmax_sum", "power_sum"):
            self.sigmoid_y = torch.sigmoid(self.y)
            out = torch.pow(graph.deg.unsqueeze(1), self.sigmoid_y) * out
        return out

    def _aggregate_partitioned(self, ctx, x, edge_attr, add_root, edge_encoder):
        """Inside ``with dist.partitioned(part)``: ``x`` holds this rank's rows, the aggregation runs through the
        partition's exchange scheme (deep_gcns_torch_amd/dist.py), the ``edge_index`` argument is not looked at."""
        from ... import dist as _dist
        if edge_attr is not None or edge_encoder is not None:
            raise NotImplementedError("node-partitioned layers: edge features are not partitioned yet")
        aggr = self.aggr
        kw = dict(eps=getattr(self, "eps", 1e-7))
        if aggr in ("softmax_sum", "power_sum"):
            raise NotImplementedError("node-partitioned layers: the degree-scaled (*_sum) aggregators are not wired up")
        if aggr in _SOFTMAX:
            kw.update(t=self.t)
            if self.learn_t:
                kw.update(learn_t=True)
        elif aggr in _POWER:
            kw.update(p=self.p)
            if isinstance(self.p, torch.nn.Parameter):
                kw.update(learn_p=True)
        out = _dist.partition_aggregate(ctx, x, aggr or "add", **kw)
        return x + out if add_root else out

    # -- PyG-flavoured entry points kept for API compatibility ----------------------------
    def propagate(self, edge_index, size=None, x=None, edge_attr=None, add_root=False, edge_encoder=None):
        """``propagate(edge_index, x=x, edge_attr=edge_attr)`` as GENConv.forward calls it
        (gcn_lib/sparse/torch_vertex.py:68): message + aggregate + update in one kernel.  ``add_root`` (extension)
        returns ``x + aggregate`` from the same kernel when ``fusable_root()``; ``edge_encoder=(weight, bias)``
        (extension) takes ``edge_attr`` as RAW features and applies the Linear edge encoder inside the kernels."""
        part = _active_partition()
        if part is not None:
            return self.update(self._aggregate_partitioned(part, x, edge_attr, add_root, edge_encoder))
        n = x.size(0) if size is None else (size[1] if isinstance(size, (tuple, list)) else size)
        return self.update(self._aggregate_fused(x, graph_of(edge_index, n), edge_attr, relu_eps=True,
                                                 eps=getattr(self, "eps", 1e-7), add_root=add_root,
                                                 edge_encoder=edge_encoder))

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        """Aggregate an ALREADY materialised (E, C) message tensor by destination ``index``
        (torch_message.py:44).  Prefer ``propagate``: it never builds ``inputs``."""
        n = int(dim_size) if dim_size is not None else int(index.max()) + 1
        return self._aggregate_fused(inputs, scatter_graph_of(index, n), None, relu_eps=False)

    def update(self, aggr_out):
        return aggr_out


class MsgNorm(torch.nn.Module):
    """msg <- normalize(msg) * ||x||_2 * s   (torch_message.py:88-99)."""

    def __init__(self, learn_msg_scale=False):
        super().__init__()
        self.msg_scale = torch.nn.Parameter(torch.Tensor([1.0]), requires_grad=learn_msg_scale)

    def forward(self, x, msg, p=2, add_x=False):
        """``add_x`` (extension) returns ``x + msg_norm(x, msg)``: GENConv's residual from the same kernel."""
        if p == 2 and node_ops.msg_norm_supported(x, msg):
            return node_ops.msg_norm_rows(x, msg, self.msg_scale, add_x=add_x)        # one HIP row kernel
        unit = F.normalize(msg, p=p, dim=1)
        out = unit * x.norm(p=p, dim=1, keepdim=True) * self.msg_scale
        return x + out if add_x else out
