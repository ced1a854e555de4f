"""ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU restatement (plain PyTorch fp32/fp64 ops, autograd-differentiable) of the reference's
SPARSE message-passing path.  Each function cites the reference lines it follows; the
third-party primitives underneath come from oracle/thirdparty.py.

Pinned against the reference's own code by tests/test_oracle_golden.py (fixtures produced by
oracle/make_golden.py, which executes /root/reference/gcn_lib/sparse/*.py unmodified on top of
the same primitives) and against the hand-computed known answers of SURVEY.md §4.2.
The third-party layer itself (torch_scatter / PyG are not installable here) is restated from
the published algorithms: "parity unpinned" for that layer, see DESIGN.md.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import thirdparty as tp

POW_LO, POW_HI = 1e-7, 1e1  # gcn_lib/sparse/torch_message.py:69


def gen_message(x_j, edge_attr=None, eps=1e-7):
    """GENConv.message -- gcn_lib/sparse/torch_vertex.py:78-85."""
    msg = x_j + edge_attr if edge_attr is not None else x_j
    return F.relu(msg) + eps


def gen_aggregate_messages(inputs, index, dim_size, aggr="softmax", t=1.0, p=1.0, y=None,
                           learn_t=False):
    """GenMessagePassing.aggregate -- gcn_lib/sparse/torch_message.py:44-85.
    ``inputs`` (E,C) per-edge messages, ``index`` = edge_index[1]."""
    if aggr in ("add", "mean", "max"):                       # :46-47 -> PyG base aggregate
        return tp.scatter(inputs, index, dim=0, dim_size=dim_size, reduce=aggr)
    if aggr in ("softmax_sg", "softmax", "softmax_sum"):     # :49-65
        if learn_t:
            w = tp.scatter_softmax(inputs * t, index, dim=0)
        else:
            with torch.no_grad():
                w = tp.scatter_softmax(inputs * t, index, dim=0)
        out = tp.scatter(inputs * w, index, dim=0, dim_size=dim_size, reduce="sum")
        if aggr == "softmax_sum":
            sigmoid_y = torch.sigmoid(y)
            degrees = tp.degree(index, num_nodes=dim_size).unsqueeze(1)
            out = torch.pow(degrees, sigmoid_y) * out
        return out
    if aggr in ("power", "power_sum"):                       # :68-82
        inputs = torch.clamp(inputs, POW_LO, POW_HI)         # reference clamps in place
        out = tp.scatter(torch.pow(inputs, p), index, dim=0, dim_size=dim_size, reduce="mean")
        out = torch.clamp(out, POW_LO, POW_HI)
        out = torch.pow(out, 1 / p)
        if aggr == "power_sum":
            sigmoid_y = torch.sigmoid(y)
            degrees = tp.degree(index, num_nodes=dim_size).unsqueeze(1)
            out = torch.pow(degrees, sigmoid_y) * out
        return out
    raise NotImplementedError("To be implemented")            # :85


def gen_propagate(x, edge_index, edge_attr=None, aggr="softmax", t=1.0, p=1.0, y=None,
                  learn_t=False, eps=1e-7, dim_size=None):
    """propagate = gather x_j = x[edge_index[0]] -> message -> aggregate over edge_index[1]
    (PyG flow source_to_target; gcn_lib/sparse/torch_vertex.py:68)."""
    x_j = x.index_select(0, edge_index[0])
    msg = gen_message(x_j, edge_attr, eps)
    n = x.size(0) if dim_size is None else dim_size
    return gen_aggregate_messages(msg, edge_index[1], n, aggr, t, p, y, learn_t)


def msg_norm(x, msg, msg_scale, p=2):
    """MsgNorm.forward -- gcn_lib/sparse/torch_message.py:95-99."""
    msg = F.normalize(msg, p=p, dim=1)
    x_norm = x.norm(p=p, dim=1, keepdim=True)
    return msg * x_norm * msg_scale


def genconv_forward(x, edge_index, mlp, edge_attr=None, edge_encoder=None, msg_scale=None, **aggr_kw):
    """GENConv.forward -- gcn_lib/sparse/torch_vertex.py:62-76 (mlp / edge_encoder are modules)."""
    edge_emb = edge_encoder(edge_attr) if (edge_encoder is not None and edge_attr is not None) else edge_attr
    m = gen_propagate(x, edge_index, edge_emb, **aggr_kw)
    if msg_scale is not None:
        m = msg_norm(x, m, msg_scale)
    return mlp(x + m)


def scatter_(name, src, index, dim=0, dim_size=None):
    """utils/pyg_util.py:4-35 (incl. the `< -10000 -> 0` fix-up for max, `> 10000` for min)."""
    assert name in ["add", "mean", "min", "max"]
    out = tp.scatter(src, index, dim=dim, dim_size=dim_size, reduce=name)
    if name == "max":
        out = torch.where(out < -10000, torch.zeros_like(out), out)
    elif name == "min":
        out = torch.where(out > 10000, torch.zeros_like(out), out)
    return out


def mr_aggregate(x, edge_index, aggr="max"):
    """The aggregation inside sparse MRConv.forward -- gcn_lib/sparse/torch_vertex.py:102:
    max_{j in N(i)} (x_j - x_i), 0 for isolated nodes."""
    diff = torch.index_select(x, 0, edge_index[0]) - torch.index_select(x, 0, edge_index[1])
    return scatter_(aggr, diff, edge_index[1], dim_size=x.shape[0])


def mrconv_forward(x, edge_index, nn):
    """MRConv.forward -- gcn_lib/sparse/torch_vertex.py:100-103."""
    return nn(torch.cat([x, mr_aggregate(x, edge_index)], dim=1))


def edgeconv_forward(x, edge_index, nn):
    """EdgConv.forward -> tg.nn.EdgeConv -- gcn_lib/sparse/torch_vertex.py:106-114:
    max_{j->i} nn([x_i, x_j - x_i]); empty rows 0."""
    x_i = x.index_select(0, edge_index[1])
    x_j = x.index_select(0, edge_index[0])
    msg = nn(torch.cat([x_i, x_j - x_i], dim=-1))
    return tp.scatter(msg, edge_index[1], dim=0, dim_size=x.size(0), reduce="max")
