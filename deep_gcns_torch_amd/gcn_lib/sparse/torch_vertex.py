"""Sparse graph convolutions and blocks: same classes, signatures and ``state_dict`` keys as the
reference's gcn_lib/sparse/torch_vertex.py, aggregation done by libdgcn's fused kernels.

  GENConv :12-88 | MRConv :91-103 | EdgConv :106-114 | GraphConv :239-264 | DynConv :267-281
  Plain/Res/DenseDynBlock :284-325 | Res/DenseGraphBlock :328-351
GAT/SAGE/RSAGE/SemiGCN/Gin wrappers (:117-236) are thin shells around PyG convolutions that the
hot path does not name; they are exposed only when torch_geometric is importable.
"""
import torch
from torch import nn

from ... import blocks, ops
from ...graph import graph_of
from .torch_edge import DilatedKnnGraph
from .torch_message import GenMessagePassing, MsgNorm
from ...nn_util import TallLinear
from .torch_nn import MLP, BondEncoder, act_layer, norm_layer  # noqa: F401

__all__ = ["GENConv", "MRConv", "EdgConv", "GATConv", "SAGEConv", "RSAGEConv", "SemiGCNConv", "GinConv",
           "GraphConv", "DynConv", "PlainDynBlock", "ResDynBlock", "DenseDynBlock", "ResGraphBlock",
           "DenseGraphBlock"]


class GENConv(GenMessagePassing):
    """GENeralized graph convolution (https://arxiv.org/abs/2006.07739):
    out = MLP( x + [MsgNorm]( AGGR_{j->i} relu(x_j [+ e_ji]) + eps ) )."""

    def __init__(self, in_dim, emb_dim, aggr='softmax', t=1.0, learn_t=False, p=1.0, learn_p=False,
                 y=0.0, learn_y=False, msg_norm=False, learn_msg_scale=True, encode_edge=False,
                 bond_encoder=False, edge_feat_dim=None, norm='batch', mlp_layers=2, eps=1e-7):
        super().__init__(aggr=aggr, t=t, learn_t=learn_t, p=p, learn_p=learn_p, y=y, learn_y=learn_y)
        widths = [in_dim] + [in_dim * 2] * (mlp_layers - 1) + [emb_dim]
        self.mlp = MLP(channels=widths, norm=norm, last_lin=True)
        self.msg_encoder = nn.ReLU()   # applied inside the kernel; kept as an attribute for parity
        self.eps = eps
        self.encode_edge = encode_edge
        self.bond_encoder = bond_encoder
        self.msg_norm = MsgNorm(learn_msg_scale=learn_msg_scale) if msg_norm else None
        if encode_edge:
            self.edge_encoder = BondEncoder(emb_dim=in_dim) if bond_encoder else TallLinear(edge_feat_dim, in_dim)

    def forward(self, x, edge_index, edge_attr=None, residual=None, want_stats=False):
        """``residual`` (extension): returns ``conv(x) + residual`` with the add folded into the last Linear's epilogue --
        the 'res+' skip connection ``h = conv(h2) + h`` (examples/ogb/ogbn_arxiv/model.py:104).  ``want_stats``
        (extension): returns ``(out, stats)`` where ``stats`` are the partial column sums of ``out`` that the next
        layer's BatchNorm1d takes (None when the row kernel did not run): its statistics pass disappears."""
        root = self.msg_norm is None and self.fusable_root() and x.dim() == 2      # h = x + m inside the kernel
        enc = None
        if isinstance(edge_attr, blocks.ComposedEdgeEmbedding):
            # two Linear maps in a row on the raw edge features: composed, evaluated per edge inside the kernels
            ce = edge_attr
            lin = getattr(self, "edge_encoder", None)
            if (self.encode_edge and isinstance(lin, nn.Linear) and ce.repeat == 1 and x.is_cuda and x.dim() == 2
                    and ops.encoder_fusable(x, ce.raw, None, narrow=True)):
                enc, edge_attr = ce.composed(lin), ce.raw
            else:
                edge_attr = ce.materialize()
        if enc is not None:
            edge_emb = edge_attr
        elif self.encode_edge and edge_attr is not None:
            lin = self.edge_encoder
            if isinstance(lin, nn.Linear) and x.is_cuda and ops.encoder_fusable(x, edge_attr, lin.weight):
                enc, edge_emb = (lin.weight, lin.bias), edge_attr              # Linear(hidden -> C) inside the kernels
            else:
                edge_emb = lin(edge_attr)
        else:
            edge_emb = edge_attr
        if root:
            h = self.propagate(edge_index, x=x, edge_attr=edge_emb, add_root=True, edge_encoder=enc)
        else:
            m = self.propagate(edge_index, x=x, edge_attr=edge_emb, edge_encoder=enc)
            if self.msg_norm is not None:
                h = self.msg_norm(x, m, add_x=True)                 # x + MsgNorm(x, m) in one row kernel
            else:
                h = x + m
        if residual is None and not want_stats:
            return self.mlp(h)
        return self.mlp(h, residual=residual, want_stats=want_stats)

    def message(self, x_j, edge_attr=None):
        """Reference semantics of one message (torch_vertex.py:78-85); the fused kernel computes
        exactly this per edge, this method exists for callers that want the formula."""
        z = x_j if edge_attr is None else x_j + edge_attr
        return self.msg_encoder(z) + self.eps


def _max_relative(x, edge_index, aggr='max'):
    """r_i = AGGR_{j->i}(x_j - x_i) with the reference's scatter_ fix-up (utils/pyg_util.py:26-31).
    x_i is constant over a row and fl(a - b) is monotone in a, so for max/min the reduction is done
    on the raw neighbour rows and x_i subtracted once per node: bit-identical, one gather not two."""
    g = graph_of(edge_index, x.size(0))
    has = (g.deg > 0).unsqueeze(1)
    if aggr == 'max':
        r = ops.gen_aggregate(x, g, aggr='max', relu_eps=False) - x
        r = torch.where(has, r, torch.zeros_like(r))
        return torch.where(r < -10000, torch.zeros_like(r), r)
    if aggr == 'min':
        r = -ops.gen_aggregate(-x, g, aggr='max', relu_eps=False) - x
        r = torch.where(has, r, torch.zeros_like(r))
        return torch.where(r > 10000, torch.zeros_like(r), r)
    if aggr == 'add':
        return ops.gen_aggregate(x, g, aggr='add', relu_eps=False) - g.deg.unsqueeze(1) * x
    if aggr == 'mean':
        r = ops.gen_aggregate(x, g, aggr='mean', relu_eps=False) - x
        return torch.where(has, r, torch.zeros_like(r))
    raise AssertionError(aggr)


class MRConv(nn.Module):
    """Max-Relative graph convolution (https://arxiv.org/abs/1904.03751): nn([x, max_j (x_j - x_i)])."""

    def __init__(self, in_channels, out_channels, act='relu', norm=None, bias=True, aggr='max'):
        super().__init__()
        self.nn = MLP([in_channels * 2, out_channels], act, norm, bias)
        self.aggr = aggr

    def forward(self, x, edge_index):
        return self.nn(torch.cat([x, _max_relative(x, edge_index, self.aggr)], dim=1))


class EdgConv(nn.Module):
    """Edge convolution max_{j->i} nn([x_i, x_j - x_i]) with nn = Linear -> [BatchNorm1d] -> act
    (reference: torch_vertex.py:106-114 on top of tg.nn.EdgeConv).

    The Linear is split per VERTEX: W [x_i ; x_j - x_i] + b = ((W1 - W2) x_i + b) + W2 x_j = P_i + Q_j,
    a per-edge affine BatchNorm and a monotone activation commute with the max up to the sign of
    the BN scale, so    out_i = act( g * (max_j or min_j of Q_j, by sign of g) + g * P_i + h ).
    Training-mode batch statistics over all E edge rows are obtained from node-level sums plus ONE
    add-aggregation (sum_e P_dst Q_src = sum_i P_i * sum_{j->i} Q_j); nothing of size (E, C) is built.
    """

    def __init__(self, in_channels, out_channels, act='relu', norm=None, bias=True, aggr='max'):
        super().__init__()
        self.nn = MLP([in_channels * 2, out_channels], act, norm, bias)
        self.aggr = aggr
        self.in_channels = in_channels
        if aggr not in ('max', 'add', 'mean'):
            raise NotImplementedError("EdgConv: aggr must be one of max / add / mean (tg.nn.EdgeConv)")
        # The per-vertex split needs a reduction that commutes with the per-edge affine norm and a monotone
        # activation: max with BatchNorm / no norm and ReLU / LeakyReLU.  Every other option of the reference
        # (aggr add / mean; layer or instance norm; PReLU) is evaluated per edge like the reference does
        # (torch_vertex.py:111 -> tg.nn.EdgeConv.message) and reduced by the aggregation kernel.
        self._per_edge = aggr != 'max' or any(isinstance(m, (nn.LayerNorm, nn.InstanceNorm1d, nn.PReLU))
                                              for m in self.nn)

    def _forward_per_edge(self, x, edge_index):
        dst = edge_index[1]
        x_i, x_j = x.index_select(0, dst), x.index_select(0, edge_index[0])
        msg = self.nn(torch.cat([x_i, x_j - x_i], dim=1))            # (E, C')
        from ...graph import scatter_graph_of
        return ops.gen_aggregate(msg, scatter_graph_of(dst, x.size(0)), aggr=self.aggr, relu_eps=False)

    def forward(self, x, edge_index):
        if self._per_edge:
            return self._forward_per_edge(x, edge_index)
        lin = self.nn[0]
        C = self.in_channels
        g = graph_of(edge_index, x.size(0))
        w1, w2 = lin.weight[:, :C], lin.weight[:, C:]
        P = torch.nn.functional.linear(x, w1 - w2, lin.bias)
        Q = torch.nn.functional.linear(x, w2)
        bn = next((m for m in self.nn if isinstance(m, nn.BatchNorm1d)), None)
        qmax = ops.gen_aggregate(Q, g, aggr='max', relu_eps=False)
        if bn is None:
            pre = P + qmax
        else:
            E = max(g.n_edges, 1)
            if bn.training or not bn.track_running_stats:
                din, dout = g.deg.unsqueeze(1), g.out_deg.unsqueeze(1)
                mu_p, mu_q = (din * P).sum(0) / E, (dout * Q).sum(0) / E
                mean = mu_p + mu_q
                Pc, Qc = P - mu_p, Q - mu_q               # centred: a_e - mean = Pc_dst + Qc_src
                S = ops.gen_aggregate(Qc, g, aggr='add', relu_eps=False)
                var = ((din * Pc * Pc).sum(0) + (dout * Qc * Qc).sum(0) + 2.0 * (Pc * S).sum(0)) / E
                var = var.clamp_min(0.0)
                if bn.track_running_stats and bn.training:
                    with torch.no_grad():
                        mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                        bn.running_mean.mul_(1 - mom).add_(mom * mean)
                        bn.running_var.mul_(1 - mom).add_(mom * var * (E / max(E - 1, 1)))
                        bn.num_batches_tracked += 1
            else:
                mean, var = bn.running_mean, bn.running_var
            scale = bn.weight * torch.rsqrt(var + bn.eps)
            shift = bn.bias - mean * scale
            qmin = -ops.gen_aggregate(-Q, g, aggr='max', relu_eps=False)
            pre = scale * (P + torch.where(scale >= 0, qmax, qmin)) + shift
        out = pre
        for m in list(self.nn.children())[1:]:
            if not isinstance(m, nn.BatchNorm1d):
                out = m(out)                                # activation (+ dropout)
        return torch.where((g.deg > 0).unsqueeze(1), out, torch.zeros_like(out))


def _pyg_wrappers():
    """GAT/SAGE/GCN/GIN shells (reference :117-236) exist only with torch_geometric installed."""
    try:
        import torch_geometric as tg  # noqa: F401
    except Exception:
        def _missing(name):
            class _Unavailable(nn.Module):
                def __init__(self, *a, **k):
                    raise NotImplementedError(
                        f"{name} wraps a torch_geometric convolution (out of the hot-path scope, "
                        "SURVEY.md §2a #5) and torch_geometric is not installed")
            _Unavailable.__name__ = name
            return _Unavailable
        return {n: _missing(n) for n in ("GATConv", "SAGEConv", "RSAGEConv", "SemiGCNConv", "GinConv")}
    from ._pyg_convs import GATConv, SAGEConv, RSAGEConv, SemiGCNConv, GinConv
    return dict(GATConv=GATConv, SAGEConv=SAGEConv, RSAGEConv=RSAGEConv, SemiGCNConv=SemiGCNConv, GinConv=GinConv)


globals().update(_pyg_wrappers())


class GraphConv(nn.Module):
    """Static graph convolution dispatcher (conv in edge|mr|gat|gcn|gin|sage|rsage)."""

    def __init__(self, in_channels, out_channels, conv='edge', act='relu', norm=None, bias=True, heads=8):
        super().__init__()
        kind = conv.lower()
        if kind == 'edge':
            self.gconv = EdgConv(in_channels, out_channels, act, norm, bias)
        elif kind == 'mr':
            self.gconv = MRConv(in_channels, out_channels, act, norm, bias)
        elif kind == 'gat':
            self.gconv = GATConv(in_channels, out_channels // heads, act, norm, bias, heads)  # noqa: F821
        elif kind == 'gcn':
            self.gconv = SemiGCNConv(in_channels, out_channels, act, norm, bias)  # noqa: F821
        elif kind == 'gin':
            self.gconv = GinConv(in_channels, out_channels, act, norm, bias)  # noqa: F821
        elif kind in ('sage', 'rsage'):
            self.gconv = RSAGEConv(in_channels, out_channels, act, norm, bias, kind == 'rsage')  # noqa: F821
        else:
            raise NotImplementedError('conv {} is not implemented'.format(conv))

    def forward(self, x, edge_index):
        return self.gconv(x, edge_index)


class DynConv(GraphConv):
    """Dynamic graph convolution: (dilated) kNN graph on the current features, then GraphConv."""

    def __init__(self, in_channels, out_channels, kernel_size=9, dilation=1, conv='edge', act='relu',
                 norm=None, bias=True, heads=8, **kwargs):
        super().__init__(in_channels, out_channels, conv, act, norm, bias, heads)
        self.k = kernel_size
        self.d = dilation
        self.dilated_knn_graph = DilatedKnnGraph(kernel_size, dilation, **kwargs)

    def forward(self, x, batch=None, edge_index=None):
        if edge_index is None:
            edge_index = self.dilated_knn_graph(x, batch)
        return super().forward(x, edge_index)


class PlainDynBlock(nn.Module):
    def __init__(self, channels, kernel_size=9, dilation=1, conv='edge', act='relu', norm=None,
                 bias=True, res_scale=1, **kwargs):
        super().__init__()
        self.body = DynConv(channels, channels, kernel_size, dilation, conv, act, norm, bias, **kwargs)
        self.res_scale = res_scale

    def forward(self, x, batch=None, edge_index=None):
        return self.body(x, batch, edge_index), batch


class ResDynBlock(nn.Module):
    def __init__(self, channels, kernel_size=9, dilation=1, conv='edge', act='relu', norm=None,
                 bias=True, res_scale=1, **kwargs):
        super().__init__()
        self.body = DynConv(channels, channels, kernel_size, dilation, conv, act, norm, bias, **kwargs)
        self.res_scale = res_scale

    def forward(self, x, batch=None, edge_index=None):
        return self.body(x, batch, edge_index) + x * self.res_scale, batch


class DenseDynBlock(nn.Module):
    def __init__(self, in_channels, out_channels=64, kernel_size=9, dilation=1, conv='edge', act='relu',
                 norm=None, bias=True, **kwargs):
        super().__init__()
        self.body = DynConv(in_channels, out_channels, kernel_size, dilation, conv, act, norm, bias, **kwargs)

    def forward(self, x, batch=None, edge_index=None):
        return torch.cat((x, self.body(x, batch, edge_index)), 1), batch


class ResGraphBlock(nn.Module):
    def __init__(self, channels, conv='edge', act='relu', norm=None, bias=True, heads=8, res_scale=1):
        super().__init__()
        self.body = GraphConv(channels, channels, conv, act, norm, bias, heads)
        self.res_scale = res_scale

    def forward(self, x, edge_index):
        return self.body(x, edge_index) + x * self.res_scale, edge_index


class DenseGraphBlock(nn.Module):
    def __init__(self, in_channels, out_channels, conv='edge', act='relu', norm=None, bias=True, heads=8):
        super().__init__()
        self.body = GraphConv(in_channels, out_channels, conv, act, norm, bias, heads)

    def forward(self, x, edge_index):
        return torch.cat((x, self.body(x, edge_index)), 1), edge_index
