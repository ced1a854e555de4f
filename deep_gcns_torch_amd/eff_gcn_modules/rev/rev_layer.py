"""Blocks wrapped by the reversible coupling: public surface of the reference's eff_gcn_modules/rev/rev_layer.py
(SharedDropout :12-24, BasicBlock :27-48, GENBlock :51-75; the GCN / SAGE / GAT blocks :78-115 are shells around
PyG convolutions and exist only when torch_geometric is installed)."""
import torch.nn as nn
import torch.nn.functional as F

from ... import node_ops
from ...gcn_lib.sparse.torch_nn import norm_layer
from ...gcn_lib.sparse.torch_vertex import GENConv

__all__ = ["SharedDropout", "BasicBlock", "GENBlock", "GCNBlock", "SAGEBlock", "GATBlock"]


class SharedDropout(nn.Module):
    """Dropout with a mask handed in by the model, identical in forward, inverse and recompute."""

    def __init__(self):
        super().__init__()
        self.mask = None

    def set_mask(self, mask):
        self.mask = mask

    def forward(self, x):
        if not self.training:
            return x
        assert self.mask is not None
        return x * self.mask


class BasicBlock(nn.Module):
    """norm -> ReLU -> shared dropout -> graph convolution.  On device rows the first three are ONE row kernel
    (node_ops.pre_activation: LayerNorm / BatchNorm1d apply pass with the ReLU and the mask multiply folded in; the
    backward recomputes the ReLU mask from x instead of keeping the activated tensor)."""

    def __init__(self, norm, in_channels):
        super().__init__()
        self.norm = norm_layer(norm, in_channels)
        self.dropout = SharedDropout()

    def forward(self, x, edge_index, dropout_mask=None, edge_emb=None, residual=None):
        """``residual`` (extension, ``node_ops.CouplingResidual`` from the additive coupling): handed to the convolution,
        whose last Linear may fold ``x_i +/- F_i(.)`` into its epilogue (GENBlock; the offer is ignored otherwise)."""
        shared = isinstance(self.dropout, SharedDropout)
        if shared and dropout_mask is not None:
            self.dropout.set_mask(dropout_mask)
        if shared and x.is_cuda and isinstance(self.norm, (node_ops.BatchNorm1d, node_ops.LayerNorm)):
            if self.training:
                assert self.dropout.mask is not None
            out = node_ops.pre_activation(self.norm, x, training=self.training,
                                          mask=self.dropout.mask if self.training else None)
        else:
            out = self.dropout(F.relu(self.norm(x)))
        if residual is not None and type(self.gcn) is GENConv:
            return self.gcn(out, edge_index, edge_emb, residual=residual)
        if edge_emb is not None:
            return self.gcn(out, edge_index, edge_emb)
        return self.gcn(out, edge_index)


class GENBlock(BasicBlock):
    def __init__(self, in_channels, out_channels, aggr='max', t=1.0, learn_t=False, p=1.0, learn_p=False, y=0.0,
                 learn_y=False, msg_norm=False, learn_msg_scale=False, encode_edge=False, edge_feat_dim=0,
                 norm='layer', mlp_layers=1):
        super().__init__(norm, in_channels)
        self.gcn = GENConv(in_channels, out_channels, aggr=aggr, t=t, learn_t=learn_t, p=p, learn_p=learn_p, y=y,
                           learn_y=learn_y, msg_norm=msg_norm, learn_msg_scale=learn_msg_scale,
                           encode_edge=encode_edge, edge_feat_dim=edge_feat_dim, norm=norm, mlp_layers=mlp_layers)


def _needs_pyg(name):
    class _Missing(BasicBlock):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"{name} wraps a torch_geometric convolution that the message-passing hot path "
                                      "does not name; install torch_geometric and use the reference's block")
    _Missing.__name__ = name
    return _Missing


GCNBlock = _needs_pyg("GCNBlock")
SAGEBlock = _needs_pyg("SAGEBlock")
GATBlock = _needs_pyg("GATBlock")
