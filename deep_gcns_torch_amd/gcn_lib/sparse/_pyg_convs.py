"""Thin shells around torch_geometric convolutions (reference gcn_lib/sparse/torch_vertex.py
:117-236).  Imported only when torch_geometric is installed; out of the hot-path scope."""
import torch
import torch_geometric as tg
from torch import nn

from .torch_nn import MLP, act_layer, norm_layer


class _ActNorm(nn.Module):
    def __init__(self, conv, out_channels, act, norm):
        super().__init__()
        tail = []
        if act:
            tail.append(act_layer(act))
        if norm:
            tail.append(norm_layer(norm, out_channels))
        self.gconv = conv
        self.unlinear = nn.Sequential(*tail)

    def forward(self, x, edge_index):
        return self.unlinear(self.gconv(x, edge_index))


class GATConv(_ActNorm):
    def __init__(self, in_channels, out_channels, act='relu', norm=None, bias=True, heads=8):
        super().__init__(tg.nn.GATConv(in_channels, out_channels, heads, bias=bias), out_channels, act, norm)


class SemiGCNConv(_ActNorm):
    def __init__(self, in_channels, out_channels, act='relu', norm=None, bias=True):
        super().__init__(tg.nn.GCNConv(in_channels, out_channels, bias=bias), out_channels, act, norm)


class SAGEConv(tg.nn.SAGEConv):
    def __init__(self, in_channels, out_channels, nn, norm=True, bias=True, relative=False, **kwargs):
        self.relative = relative
        super().__init__(in_channels, out_channels, None if norm is None else True, bias, **kwargs)
        self.nn = nn


class RSAGEConv(SAGEConv):
    def __init__(self, in_channels, out_channels, act='relu', norm=None, bias=True, relative=False):
        super().__init__(in_channels, out_channels, MLP([out_channels + in_channels, out_channels], act, norm, bias),
                         norm, bias, relative)


class GinConv(tg.nn.GINConv):
    def __init__(self, in_channels, out_channels, act='relu', norm=None, bias=True, aggr='add'):
        super().__init__(MLP([in_channels, out_channels], act, norm, bias))
