"""Node-wise building blocks of the sparse library (library GEMMs; BatchNorm1d [+ReLU] on the HIP row kernels).

Mirrors the public surface of the reference's gcn_lib/sparse/torch_nn.py -- act_layer :9-20,
norm_layer :23-34, MultiSeq :37-47, MLP :50-71, AtomEncoder :74-92, BondEncoder :95-113 --
with identical constructor signatures and ``state_dict`` keys (Sequential indices: Linear,
norm, act, [dropout] per hidden layer; nothing after the last Linear when ``last_lin``).
"""
import torch
from torch import nn

from ...nn_util import TallLinear
from ...node_ops import BatchNorm1d, CouplingResidual, LayerNorm
from ...utils.data_util import get_atom_feature_dims, get_bond_feature_dims

__all__ = ["act_layer", "norm_layer", "MultiSeq", "MLP", "AtomEncoder", "BondEncoder"]


def act_layer(act_type, inplace=False, neg_slope=0.2, n_prelu=1):
    """Activation factory: relu | leakyrelu | prelu."""
    kind = act_type.lower()
    if kind == "relu":
        return nn.ReLU(inplace)
    if kind == "leakyrelu":
        return nn.LeakyReLU(neg_slope, inplace)
    if kind == "prelu":
        return nn.PReLU(num_parameters=n_prelu, init=neg_slope)
    raise NotImplementedError("activation layer [%s] is not found" % kind)


def norm_layer(norm_type, nc):
    """1-D normalisation factory: batch | layer | instance."""
    kind = norm_type.lower()
    if kind == "batch":
        return BatchNorm1d(nc, affine=True)          # an nn.BatchNorm1d; (rows, C) inputs run on the HIP kernels
    if kind == "layer":
        return LayerNorm(nc, elementwise_affine=True)    # an nn.LayerNorm; device rows run on the HIP kernels
    if kind == "instance":
        return nn.InstanceNorm1d(nc, affine=False)
    raise NotImplementedError("normalization layer [%s] is not found" % kind)


def _enabled(opt):
    return opt is not None and opt.lower() != "none"


class MultiSeq(nn.Sequential):
    """Sequential whose stages may return tuples that are splatted into the next stage
    (blocks return ``(features, edge_index)``)."""

    def forward(self, *inputs):
        for stage in self._modules.values():
            inputs = stage(*inputs) if type(inputs) == tuple else stage(inputs)
        return inputs


class MLP(nn.Sequential):
    """Linear -> norm -> act -> dropout per layer (norm BEFORE act, unlike the dense BasicConv);
    with ``last_lin`` the final Linear stays bare."""

    def __init__(self, channels, act="relu", norm=None, bias=True, drop=0., last_lin=False):
        stages = []
        last = len(channels) - 1
        for i in range(1, len(channels)):
            stages.append(TallLinear(channels[i - 1], channels[i], bias))   # an nn.Linear; split-K weight grad
            if last_lin and i == last:
                continue
            if _enabled(norm):
                stages.append(norm_layer(norm, channels[i]))
            if _enabled(act):
                stages.append(act_layer(act))
            if drop > 0:
                stages.append(nn.Dropout2d(drop))
        self.m = stages
        super().__init__(*stages)

    def forward(self, x, residual=None, want_stats: bool = False):
        """Same stage order as nn.Sequential.  Fused on device rows: BatchNorm1d / LayerNorm directly followed by ReLU is
        one kernel; a Linear in front of a training-mode BatchNorm1d hands it the statistics of its own output (no
        separate statistics pass); ``residual`` (extension) is added in the LAST Linear's epilogue and ``want_stats``
        (extension) returns ``(y, stats)`` with the last Linear's output statistics for the caller's next BatchNorm1d."""
        mods = list(self._modules.values())
        last_lin = max((i for i, m in enumerate(mods) if isinstance(m, TallLinear)), default=-1)
        if isinstance(residual, CouplingResidual) and last_lin != len(mods) - 1:
            residual = None                       # an offer, not a request: the coupling adds it itself
        ext = residual is not None or want_stats
        if ext and last_lin != len(mods) - 1:
            raise ValueError("residual / want_stats need the MLP to end with its Linear (last_lin=True)")
        stats = None
        i = 0
        while i < len(mods):
            m = mods[i]
            rows2d = isinstance(x, torch.Tensor) and x.dim() == 2
            if isinstance(m, TallLinear) and rows2d:
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                if i == last_lin and ext:
                    out = m(x, residual=residual, want_stats=want_stats)
                    x, stats = out if want_stats else (out, None)
                elif isinstance(nxt, BatchNorm1d) and nxt.training and x.is_cuda:
                    x, stats = m(x, want_stats=True)
                else:
                    x, stats = m(x), None
                i += 1
            elif isinstance(m, (BatchNorm1d, LayerNorm)) and rows2d:
                relu = i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU
                x = m(x, fuse_relu=relu, stats=stats)
                stats = None
                i += 2 if relu else 1
            else:
                x = m(x)
                stats = None
                i += 1
        return (x, stats) if want_stats else x


class _SumOfEmbeddings(nn.Module):
    """out = sum_k Embedding_k(feature column k), xavier-initialised tables."""

    def _build(self, dims, emb_dim):
        tables = nn.ModuleList()
        for dim in dims:
            table = nn.Embedding(dim, emb_dim)
            nn.init.xavier_uniform_(table.weight.data)
            tables.append(table)
        return tables

    @staticmethod
    def _embed(tables, feats):
        total = 0
        for k in range(feats.shape[1]):
            total = total + tables[k](feats[:, k])
        return total


class AtomEncoder(_SumOfEmbeddings):
    def __init__(self, emb_dim):
        super().__init__()
        self.atom_embedding_list = self._build(get_atom_feature_dims(), emb_dim)

    def forward(self, x):
        return self._embed(self.atom_embedding_list, x)


class BondEncoder(_SumOfEmbeddings):
    def __init__(self, emb_dim):
        super().__init__()
        self.bond_embedding_list = self._build(get_bond_feature_dims(), emb_dim)

    def forward(self, edge_attr):
        return self._embed(self.bond_embedding_list, edge_attr)
