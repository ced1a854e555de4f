"""`scatter_` -- the reference's thin wrapper over torch_scatter (utils/pyg_util.py:4-35),
backed by the fused libdgcn aggregation kernel instead of atomics."""
import torch

from .. import ops
from ..graph import scatter_graph_of

__all__ = ["scatter_"]


def scatter_(name, src, index, dim=0, dim_size=None):
    """Aggregate rows of ``src`` (E, C) into ``dim_size`` rows by ``index`` with
    ``name`` in {add, mean, min, max}.  Empty rows give 0; for max (min) every result below
    -10000 (above 10000) is reset to 0, as the reference does (:30-33)."""
    assert name in ["add", "mean", "min", "max"]
    if dim not in (0, -2) or src.dim() != 2:
        raise NotImplementedError("scatter_ supports (E, C) tensors along dim 0")
    n = int(dim_size) if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
    # rows of `src` are already per-edge values: gather them through the identity "source" map; the structure
    # depends on `index` only and is cached per live index tensor (no sort / host sync per call)
    g = scatter_graph_of(index, n)
    if name == "min":
        out = -ops.gen_aggregate(-src, g, aggr="max", relu_eps=False)
        return torch.where(out > 10000, torch.zeros_like(out), out)
    out = ops.gen_aggregate(src, g, aggr=name, relu_eps=False)
    if name == "max":
        out = torch.where(out < -10000, torch.zeros_like(out), out)
    return out
