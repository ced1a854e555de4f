"""Fast layer loops for the reference's UNCHANGED model files (``install(fuse_models=True)``).

The deep GENConv models write their layer loop in the MODEL file, not in gcn_lib:

* examples/ogb/ogbn_arxiv/model.py:88-106 and ogbn_products/model.py (DeeperGCN 'res+'):
  ``norm -> relu -> dropout -> [checkpoint](GENConv) + h`` per layer -- six full read-modify-write passes over the
  (N, C) features around every convolution.  ``blocks.res_plus_layer`` is the same arithmetic in three launches;
* examples/ogb_eff/ogbn_proteins/model_rev.py:98-99 (RevGCN): ``edge_emb = cat([edge_encoder(edge_attr)] * group)`` --
  an (E, hidden) embedding that every GENConv multiplies by its own Linear(hidden -> C) again.
  ``blocks.ComposedEdgeEmbedding`` composes the two Linear maps and evaluates them per edge inside the aggregation.

Both rewrites are the same function of the same parameters (same ``state_dict``), so they can be applied to the
reference's classes from outside: ``fuse_model_class(cls)`` swaps ``cls.forward`` for a version that takes the fused
route when the instance qualifies (device tensors, this package's GENConv / norm layers, the 'res+' block, ...) and calls
the original ``forward`` otherwise.  ``install(fuse_models=True)`` applies it to every class named ``DeeperGCN`` /
``RevGCN`` of a module named ``model`` / ``model_rev`` as the example scripts import it (a ``sys.meta_path`` hook that
post-processes the freshly executed module; the hook is process-wide: ANY top-level module of that name that defines
such a class is post-processed, ``install(fuse_models=False)`` removes it); ``fuse_model(instance)`` does it for one
object.

Checkpointing: the reference wraps the convolutions of deep softmax / power stacks in ``torch.utils.checkpoint``
(ogbn_arxiv/model.py:38-40,98-100).  ``CHECKPOINT = "full"`` (default) keeps that memory behaviour to the letter -- the
backward recomputes the whole convolution, aggregation included; ``"aggregation"`` keeps the aggregation's (N, C)
results of the first pass (2-3 node-sized arrays per layer) and recomputes the node-wise part only.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import inspect
import sys

import torch
import torch.nn.functional as F

from . import blocks, node_ops

__all__ = ["fuse_model", "fuse_model_class", "enable_import_hook", "disable_import_hook", "CHECKPOINT"]

CHECKPOINT = "full"             # "full" | "aggregation"  (see the module docstring)
_ORIG = "_dgcn_original_forward"
_MODEL_MODULES = ("model", "model_rev")
_CLASS_NAMES = ("DeeperGCN", "RevGCN")


# ---------------------------------------------------------------------------------------------------------------------
# DeeperGCN 'res+'  (examples/ogb/ogbn_arxiv/model.py:84-112, ogbn_products/model.py)
# ---------------------------------------------------------------------------------------------------------------------
def _is_gen_stack(model) -> bool:
    from .gcn_lib.sparse.torch_vertex import GENConv
    gcns, norms = getattr(model, "gcns", None), getattr(model, "norms", None)
    if gcns is None or norms is None or len(gcns) != len(norms) or len(gcns) < 1:
        return False
    if not all(isinstance(g, GENConv) and not getattr(g, "encode_edge", False) for g in gcns):
        return False
    return all(isinstance(n, (node_ops.BatchNorm1d, node_ops.LayerNorm)) for n in norms)


def _deepergcn_qualifies(model, x, edge_index) -> bool:
    return (getattr(model, "block", None) == "res+" and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 2
            and isinstance(edge_index, torch.Tensor) and hasattr(model, "node_features_encoder")
            and hasattr(model, "node_pred_linear") and hasattr(model, "num_layers") and _is_gen_stack(model)
            and len(model.gcns) == model.num_layers)


def _deepergcn_forward(self, x, edge_index):
    """ogbn_arxiv/model.py:84-140 for block == 'res+', the loop body through ``blocks.res_plus_layer``."""
    if not _deepergcn_qualifies(self, x, edge_index):
        return getattr(type(self), _ORIG)(self, x, edge_index)
    h = self.node_features_encoder(x)
    h, stats = self.gcns[0](h, edge_index, want_stats=True)
    ckpt = bool(getattr(self, "checkpoint_grad", False))
    ckp_k = getattr(self, "ckp_k", 0) or 1
    mode = "full" if CHECKPOINT == "full" else True
    for layer in range(1, self.num_layers):
        h, stats = blocks.res_plus_layer(self.norms[layer - 1], self.gcns[layer], h, edge_index, p=self.dropout,
                                         training=self.training, stats=stats,
                                         use_checkpoint=(ckpt and layer % ckp_k != 0) and mode)
    h = node_ops.pre_activation(self.norms[self.num_layers - 1], h, p=self.dropout, training=self.training, stats=stats)
    return torch.log_softmax(self.node_pred_linear(h), dim=-1)


# ---------------------------------------------------------------------------------------------------------------------
# RevGCN  (examples/ogb_eff/ogbn_proteins/model_rev.py:85-112)
# ---------------------------------------------------------------------------------------------------------------------
def _revgcn_qualifies(model, x, edge_attr) -> bool:
    from .eff_gcn_modules.rev import memgcn, rev_layer
    from .gcn_lib.sparse.torch_vertex import GENConv
    enc = getattr(model, "edge_encoder", None)
    if not all(hasattr(model, a) for a in ("node_features", "node_features_encoder", "last_norm", "node_pred_linear",
                                           "use_one_hot_encoding", "num_layers", "dropout")):
        return False
    if not (isinstance(enc, torch.nn.Linear) and isinstance(edge_attr, torch.Tensor) and edge_attr.is_cuda
            and edge_attr.dim() == 2 and edge_attr.is_floating_point() and not edge_attr.requires_grad
            and edge_attr.size(1) == enc.in_features and hasattr(model, "group") and hasattr(model, "gcns")):
        return False
    for wrapper in model.gcns:
        coupling = getattr(wrapper, "_fn", None)
        if not isinstance(wrapper, memgcn.InvertibleModuleWrapper) or not isinstance(coupling, memgcn.GroupAdditiveCoupling):
            return False
        for fm in coupling.Fms:
            gcn = getattr(fm, "gcn", None)
            if not (isinstance(fm, rev_layer.GENBlock) and isinstance(gcn, GENConv) and gcn.encode_edge
                    and isinstance(getattr(gcn, "edge_encoder", None), torch.nn.Linear)
                    and gcn.edge_encoder.in_features == enc.out_features):
                return False
    return True


def _revgcn_forward(self, x, node_index, edge_index, edge_attr, epoch=-1):
    """model_rev.py:85-112 with the two edge-embedding lines (:98-99) replaced by the composed form."""
    if not _revgcn_qualifies(self, x, edge_attr):
        return getattr(type(self), _ORIG)(self, x, node_index, edge_index, edge_attr, epoch)
    node_features_1st = self.node_features[node_index]
    if self.use_one_hot_encoding:
        node_features = torch.cat((node_features_1st, self.node_one_hot_encoder(x)), dim=1)
    else:
        node_features = node_features_1st
    h = self.node_features_encoder(node_features)
    edge_emb = blocks.ComposedEdgeEmbedding(self.edge_encoder, edge_attr, repeat=self.group)
    m = torch.zeros_like(h).bernoulli_(1 - self.dropout)
    mask = m.requires_grad_(False) / (1 - self.dropout)
    for layer in range(self.num_layers):
        h = self.gcns[layer](h, edge_index, mask, edge_emb)
    h = F.relu(self.last_norm(h))
    h = F.dropout(h, p=self.dropout, training=self.training)
    return self.node_pred_linear(h)


# ---------------------------------------------------------------------------------------------------------------------
def _replacement_for(cls):
    """The fused forward that fits ``cls.forward``'s signature, or None."""
    try:
        params = list(inspect.signature(cls.forward).parameters)
    except (TypeError, ValueError):
        return None
    if params == ["self", "x", "edge_index"]:
        return _deepergcn_forward
    if (params[:5] == ["self", "x", "node_index", "edge_index", "edge_attr"] and params[5:] in ([], ["epoch"])
            and any(b.__name__ == "RevGCN" for b in cls.__mro__)):
        return _revgcn_forward
    return None


def fuse_model_class(cls) -> bool:
    """Swap ``cls.forward`` for the fused route (falls back to the original per call).  Idempotent.  Returns whether a
    replacement exists for this class's ``forward`` signature."""
    if not (isinstance(cls, type) and issubclass(cls, torch.nn.Module)):
        raise TypeError("fuse_model_class expects an nn.Module subclass")
    if _ORIG in cls.__dict__ or cls.forward in (_deepergcn_forward, _revgcn_forward):
        # already fused -- itself, or through a fused base class whose replacement it inherits (wrapping that again would
        # record the replacement as the "original" and the non-qualifying fall-back would call itself for ever)
        return True
    repl = _replacement_for(cls)
    if repl is None:
        return False
    setattr(cls, _ORIG, cls.forward)
    cls.forward = repl
    # said once per class: install() routes a model file's layer loop differently than the file reads
    import logging
    logging.getLogger("deep_gcns_torch_amd").info(
        "%s.%s.forward now runs through deep_gcns_torch_amd.fuse (same parameters, state_dict and values; "
        "install(fuse_models=False) or fuse.unfuse_model_class restores the file's own loop)", cls.__module__, cls.__name__)
    return True


def unfuse_model_class(cls) -> None:
    if _ORIG in cls.__dict__:
        cls.forward = cls.__dict__[_ORIG]
        delattr(cls, _ORIG)


def fuse_model(model: torch.nn.Module) -> torch.nn.Module:
    """Fuse ONE instance (its class is left alone): ``model = fuse_model(DeeperGCN(args))``."""
    cls = type(model)
    if _ORIG in cls.__dict__ or cls.forward in (_deepergcn_forward, _revgcn_forward):
        return model
    repl = _replacement_for(cls)
    if repl is None:
        raise TypeError(f"no fused layer loop for {cls.__name__}.forward{inspect.signature(cls.forward)}")
    sub = type(cls.__name__, (cls,), {_ORIG: cls.forward, "forward": repl, "__module__": cls.__module__})
    model.__class__ = sub
    return model


class _PostExecLoader(importlib.abc.Loader):
    def __init__(self, inner):
        self._inner = inner

    def create_module(self, spec):
        return self._inner.create_module(spec)

    def exec_module(self, module):
        self._inner.exec_module(module)
        for name in _CLASS_NAMES:
            cls = module.__dict__.get(name)
            if isinstance(cls, type) and issubclass(cls, torch.nn.Module) and cls.__module__ == module.__name__:
                fuse_model_class(cls)

    def __getattr__(self, item):                       # get_source / get_filename / ... of the real loader
        return getattr(self._inner, item)


class _ModelFinder(importlib.abc.MetaPathFinder):
    """Finds ``model`` / ``model_rev`` like the default machinery does, then fuses the DeeperGCN / RevGCN class the
    module defines.  Any other import is not touched."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname not in _MODEL_MODULES:
            return None
        spec = importlib.machinery.PathFinder.find_spec(fullname, path, target)
        if spec is None or spec.loader is None:
            return None
        spec.loader = _PostExecLoader(spec.loader)
        return spec


_FINDER = _ModelFinder()


def enable_import_hook() -> None:
    if _FINDER not in sys.meta_path:
        sys.meta_path.insert(0, _FINDER)
    for name in _MODEL_MODULES:                        # already imported: fuse in place
        mod = sys.modules.get(name)
        if mod is not None:
            for cname in _CLASS_NAMES:
                cls = mod.__dict__.get(cname)
                if isinstance(cls, type) and issubclass(cls, torch.nn.Module):
                    fuse_model_class(cls)


def disable_import_hook() -> None:
    if _FINDER in sys.meta_path:
        sys.meta_path.remove(_FINDER)
