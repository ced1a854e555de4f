"""Model-level pre-activation residual layer of the reference's deep GENConv models, fused.

The 'res+' layer loop is written in the reference's MODEL files, not in gcn_lib
(examples/ogb/ogbn_arxiv/model.py:90-106, ogbn_products/model.py, ogbn_proteins/model.py:116-127, ogbg_*/model.py):

    h1 = self.norms[layer - 1](h); h2 = F.relu(h1); h2 = F.dropout(h2, p=self.dropout, training=self.training)
    h = self.gcns[layer](h2, edge_index) + h              # or: checkpoint(self.gcns[layer], h2, edge_index) + h

Written that way it costs, besides the convolution: a statistics pass, an apply pass, a ReLU pass, a dropout pass
(+ a stored mask) and a residual add, each a full read-modify-write of the (N, C) features.  ``res_plus_layer`` is the
same arithmetic as three launches around the aggregation:

    norm -> ReLU -> dropout   one apply pass (node_ops.pre_activation; statistics taken from the PREVIOUS layer's GEMM)
    GENConv                   aggregation with x + m fused, then the MLP whose last Linear adds bias AND h in its epilogue
                              and leaves the statistics of the new h for the next layer's BatchNorm

The example's loop body becomes (INTEGRATION.md shows the diff):

    h, stats = res_plus_layer(self.norms[layer - 1], self.gcns[layer], h, edge_index, p=self.dropout,
                              training=self.training, stats=stats, use_checkpoint=...)
"""
from __future__ import annotations

import torch
from torch.utils.checkpoint import checkpoint

from . import node_ops, ops

__all__ = ["res_plus_layer", "ResPlusLayer", "ComposedEdgeEmbedding"]


def _active_partition():
    import sys
    d = sys.modules.get(__package__ + ".dist")           # only a process that imported dist can have an active partition
    return d.active_partition() if d is not None else None


class _reentered:
    """dist.reentered without importing dist (torch.distributed) into single-GPU processes."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def _conv_res(conv, h2, edge_index, edge_attr, h, want_stats):
    if edge_attr is None:
        return conv(h2, edge_index, residual=h, want_stats=want_stats)
    return conv(h2, edge_index, edge_attr, residual=h, want_stats=want_stats)


def res_plus_layer(norm, conv, h, edge_index, edge_attr=None, p: float = 0.0, training: bool = True, stats=None,
                   want_stats: bool = True, use_checkpoint: bool = False):
    """``h + conv(dropout(relu(norm(h))), edge_index[, edge_attr])`` -> ``(h_new, stats_new)``.

    norm: ``gcn_lib.sparse.torch_nn.norm_layer`` module; conv: ``gcn_lib.sparse.torch_vertex.GENConv``.
    stats: what the previous call returned (BatchNorm statistics of ``h`` taken in the GEMM that produced it) or None.
    use_checkpoint: wrap the convolution in ``torch.utils.checkpoint`` as the reference does for deep stacks
    (ogbn_arxiv/model.py:101: the convolution is recomputed in the backward; ``h`` and ``h2`` stay alive).  True or
    "aggregation": the recomputation re-runs the node-wise part only -- the aggregation's outputs (two or three (N, C)
    arrays per layer; the reference checkpoints because ITS aggregation keeps (E, C) temporaries) are kept from the first
    pass (ops.AggregationStash); "full": everything is recomputed, aggregation included, exactly as
    torch.utils.checkpoint around the reference's GENConv would."""
    # h_skip is h: the second output routes the skip connection's gradient into the pre-activation's backward kernel
    h2, h_skip = node_ops.pre_activation(norm, h, p=p, training=training, stats=stats, skip=True)
    if use_checkpoint and torch.is_grad_enabled():
        if use_checkpoint not in (True, "aggregation", "full"):
            raise ValueError("use_checkpoint: False, True / 'aggregation', or 'full'")
        stash = None if use_checkpoint == "full" else ops.AggregationStash()
        # a node-partitioned run (dist.partitioned): the recomputation happens inside the backward pass -- on the
        # device's autograd thread, possibly after the ``with`` block was left -- and must exchange rows / statistics
        # exactly like the first pass: the context object travels with the closure
        part_ctx = _active_partition()
        opt_ctx = ops.captured_options()                   # the caller's ``ops.options`` block, for the same reason

        def run(h2_, h_, ea_):
            with _reentered(part_ctx), opt_ctx:
                return run_(h2_, h_, ea_)

        def run_(h2_, h_, ea_):
            # (edge features are an INPUT of the checkpoint, as in the reference's checkpoint(self.gcns[layer], h2,
            # edge_index, edge_emb): a tensor with history must not be reached through a closure, its graph would be
            # walked once per layer)
            # (h_ enters this function for the skip connection only: its gradient may leave as a view of the upstream one)
            if stash is None:
                with node_ops.residual_gradient_is_last_use():
                    out = _conv_res(conv, h2_, edge_index, ea_, h_, want_stats)
            else:
                with ops.stash_aggregation(stash, "replay" if torch.is_grad_enabled() else "record"), \
                        node_ops.residual_gradient_is_last_use():
                    out = _conv_res(conv, h2_, edge_index, ea_, h_, want_stats)
            return out if want_stats else (out, None)
        # the second output (statistics) is not differentiable; checkpoint hands it through
        hn, st = checkpoint(run, h2, h_skip, edge_attr, use_reentrant=True)
    else:
        out = _conv_res(conv, h2, edge_index, edge_attr, h_skip, want_stats)
        hn, st = out if want_stats else (out, None)
    return hn, st


class ResPlusLayer(torch.nn.Module):
    """``res_plus_layer`` as a module owning ``norm`` and ``conv`` (attribute names as in the reference's ModuleLists are
    the caller's business; this class is for code that builds the stack itself)."""

    def __init__(self, norm, conv, dropout: float = 0.0, use_checkpoint: bool = False):
        super().__init__()
        self.norm, self.conv = norm, conv
        self.dropout, self.use_checkpoint = dropout, use_checkpoint

    def forward(self, h, edge_index, edge_attr=None, stats=None):
        return res_plus_layer(self.norm, self.conv, h, edge_index, edge_attr, p=self.dropout, training=self.training,
                              stats=stats, use_checkpoint=self.use_checkpoint)


class _ComposeEncoder(torch.autograd.Function):
    """(W', b') = (W_l We, W_l b_e + b_l) and its backward, one library launch each way (csrc/enc_compose.hip) instead of
    matmul + addmv forward and four small GEMM / GEMV launches backward per coupling function."""

    @staticmethod
    def forward(ctx, layer_w, layer_b, enc_w, enc_b):
        from . import _lib
        lib = _lib.load()
        dev = layer_w.device
        lw, ew = layer_w.detach().contiguous(), enc_w.detach().contiguous()
        lb = None if layer_b is None else layer_b.detach().contiguous()
        eb = None if enc_b is None else enc_b.detach().contiguous()
        C, H = lw.shape
        F = ew.size(1)
        # a function that runs twice per step on the same parameters (the reversible layers' forward and the grad-enabled
        # evaluation inside their backward; a checkpointed layer): the second pass takes the first one's (W', b') from
        # the pass's stash (ops.AggregationStash, record / replay) -- no launch; the graph is recorded all the same
        from . import ops
        stash = ops._active_stash()
        key = ("compose", lw.data_ptr(), ew.data_ptr(), C, H, F)
        kept = stash.extra.get(key) if (stash is not None and stash.mode == "replay") else None
        if kept is not None:
            out_w, out_b = kept[0].detach(), (None if kept[1] is None else kept[1].detach())
        else:
            out_w = torch.empty(C, F, device=dev, dtype=torch.float32)
            out_b = torch.empty(C, device=dev, dtype=torch.float32) if (lb is not None or eb is not None) else None
            with _lib.device_ctx(dev):
                _lib.check(lib.dgcn_enc_compose_fwd_f32(lw.data_ptr(), _lib.ptr(lb), ew.data_ptr(), _lib.ptr(eb), C, H, F,
                                                        out_w.data_ptr(), _lib.ptr(out_b),
                                                        _lib.current_stream_handle(dev)), "dgcn_enc_compose_fwd_f32")
            if stash is not None and stash.mode == "record":
                stash.extra[key] = (out_w, out_b)
        ctx.save_for_backward(lw, ew, eb)
        ctx.has_lb = lb is not None
        if out_b is None:
            ctx.mark_non_differentiable()
            return out_w, None
        return out_w, out_b

    @staticmethod
    def backward(ctx, gw, gb):
        from . import _lib
        lib = _lib.load()
        lw, ew, eb = ctx.saved_tensors
        dev = lw.device
        C, H = lw.shape
        F = ew.size(1)
        gw = torch.zeros(C, F, device=dev, dtype=torch.float32) if gw is None else gw.float().contiguous()
        gb = None if gb is None else gb.float().contiguous()
        need_l, need_e, need_eb = ctx.needs_input_grad[0], ctx.needs_input_grad[2], ctx.needs_input_grad[3] and eb is not None
        d_l = torch.empty(C, H, device=dev, dtype=torch.float32) if need_l else None
        d_e = torch.empty(H, F, device=dev, dtype=torch.float32) if (need_e or need_eb) else None
        d_eb = torch.empty(H, device=dev, dtype=torch.float32) if (need_eb and gb is not None) else None
        with _lib.device_ctx(dev):
            _lib.check(lib.dgcn_enc_compose_bwd_f32(lw.data_ptr(), ew.data_ptr(), _lib.ptr(eb), gw.data_ptr(), _lib.ptr(gb), C, H, F,
                                                    _lib.ptr(d_l), _lib.ptr(d_e), _lib.ptr(d_eb),
                                                    _lib.current_stream_handle(dev)), "dgcn_enc_compose_bwd_f32")
        d_lb = gb if (ctx.has_lb and ctx.needs_input_grad[1]) else None
        return d_l, d_lb, (d_e if need_e else None), d_eb


def _compose_on_device(layer_w, layer_b, enc_w, enc_b) -> bool:
    ts = [t for t in (layer_w, layer_b, enc_w, enc_b) if t is not None]
    return (all(t.is_cuda and t.dtype == torch.float32 for t in ts) and enc_w.size(1) <= 16
            and not torch.is_autocast_enabled())


class ComposedEdgeEmbedding:
    """``edge_encoder(edge_attr)`` [repeated ``repeat`` times along the feature axis] WITHOUT building it.

    The reference's models with edge features apply two Linear maps in a row to the 8 raw edge features, with nothing in
    between: the model-level ``edge_emb = self.edge_encoder(edge_attr)`` -- Linear(8 -> hidden), repeated per group in
    the reversible models (examples/ogb_eff/ogbn_proteins/model_rev.py:98-100; ogbn_proteins/model.py:103-107) -- and
    every GENConv's own ``edge_encoder`` = Linear(hidden -> C) (gcn_lib/sparse/torch_vertex.py:56-66).  Their
    composition is ONE Linear(8 -> C) per layer: ``W' = W_l We``, ``b' = W_l be + b_l`` (two tiny products, written
    with autograd ops so that the gradients of both layers follow).  A GENConv that receives this object instead of the
    (E, hidden) tensor evaluates ``W' f_e + b'`` per edge inside the aggregation kernels from the 32 raw bytes
    (dgcn_gen_aggr_enc_{fwd,bwd}_f32): neither the (E, hidden) embedding, nor the per-layer (E, hidden) x (hidden, C)
    GEMM, nor the (E, hidden) gradient accumulated over the layers exist any more.  The model's lines become

        edge_emb = ComposedEdgeEmbedding(self.edge_encoder, edge_attr, repeat=self.group)

    Same function of the same parameters (the two roundings of ``(A We) W_l`` become one of ``A (We W_l)``).  Consumers
    that cannot fuse (another convolution, a non-Linear encoder, CPU tensors) call ``materialize()``."""

    def __init__(self, encoder: torch.nn.Linear, edge_attr: torch.Tensor, repeat: int = 1):
        if not isinstance(encoder, torch.nn.Linear):
            raise TypeError("ComposedEdgeEmbedding: the model-level edge encoder must be an nn.Linear")
        if edge_attr.dim() != 2 or edge_attr.size(1) != encoder.in_features:
            raise ValueError("edge_attr must be (E, encoder.in_features)")
        self.encoder = encoder
        # (the same tensor object for every group view / layer: per-tensor caches of the kernels' host side hit)
        # Raw features that require a gradient (learned / upstream-computed edge features): the per-edge kernels have no
        # gradient w.r.t. them and the reversible wrapper routes gradients of tensor arguments only -- refuse instead of
        # dropping that gradient silently (ops._GenAggregate raises for the same case)
        if edge_attr.requires_grad:
            raise ValueError("ComposedEdgeEmbedding: edge_attr requires grad; the composed per-edge encoder has no gradient "
                             "w.r.t. the raw edge features -- pass edge_attr.detach() if that gradient is not needed, or "
                             "keep the model's own edge_emb = edge_encoder(edge_attr)")
        self.raw = edge_attr
        self.repeat = int(repeat)
        self._full = None

    # -- what the reversible wrapper / the coupling need from an argument --------------------------------------------
    def parameters(self):
        return [p for p in self.encoder.parameters() if p.requires_grad]

    def group_view(self, i: int, groups: int):
        """Chunk ``i`` of ``groups`` along the feature axis: every chunk of a repeated embedding is the embedding."""
        if groups != self.repeat:
            raise ValueError(f"embedding repeated {self.repeat}x cannot be split into {groups} groups")
        return self if self.repeat == 1 else ComposedEdgeEmbedding(self.encoder, self.raw, 1)

    @property
    def hidden(self) -> int:
        return self.encoder.out_features

    def composed(self, layer_encoder: torch.nn.Linear):
        """(W', b') of ``layer_encoder(encoder(.))`` for a per-layer Linear(hidden -> C)."""
        if self.repeat != 1:
            raise ValueError("compose a group view, not the repeated embedding")
        if _compose_on_device(layer_encoder.weight, layer_encoder.bias, self.encoder.weight, self.encoder.bias):
            return _ComposeEncoder.apply(layer_encoder.weight, layer_encoder.bias, self.encoder.weight, self.encoder.bias)
        w = layer_encoder.weight @ self.encoder.weight                               # (C, 8)
        if self.encoder.bias is None:
            return w, layer_encoder.bias
        if layer_encoder.bias is None:
            return w, layer_encoder.weight @ self.encoder.bias
        return w, torch.addmv(layer_encoder.bias, layer_encoder.weight, self.encoder.bias)    # one op: b_l + W_l b_e

    def materialize(self) -> torch.Tensor:
        e = self.encoder(self.raw)
        return e if self.repeat == 1 else torch.cat([e] * self.repeat, dim=-1)
