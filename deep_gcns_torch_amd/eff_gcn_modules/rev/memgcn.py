"""Group additive coupling: public surface of the reference's eff_gcn_modules/rev/memgcn.py (:8-52), plus the fused
backward step used by ``gcn_revop.InvertibleCheckpointFunction``.

    forward :  y_0 = x_0 + F_0(x_1 + ... + x_{g-1}),   y_i = x_i + F_i(y_{i-1})
    inverse :  x_i = y_i - F_i(y_{i-1})  (i = g-1 .. 1),   x_0 = y_0 - F_0(x_1 + ... + x_{g-1})
"""
import torch

from ... import node_ops, ops
from .gcn_revop import InvertibleModuleWrapper  # noqa: F401  (model_rev.py reaches it as memgcn.InvertibleModuleWrapper)

__all__ = ["GroupAdditiveCoupling", "InvertibleModuleWrapper"]



def _sum_of(parts):
    """``sum(parts)`` without the launch for ``0 + parts[0]`` when there is one part (group = 2: the input of F_0 is the
    other group itself, the same values bit for bit).  For passes that record no graph only."""
    return parts[0] if len(parts) == 1 else sum(parts)


FOLD_COUPLING = True      # x_i +/- F_i(.) in the epilogue of F_i's last Linear (False: an elementwise launch; A/B)


def _takes_residual(fm) -> bool:
    """Whether the wrapped block understands the ``residual`` extension: this package's rev_layer.BasicBlock and its
    subclasses only (a foreign block with a parameter of that name must not be handed a CouplingResidual)."""
    from . import rev_layer
    return isinstance(fm, rev_layer.BasicBlock)


def _offer(fm, res, out, negate):
    """The coupling's add / subtract offered to F_i's last Linear (node_ops.CouplingResidual), or None."""
    if not (FOLD_COUPLING and res.is_cuda and res.dtype == torch.float32 and res.dim() == 2 and _takes_residual(fm)):
        return None
    return node_ops.CouplingResidual(res, out, negate)


class GroupAdditiveCoupling(torch.nn.Module):
    def __init__(self, Fms, split_dim=-1, group=2):
        super().__init__()
        self.Fms = Fms
        self.split_dim = split_dim
        self.group = group

    def _arg_chunks(self, args):
        # (an argument object with group_view -- blocks.ComposedEdgeEmbedding -- hands out its own group views)
        per_arg = [[a.group_view(i, self.group) for i in range(self.group)] if hasattr(a, "group_view")
                   else torch.chunk(a, self.group, dim=self.split_dim) for a in args]
        return list(zip(*per_arg))                      # [group][arg]

    def new_stashes(self, node_sized_only: bool = True):
        """One ops.AggregationStash per coupling function (see ``forward(..., _stashes)``)."""
        return [ops.AggregationStash(node_sized_only=node_sized_only) for _ in range(self.group)]

    def forward(self, x, edge_index, *args, _stashes=None):
        """_stashes (the reversible wrapper's no_grad forward only): the aggregation launches of F_i record their
        (N, C) outputs and what their backward needs there; ``fused_backward`` hands them out again instead of
        launching -- F_i sees the same input in both passes (y_{i-1} exactly; F_0 the rebuilt sum of the other groups,
        equal up to the rounding of the reconstruction)."""
        xs = torch.chunk(x, self.group, dim=self.split_dim)
        extra = self._arg_chunks(args)
        if not torch.is_grad_enabled() and x.dim() == 2 and self.split_dim in (-1, 1):
            # (no graph is recorded here: with one other group its view serves as F_0's input, no ``0 + x`` launch.  Where a
            #  graph IS recorded the fresh tensor is needed: a view of a buffer that is written to afterwards -- the
            #  column blocks of x in fused_backward -- fails autograd's version check of the saved input)
            y_in = _sum_of(xs[1:])
            # the reversible wrapper's forward (no graph): every y_i is written straight into its columns of the result
            y = torch.empty_like(x)
            for i, yv in enumerate(torch.chunk(y, self.group, dim=1)):
                cr = _offer(self.Fms[i], xs[i], yv, False)
                kw = {} if cr is None else {"residual": cr}
                if _stashes is None:
                    f = self.Fms[i](y_in, edge_index, *extra[i], **kw)
                else:
                    with ops.stash_aggregation(_stashes[i], "record"):
                        f = self.Fms[i](y_in, edge_index, *extra[i], **kw)
                # (folded: F_i's last Linear wrote x_i + F_i into the column block itself)
                y_in = yv if (cr is not None and cr.used) else torch.add(xs[i], f, out=yv)
            return y
        ys = []
        y_in = sum(xs[1:])
        for i in range(self.group):
            y_in = xs[i] + self.Fms[i](y_in, edge_index, *extra[i])
            ys.append(y_in)
        return torch.cat(ys, dim=self.split_dim)

    def inverse(self, y, edge_index, *args):
        ys = torch.chunk(y, self.group, dim=self.split_dim)
        extra = self._arg_chunks(args)
        xs = [None] * self.group
        for i in range(self.group - 1, -1, -1):
            y_in = ys[i - 1] if i != 0 else sum(xs[1:])
            xs[i] = ys[i] - self.Fms[i](y_in, edge_index, *extra[i])
        return torch.cat(xs, dim=self.split_dim)

    # ------------------------------------------------------------------------------------------------------------
    def arg_sink_ok(self, t) -> bool:
        """Whether ``make_arg_sink`` can serve ``t`` (decided in the wrapper's forward: the fused backward is only taken
        when every shared argument qualifies)."""
        return t.dim() == 2 and self.split_dim in (-1, 1) and t.size(-1) % self.group == 0

    def make_arg_sink(self, t):
        """Running gradient sum of a tensor argument every layer receives: one CONTIGUOUS (rows, width / group) block
        per group (GENConv's ``dz @ W`` is accumulated into it by a plain GEMM with beta = 1, no strided output)."""
        width = t.size(self.split_dim)
        assert t.dim() == 2 and self.split_dim in (-1, 1) and width % self.group == 0
        return torch.zeros(self.group, t.size(0), width // self.group, device=t.device, dtype=torch.float32)

    def finish_arg_sink(self, buf, t):
        """(group, rows, w) -> the layout of the argument (rows, group * w)."""
        return buf.permute(1, 0, 2).reshape(t.shape).to(t.dtype)

    def fused_backward(self, y, grad_y, edge_index, args, weights, arg_sinks, sink_ctx, stashes=None):
        """Inverse and gradient of one coupling step from ONE grad-enabled evaluation of every F_i.

        ``F_i`` sees the same input in ``inverse`` and in the recompute of ``forward`` (y_{i-1}, or the sum of the
        other groups for i = 0), so its output is computed once: ``x_i = y_i - F_i(.)`` rebuilds the input and the
        recorded graph is differentiated with the total gradient that reaches ``y_i``.
        Returns (x, grad_x, gradients of ``weights`` in order).  Gradients of the extra tensor arguments are ADDED
        into ``arg_sinks`` (one ``make_arg_sink`` buffer per floating argument that requires grad);
        ``sink_ctx(arg_leaf, buffer_block)`` lets GENConv add its edge-feature gradient in place."""
        g = self.group
        dim = self.split_dim
        ys = torch.chunk(y, g, dim=dim)
        gys = torch.chunk(grad_y, g, dim=dim)
        float_args = [a for a in args if isinstance(a, torch.Tensor) and a.requires_grad and a.is_floating_point()]
        assert len(float_args) == len(arg_sinks)
        sink_of = {id(a): s for a, s in zip(float_args, arg_sinks)}
        wpos = {id(w): k for k, w in enumerate(weights)}
        wgrads = [None] * len(weights)
        xs = [None] * g
        gx = [None] * g
        own_version = [False] * g
        flat = y.dim() == 2 and dim in (-1, 1) and not torch.is_grad_enabled()
        if flat:                                       # x and grad_x are assembled in place, column block by column block
            x_buf, gx_buf = torch.empty_like(y), torch.empty_like(grad_y)
            xv, gxv = torch.chunk(x_buf, g, dim=1), torch.chunk(gx_buf, g, dim=1)
        carry = None                                   # gradient flowing into y_{i-1} from F_i's input
        for i in range(g - 1, -1, -1):
            Fm = self.Fms[i]
            with torch.enable_grad():
                # F_0's input is the sum of the other groups: with one other group whose x_1 came out of the folded
                # epilogue -- a tensor with its own version counter over that column block of x -- it is that tensor
                # itself; a plain view of x_buf would fail autograd's version check once x_0 is written next to it
                src = ys[i - 1] if i != 0 else (xs[1] if (g == 2 and own_version[1]) else sum(xs[1:]))
                leaf = src.detach().requires_grad_(True)
                leaves, views = [], []
                call_args = []
                for a in args:
                    if isinstance(a, torch.Tensor):
                        c = torch.chunk(a.detach(), g, dim=dim)[i]
                        if id(a) in sink_of:
                            c.requires_grad_(True)
                            leaves.append(c)
                            s = sink_of[id(a)]
                            views.append(None if s is None else s[i])
                        call_args.append(c)
                    elif hasattr(a, "group_view"):
                        call_args.append(a.group_view(i, g))
                    else:
                        call_args.append(a)
                ctxs = [sink_ctx(c, v) for c, v in zip(leaves, views) if v is not None]
                if stashes is not None and stashes[i].items:
                    ctxs.append(ops.stash_aggregation(stashes[i], "replay"))
                for cm in ctxs:
                    cm.__enter__()
                try:
                    cr = _offer(Fm, ys[i], xv[i], True) if flat else None
                    out = Fm(leaf, edge_index, *call_args, **({} if cr is None else {"residual": cr}))
                    if flat:
                        with torch.no_grad():
                            if cr is not None and cr.used:
                                # ``out`` holds y_i - F_i (the values of x_i, in place in x's column block) and the
                                # graph of F_i: differentiated below with the gradient that reaches F_i's output
                                xs[i], own_version[i] = out.detach(), True
                            else:
                                xs[i] = torch.sub(ys[i], out, out=xv[i])
                            # (the last group's gradient is grad_y's own column block until F_0's input gradient
                            #  joins it below: no copy into the buffer first)
                            total = gys[i] if carry is None else torch.add(gys[i], carry, out=gxv[i])
                    else:
                        xs[i] = ys[i] - out.detach()
                        total = gys[i] if carry is None else gys[i] + carry
                    gx[i] = total
                    params = [p for p in Fm.parameters() if p.requires_grad]
                    for a in args:                     # parameters behind an argument object (see _arg_chunks)
                        if not isinstance(a, torch.Tensor) and hasattr(a, "group_view"):
                            params += [p for p in a.parameters() if all(p is not q for q in params)]
                    grads = torch.autograd.grad(out, [leaf] + leaves + params, total, allow_unused=True)
                finally:
                    for cm in reversed(ctxs):
                        cm.__exit__(None, None, None)
            carry = grads[0]
            for c, v, gr in zip(leaves, views, grads[1:1 + len(leaves)]):
                if gr is not None and v is not None:
                    v.add_(gr)                         # anything that did not go through the in-place sink
            for p, gr in zip(params, grads[1 + len(leaves):]):
                k = wpos.get(id(p))
                if k is not None and gr is not None:
                    wgrads[k] = gr if wgrads[k] is None else wgrads[k] + gr
        # the gradient into F_0's input (the sum of the other groups) goes to every x_i, i >= 1
        if carry is not None:
            for i in range(1, g):
                if not flat:
                    gx[i] = gx[i] + carry
                elif gx[i] is gys[i]:
                    gx[i] = torch.add(gys[i], carry.detach(), out=gxv[i])
                else:
                    gx[i] = gx[i].add_(carry.detach())
        if flat:
            for i in range(g):
                if gx[i] is gys[i]:                    # (one group, or no gradient into F_0's input)
                    gxv[i].copy_(gys[i])
        x = x_buf if flat else torch.cat(xs, dim=dim)
        grad_x = gx_buf if flat else torch.cat(gx, dim=dim)
        wgrads = [torch.zeros_like(w) if gr is None else gr for w, gr in zip(weights, wgrads)]
        return x, grad_x, wgrads
