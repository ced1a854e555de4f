"""Node-wise Linear layers around the hot path.

``TallLinear`` is an ``nn.Linear`` (same parameters, same ``state_dict`` keys, ``isinstance`` still holds) for inputs
whose rows are graph nodes or edges (10^4 .. 10^6 of them, 16 .. 256 columns):

* forward and input gradient run on ``node_ops.rows_linear`` (csrc/rows_linear.hip: fp32-faithful on the bf16 matrix
  pipe, bias / 'res+' residual / the next BatchNorm's statistics / the bias gradient folded into the same sweep) when
  the shape is one the kernel takes, else on the library GEMM;
* the weight gradient ``g^T x`` -- a GEMM with a 10^5..10^6-long reduction and a 128x128 result -- is a batched split-K
  product plus a fixed-order sum.  hipBLASLt runs the un-split shape on a handful of CUs (measured 412 us for
  169,343 x 128 x 128, ~13 TFLOP/s); the split form uses the whole chip.
"""
import torch
from torch import nn

_MIN_ROWS = 4096     # from here on the un-split g^T x runs on a handful of CUs (measured 85 us at 13,253 x 112 x 224)


def splitk_xt_g(g2: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """g2^T @ x2 for (R, M), (R, C) with R >> M, C."""
    R = g2.size(0)
    S = 1
    while S < 128 and R // (S * 2) >= 256:
        S *= 2
    if S == 1:
        return g2.t() @ x2
    Rp = (R // S) * S                     # ragged tail handled separately
    part = torch.bmm(g2[:Rp].view(S, Rp // S, -1).transpose(1, 2), x2[:Rp].view(S, Rp // S, -1)).sum(0)
    if Rp != R:
        part = part + g2[Rp:].t() @ x2[Rp:]
    return part


class _TallLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        g2 = g.reshape(-1, g.size(-1))
        if ctx.needs_input_grad[0]:
            gx = g @ weight
        if ctx.needs_input_grad[1]:
            from . import node_ops
            gw = node_ops.rows_tn(g2, x.reshape(-1, x.size(-1)))     # matrix-pipe kernel, or the split-K library form
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum(0)
        return gx, gw, gb


ROWS_KERNEL = True      # False: library GEMMs everywhere (A/B measurements)


class TallLinear(nn.Linear):
    def forward(self, x, residual=None, want_stats: bool = False):
        """``residual`` (extension): added to the result in the GEMM epilogue -- the 'res+' skip connection
        (examples/ogb/ogbn_arxiv/model.py:104).  ``want_stats`` (extension): also return the (parts, 2, C) partial sums
        of the result that ``BatchNorm1d(stats=...)`` takes, or None when the row kernel did not run."""
        from . import node_ops
        if ROWS_KERNEL and node_ops.rows_linear_supported(x, self.weight):
            return node_ops.rows_linear(x, self.weight, self.bias, residual, want_stats)
        if ROWS_KERNEL and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and not torch.is_autocast_enabled():
            node_ops.warn_library_gemm("a node Linear", x.size(0), self.in_features, self.out_features,
                                       "it takes 16 <= in-features <= 256 and out-features <= 256, multiples of 4")
        if x.is_cuda and x.dim() == 2 and x.size(0) >= _MIN_ROWS and torch.is_grad_enabled() \
                and x.dtype == self.weight.dtype and not torch.is_autocast_enabled():
            y = _TallLinearFn.apply(x, self.weight, self.bias)
        else:
            y = super().forward(x)
        if residual is not None and not isinstance(residual, node_ops.CouplingResidual):   # (left to the coupling: not folded)
            y = y + residual
        return (y, None) if want_stats else y
