"""Node-wise helpers around the hot path (plain library GEMMs, no custom kernels).

``TallLinear`` is an ``nn.Linear`` (same parameters, same ``state_dict`` keys, ``isinstance`` still holds)
whose weight gradient ``g^T x`` -- a GEMM with a 10^5..10^6-long reduction and a 128x128 result when the rows
are graph nodes or edges -- is computed as a batched split-K product plus a fixed-order sum.  hipBLASLt runs
the un-split shape on a handful of CUs (measured 412 us for 169,343 x 128 x 128, ~13 TFLOP/s); the split
form uses the whole chip.  Forward and input gradient are the stock GEMMs.
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
            gw = splitk_xt_g(g2.contiguous(), x.reshape(-1, x.size(-1)).contiguous())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum(0)
        return gx, gw, gb


class TallLinear(nn.Linear):
    def forward(self, x):
        if x.is_cuda and x.dim() == 2 and x.size(0) >= _MIN_ROWS and torch.is_grad_enabled() \
                and x.dtype == self.weight.dtype and not torch.is_autocast_enabled():
            return _TallLinearFn.apply(x, self.weight, self.bias)
        return super().forward(x)
