"""Node-wise ops on (rows, C) feature matrices around the sparse hot path.

``layer_norm_rows`` / ``LayerNorm`` = nn.LayerNorm over the channel dimension (norm_layer('layer', C), the default norm
of the ogbn-proteins / ogbg-ppa / RevGCN configurations), same kernels file, optional fused ReLU.

``batch_norm_rows`` = nn.BatchNorm1d forward/backward (training and eval semantics, running statistics) with an
optional fused ReLU, as HIP streaming kernels (csrc/rows_norm.hip).  ``BatchNorm1d`` is the drop-in module that
``gcn_lib.sparse.torch_nn.norm_layer('batch', C)`` returns: an ``nn.BatchNorm1d`` subclass (same parameters,
buffers, ``state_dict`` keys; ``isinstance`` still holds), reference: gcn_lib/sparse/torch_nn.py:23-34.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib


class _BatchNormRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, num_batches, use_batch_stats: bool, momentum: float,
                eps: float, relu: bool, track: bool):
        lib = _lib.load()
        dev = _lib.require_device(x)
        stream = _lib.current_stream_handle(dev)
        if x.stride(1) != 1 or x.data_ptr() % 16 != 0 or (x.size(0) > 1 and x.stride(0) % 4 != 0):
            x = x.contiguous()     # the float4 row layout needs 16-byte aligned rows (a sliced view may not be)
        rows, C = x.shape
        ld = x.stride(0) if rows > 1 else C
        y = torch.empty(rows, C, device=dev, dtype=torch.float32)
        bnbuf = torch.empty(4, C, device=dev, dtype=torch.float32)
        w = None if weight is None else weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        with _lib.device_ctx(dev):
            nparts, stats = 0, None
            if use_batch_stats:
                nparts = lib.dgcn_rows_num_partials(rows, C)
                stats = torch.empty(nparts, 2, C, device=dev, dtype=torch.float32)
                _lib.check(lib.dgcn_rows_stats_f32(x.data_ptr(), ld, rows, C, stats.data_ptr(), stream),
                           "dgcn_rows_stats_f32")
            _lib.check(lib.dgcn_bn_finalize_f32(
                _lib.ptr(stats), nparts, C, float(rows), _lib.ptr(w), _lib.ptr(b), _lib.ptr(running_mean),
                _lib.ptr(running_var), _lib.ptr(num_batches), 1 if use_batch_stats else 0, float(momentum),
                float(eps), bnbuf.data_ptr(), stream), "dgcn_bn_finalize_f32")
            _lib.check(lib.dgcn_rows_bn_apply_f32(x.data_ptr(), ld, bnbuf.data_ptr(), 1 if relu else 0,
                                                  y.data_ptr(), rows, C, stream), "dgcn_rows_bn_apply_f32")
        if track and any(ctx.needs_input_grad[:3]):
            ctx.save_for_backward(x, bnbuf, y if relu else None)
            ctx.cfg = (use_batch_stats, weight is not None, bias is not None, relu)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, bnbuf, y = ctx.saved_tensors
        use_batch_stats, has_w, has_b, relu = ctx.cfg
        dev = x.device
        stream = _lib.current_stream_handle(dev)
        rows, C = x.shape
        ld = x.stride(0) if rows > 1 else C
        g = g.float().contiguous()
        nparts = lib.dgcn_rows_num_partials(rows, C)
        partial = torch.empty(nparts, 2, C, device=dev, dtype=torch.float32)
        coef = torch.empty(4, C, device=dev, dtype=torch.float32)
        dx = torch.empty(rows, C, device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        with _lib.device_ctx(dev):
            _lib.check(lib.dgcn_rows_bn_bwd_stats_f32(g.data_ptr(), x.data_ptr(), ld, _lib.ptr(y), bnbuf.data_ptr(),
                                                      partial.data_ptr(), rows, C, stream), "dgcn_rows_bn_bwd_stats_f32")
            _lib.check(lib.dgcn_rows_bn_bwd_finalize_f32(partial.data_ptr(), nparts, C, float(rows),
                                                         1 if use_batch_stats else 0, coef.data_ptr(), stream),
                       "dgcn_rows_bn_bwd_finalize_f32")
            if dx is not None:
                _lib.check(lib.dgcn_rows_bn_bwd_apply_f32(g.data_ptr(), x.data_ptr(), ld, _lib.ptr(y), bnbuf.data_ptr(),
                                                          coef.data_ptr(), dx.data_ptr(), rows, C, stream),
                           "dgcn_rows_bn_bwd_apply_f32")
        gw = coef[0] if (has_w and ctx.needs_input_grad[1]) else None      # rows of a fresh tensor: no copy needed
        gb = coef[1] if (has_b and ctx.needs_input_grad[2]) else None
        return dx, gw, gb, None, None, None, None, None, None, None, None


def _supported(x: torch.Tensor) -> bool:
    C = x.size(-1)
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.size(0) > 0
            and ((C % 4 == 0 and C <= 1024) or C <= 256) and not torch.is_autocast_enabled())


def batch_norm_rows(x, weight, bias, running_mean, running_var, num_batches, training: bool, momentum: float,
                    eps: float, relu: bool = False) -> torch.Tensor:
    """BatchNorm1d over the rows of ``x`` (rows, C) [+ ReLU]; ``training`` selects batch statistics (and updates the
    running buffers in place when given)."""
    return _BatchNormRows.apply(x, weight, bias, running_mean, running_var, num_batches, bool(training),
                                float(momentum), float(eps), bool(relu), torch.is_grad_enabled())


class BatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d whose 2-D fp32 device inputs run on the HIP kernels; anything else (3-D inputs, other dtypes,
    autocast, CPU tensors of a model not yet moved) takes the stock implementation."""

    def _hip_args(self, x):
        if not _supported(x) or x.size(1) != self.num_features:
            return None
        if self.training and x.size(0) == 1:
            return None                              # stock path raises the "more than 1 value per channel" error
        use_batch = self.training or (self.running_mean is None and self.running_var is None)
        momentum = 0.0 if self.momentum is None else self.momentum
        nb = None
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            nb = self.num_batches_tracked
            if self.momentum is None:                # cumulative moving average (host sync, as in torch)
                momentum = 1.0 / float(int(self.num_batches_tracked) + 1)
        rm = self.running_mean if (not self.training or self.track_running_stats) else None
        rv = self.running_var if (not self.training or self.track_running_stats) else None
        return use_batch, momentum, nb, rm, rv

    def forward(self, x, fuse_relu: bool = False):
        a = self._hip_args(x)
        if a is None:
            y = super().forward(x)
            return torch.relu(y) if fuse_relu else y
        use_batch, momentum, nb, rm, rv = a
        return batch_norm_rows(x, self.weight, self.bias, rm, rv, nb, use_batch, momentum, self.eps, relu=fuse_relu)


class _LayerNormRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps: float, relu: bool, track: bool):
        lib = _lib.load()
        dev = _lib.require_device(x)
        stream = _lib.current_stream_handle(dev)
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        if x2.stride(1) != 1 or x2.stride(0) % 4 != 0 or x2.data_ptr() % 16 != 0:
            x2 = x2.contiguous()
        rows, C = x2.shape
        ld = x2.stride(0) if rows > 1 else C
        y = torch.empty(rows, C, device=dev, dtype=torch.float32)
        mean = torch.empty(rows, device=dev, dtype=torch.float32)
        rstd = torch.empty(rows, device=dev, dtype=torch.float32)
        w = None if weight is None else weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        with _lib.device_ctx(dev):
            _lib.check(lib.dgcn_rows_ln_fwd_f32(x2.data_ptr(), ld, _lib.ptr(w), _lib.ptr(b), float(eps), 1 if relu else 0,
                                                y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, C, stream),
                       "dgcn_rows_ln_fwd_f32")
        if track and any(ctx.needs_input_grad[:3]):
            ctx.save_for_backward(x2, w, mean, rstd, y if relu else None)
            ctx.cfg = (weight is not None, bias is not None, tuple(shape))
        return y.view(shape)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x2, w, mean, rstd, y = ctx.saved_tensors
        has_w, has_b, shape = ctx.cfg
        dev = x2.device
        stream = _lib.current_stream_handle(dev)
        rows, C = x2.shape
        ld = x2.stride(0) if rows > 1 else C
        g2 = g.reshape(rows, C).float().contiguous()
        need_dx = ctx.needs_input_grad[0]
        need_p = (has_w and ctx.needs_input_grad[1]) or (has_b and ctx.needs_input_grad[2])
        dx = torch.empty(rows, C, device=dev, dtype=torch.float32) if need_dx else None
        partial = psum = None
        with _lib.device_ctx(dev):
            if need_p:
                nparts = lib.dgcn_rows_ln_num_partials(rows, C)
                partial = torch.empty(nparts, 2, C, device=dev, dtype=torch.float32)
            _lib.check(lib.dgcn_rows_ln_bwd_f32(g2.data_ptr(), x2.data_ptr(), ld, _lib.ptr(y), _lib.ptr(w), mean.data_ptr(),
                                                rstd.data_ptr(), _lib.ptr(dx), _lib.ptr(partial), rows, C, stream),
                       "dgcn_rows_ln_bwd_f32")
            if need_p:
                psum = partial.sum(0)                 # (2, C): sum g' | sum g' xhat over <= 1024 workgroup partials
        gw = psum[1] if (has_w and ctx.needs_input_grad[1]) else None
        gb = psum[0] if (has_b and ctx.needs_input_grad[2]) else None
        return (dx.view(shape) if dx is not None else None), gw, gb, None, None, None


def _ln_supported(x: torch.Tensor, C: int) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 2 and x.size(-1) == C and C % 4 == 0 and C <= 1024
            and x.numel() > 0 and not torch.is_autocast_enabled())


def layer_norm_rows(x, weight, bias, eps: float = 1e-5, relu: bool = False) -> torch.Tensor:
    """LayerNorm over the last dimension of ``x`` (..., C) [+ ReLU]."""
    return _LayerNormRows.apply(x, weight, bias, float(eps), bool(relu), torch.is_grad_enabled())


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm (normalised over the last dimension only) whose fp32 device inputs run on the HIP row kernels; other
    shapes / dtypes take the stock implementation.  Same parameters and ``state_dict`` keys."""

    def forward(self, x, fuse_relu: bool = False):
        if len(self.normalized_shape) == 1 and _ln_supported(x, self.normalized_shape[0]):
            return layer_norm_rows(x, self.weight, self.bias, self.eps, relu=fuse_relu)
        y = super().forward(x)
        return torch.relu(y) if fuse_relu else y


class _MsgNormRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, msg, scale, add_x: bool, track: bool):
        lib = _lib.load()
        dev = _lib.require_device(x, msg, scale)
        stream = _lib.current_stream_handle(dev)
        if x.stride(1) != 1 or x.stride(0) % 4 != 0 or x.data_ptr() % 16 != 0:
            x = x.contiguous()
        msg = msg.float().contiguous()
        sc = scale.detach().float().contiguous()
        rows, C = x.shape
        ldx = x.stride(0) if rows > 1 else C
        y = torch.empty(rows, C, device=dev, dtype=torch.float32)
        with _lib.device_ctx(dev):
            _lib.check(lib.dgcn_rows_msgnorm_fwd_f32(x.data_ptr(), ldx, msg.data_ptr(), sc.data_ptr(), 1 if add_x else 0,
                                                     y.data_ptr(), rows, C, stream), "dgcn_rows_msgnorm_fwd_f32")
        if track and any(ctx.needs_input_grad[:3]):
            ctx.save_for_backward(x, msg, sc)
            ctx.add_x = add_x
            ctx.scale_shape = tuple(scale.shape)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, msg, sc = ctx.saved_tensors
        dev = x.device
        stream = _lib.current_stream_handle(dev)
        rows, C = x.shape
        ldx = x.stride(0) if rows > 1 else C
        g = g.float().contiguous()
        dx = torch.empty(rows, C, device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        dm = torch.empty(rows, C, device=dev, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        part = None
        if ctx.needs_input_grad[2]:
            part = torch.empty(lib.dgcn_rows_ln_num_partials(rows, C), device=dev, dtype=torch.float32)
        with _lib.device_ctx(dev):
            _lib.check(lib.dgcn_rows_msgnorm_bwd_f32(g.data_ptr(), x.data_ptr(), ldx, msg.data_ptr(), sc.data_ptr(),
                                                     1 if ctx.add_x else 0, _lib.ptr(dx), _lib.ptr(dm), _lib.ptr(part),
                                                     rows, C, stream), "dgcn_rows_msgnorm_bwd_f32")
        ds = part.sum().reshape(ctx.scale_shape) if part is not None else None
        return dx, dm, ds, None, None


def msg_norm_rows(x: torch.Tensor, msg: torch.Tensor, scale: torch.Tensor, add_x: bool = False) -> torch.Tensor:
    """``[x +] normalize(msg) * ||x||_2 * scale`` row by row (MsgNorm, torch_message.py:95-99, p = 2) in one kernel."""
    return _MsgNormRows.apply(x, msg, scale, bool(add_x), torch.is_grad_enabled())


def msg_norm_supported(x: torch.Tensor, msg: torch.Tensor) -> bool:
    C = x.size(-1)
    return (x.is_cuda and x.dim() == 2 and msg.shape == x.shape and x.dtype == torch.float32 and msg.dtype == torch.float32
            and C % 4 == 0 and C <= 1024 and x.size(0) > 0 and not torch.is_autocast_enabled())
