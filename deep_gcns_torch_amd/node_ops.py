"""Node-wise ops on (rows, C) feature matrices around the sparse hot path.

``layer_norm_rows`` / ``LayerNorm`` = nn.LayerNorm over the channel dimension (norm_layer('layer', C), the default norm
of the ogbn-proteins / ogbg-ppa / RevGCN configurations), same kernels file, optional fused ReLU.

``batch_norm_rows`` = nn.BatchNorm1d forward/backward (training and eval semantics, running statistics) with an
optional fused ReLU, as HIP streaming kernels (csrc/rows_norm.hip).  ``BatchNorm1d`` is the drop-in module that
``gcn_lib.sparse.torch_nn.norm_layer('batch', C)`` returns: an ``nn.BatchNorm1d`` subclass (same parameters,
buffers, ``state_dict`` keys; ``isinstance`` still holds), reference: gcn_lib/sparse/torch_nn.py:23-34.

``pre_activation`` = the ``norm -> ReLU -> dropout`` run in front of every convolution of the 'res+' models
(examples/ogb/ogbn_arxiv/model.py:90-106) and of the reversible BasicBlock (eff_gcn_modules/rev/rev_layer.py:35-51) as
ONE apply pass: the dropout mask is a counter hash of (seed, element index) regenerated in the backward (nothing stored)
or the shared mask tensor of the reversible model; the backward recomputes the ReLU mask from the input.

``rows_linear`` / ``RowsLinear`` = ``nn.Linear`` over node rows on the bf16 matrix pipe (fp32-faithful six-product
split, csrc/rows_linear.hip) with bias, the 'res+' residual and the NEXT BatchNorm's statistics folded into the same
sweep (gcn_lib/sparse/torch_nn.py:50-71, gcn_lib/sparse/torch_vertex.py:70-76).
"""
from __future__ import annotations

import threading

import numpy as np
import torch
from torch import nn

from . import _lib

# ---- dropout inside the row kernels (csrc/rows_norm.hip: drop_rand4) ------------------------------------------------
DROP_NONE, DROP_HASH, DROP_MASK = 0, 1, 2


class DropSpec:
    """How the fused pre-activation drops: ``mode`` (DROP_*), the shared ``mask`` tensor or the hash ``seed`` words and
    16-bit threshold (drop probability thr / 65536; kept values are scaled by 65536 / (65536 - thr))."""
    __slots__ = ("mode", "mask", "s0", "s1", "thr")

    def __init__(self, mode=DROP_NONE, mask=None, s0=0, s1=0, thr=0):
        self.mode, self.mask, self.s0, self.s1, self.thr = mode, mask, s0, s1, thr

    @staticmethod
    def none():
        return DropSpec()

    @staticmethod
    def hashed(p: float, seed=None):
        """Bernoulli(1 - p) keep mask from a fresh seed (drawn from torch's CPU generator: reproducible under
        torch.manual_seed; a step replayed as a HIP graph replays the same mask)."""
        thr = int(round(float(p) * 65536.0))
        if thr <= 0:
            return DropSpec()
        thr = min(thr, 65535)
        if seed is None:
            w = torch.randint(0, 2 ** 31 - 1, (2,), dtype=torch.int64)
            seed = (int(w[0]), int(w[1]))
        return DropSpec(DROP_HASH, None, int(seed[0]) & 0xFFFFFFFF, int(seed[1]) & 0xFFFFFFFF, thr)

    @staticmethod
    def shared(mask: torch.Tensor):
        return DropSpec(DROP_MASK, mask)

    def args(self):
        return (self.mode, _lib.ptr(self.mask), self.mask.stride(0) if self.mask is not None else 0, self.s0, self.s1,
                self.thr)


def hash_keep_factors(rows: int, C: int, s0: int, s1: int, thr: int) -> torch.Tensor:
    """Host replica of the kernels' hash dropout: the (rows, C) factor array (0 or 65536 / (65536 - thr)).  For tests and
    for callers that need the mask itself; the kernels never materialise it."""
    M = np.uint64(0xFFFFFFFF)

    def mix(h):
        h = h ^ (h >> np.uint64(16)); h = (h * np.uint64(0x85ebca6b)) & M
        h = h ^ (h >> np.uint64(13)); h = (h * np.uint64(0xc2b2ae35)) & M
        return h ^ (h >> np.uint64(16))

    n = rows * C
    q = np.arange((n + 3) // 4, dtype=np.uint64)
    lo, hi = q & M, q >> np.uint64(32)
    h0 = mix(lo ^ np.uint64(s0))
    h0 = mix((h0 + ((hi * np.uint64(0x9E3779B1)) & M) + np.uint64(s1)) & M)
    h1 = mix(h0 ^ np.uint64(0x68E31DA4))
    r = np.stack([h0 & np.uint64(0xffff), h0 >> np.uint64(16), h1 & np.uint64(0xffff), h1 >> np.uint64(16)], 1).reshape(-1)[:n]
    keep = r >= np.uint64(thr)
    return torch.from_numpy(np.where(keep, np.float32(65536.0 / (65536 - thr)), np.float32(0.0)).astype(np.float32)).view(rows, C)


def _mask_rows(drop: DropSpec, rows: int, C: int, dev):
    """The shared mask as fp32 (rows, C) rows with unit column stride on ``dev`` (a torch.chunk view of the model-level
    (N, hidden) mask is used in place: its row stride goes to the kernel)."""
    if drop.mode != DROP_MASK:
        return drop
    m = drop.mask
    if m.shape != (rows, C) or m.device != dev:
        raise ValueError(f"dropout mask must be a ({rows}, {C}) tensor on {dev}")
    if m.dtype != torch.float32 or m.stride(1) != 1 or (rows > 1 and m.stride(0) < C):
        m = m.float().contiguous()
    return DropSpec(DROP_MASK, m)


class _BatchNormRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, num_batches, use_batch_stats: bool, momentum: float,
                eps: float, relu: bool, track: bool, drop: DropSpec = None, stats=None, sync=None, skip: bool = False):
        """``sync`` (dist.BatchSync or None): all-reduce the statistics partials over the ranks of a node-partitioned
        graph, forward and backward: the batch is the whole graph's rows.
        ``skip``: also return the input itself (an alias) for the caller's skip connection; its gradient is added to
        ``dx`` inside the backward's apply kernel instead of by an autograd accumulation pass."""
        lib = _lib.load()
        dev = _lib.require_device(x)
        stream = _lib.current_stream_handle(dev)
        if x.stride(1) != 1 or x.data_ptr() % 16 != 0 or (x.size(0) > 1 and x.stride(0) % 4 != 0):
            x = x.contiguous()     # the float4 row layout needs 16-byte aligned rows (a sliced view may not be)
        rows, C = x.shape
        ld = x.stride(0) if rows > 1 else C
        drop = _mask_rows(drop or DropSpec.none(), rows, C, dev)
        y = torch.empty(rows, C, device=dev, dtype=torch.float32)
        bnbuf = torch.empty(4, C, device=dev, dtype=torch.float32)
        w = None if weight is None else weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        with _lib.device_ctx(dev):
            nparts = 0
            if use_batch_stats:
                if stats is not None:
                    # per-workgroup sum x | sum x^2 partials that the producing kernel (rows_linear) already wrote
                    if stats.dim() != 3 or stats.shape[1:] != (2, C) or stats.dtype != torch.float32 or not stats.is_contiguous():
                        raise ValueError("stats must be contiguous fp32 (parts, 2, C) partial sums")
                    nparts = stats.size(0)
                else:
                    nparts = lib.dgcn_rows_num_partials(rows, C)
                    stats = torch.empty(nparts, 2, C, device=dev, dtype=torch.float32)
                    _lib.check(lib.dgcn_rows_stats_f32(x.data_ptr(), ld, rows, C, stats.data_ptr(), stream),
                               "dgcn_rows_stats_f32")
            else:
                stats = None
            count = float(rows)
            if sync is not None and use_batch_stats:
                stats = sync.reduce(stats)
                nparts, count = stats.size(0), float(sync.total_rows)
            _lib.check(lib.dgcn_bn_finalize_f32(
                _lib.ptr(stats), nparts, C, count, _lib.ptr(w), _lib.ptr(b), _lib.ptr(running_mean),
                _lib.ptr(running_var), _lib.ptr(num_batches), 1 if use_batch_stats else 0, float(momentum),
                float(eps), bnbuf.data_ptr(), stream), "dgcn_bn_finalize_f32")
            _lib.check(lib.dgcn_rows_bn_act_apply_f32(x.data_ptr(), ld, bnbuf.data_ptr(), 1 if relu else 0, *drop.args(),
                                                      y.data_ptr(), rows, C, stream), "dgcn_rows_bn_act_apply_f32")
        if track and any(ctx.needs_input_grad[:3]):
            ctx.save_for_backward(x, bnbuf, drop.mask)
            ctx.cfg = (use_batch_stats, weight is not None, bias is not None, relu, drop, sync)
        if skip:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, g, g_skip=None):
        lib = _lib.load()
        x, bnbuf, mask = ctx.saved_tensors
        use_batch_stats, has_w, has_b, relu, drop, sync = ctx.cfg
        dev = x.device
        stream = _lib.current_stream_handle(dev)
        rows, C = x.shape
        ld = x.stride(0) if rows > 1 else C
        if g is None:                                   # only the skip output was used
            g = torch.zeros(rows, C, device=dev, dtype=torch.float32)
        g = g.float().contiguous()
        if g_skip is not None:
            g_skip = g_skip.float().contiguous()
        nparts = lib.dgcn_rows_num_partials(rows, C)
        partial = torch.empty(nparts, 2, C, device=dev, dtype=torch.float32)
        coef = torch.empty(4, C, device=dev, dtype=torch.float32)
        dx = torch.empty(rows, C, device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        with _lib.device_ctx(dev):
            # the ReLU mask is recomputed from x and the saved coefficients, the hash mask from its seed
            _lib.check(lib.dgcn_rows_bn_act_bwd_stats_f32(g.data_ptr(), x.data_ptr(), ld, bnbuf.data_ptr(),
                                                          1 if relu else 0, *drop.args(), partial.data_ptr(), rows, C,
                                                          stream), "dgcn_rows_bn_act_bwd_stats_f32")
            count = float(rows)
            local = None
            if sync is not None and use_batch_stats:
                # c1, c2 need the sums over ALL ranks' rows; dgamma / dbeta stay this rank's share (the caller sums the
                # replicated parameters' gradients over the ranks)
                local = partial.double().sum(0).float()
                partial = sync.reduce(partial)
                nparts, count = partial.size(0), float(sync.total_rows)
            _lib.check(lib.dgcn_rows_bn_bwd_finalize_f32(partial.data_ptr(), nparts, C, count,
                                                         1 if use_batch_stats else 0, coef.data_ptr(), stream),
                       "dgcn_rows_bn_bwd_finalize_f32")
            if dx is not None:
                _lib.check(lib.dgcn_rows_bn_act_bwd_apply_f32(g.data_ptr(), x.data_ptr(), ld, bnbuf.data_ptr(),
                                                              coef.data_ptr(), 1 if relu else 0, *drop.args(),
                                                              _lib.ptr(g_skip), dx.data_ptr(), rows, C, stream),
                           "dgcn_rows_bn_act_bwd_apply_f32")
        if local is not None:
            gw = local[1] if (has_w and ctx.needs_input_grad[1]) else None
            gb = local[0] if (has_b and ctx.needs_input_grad[2]) else None
        else:
            gw = coef[0] if (has_w and ctx.needs_input_grad[1]) else None  # rows of a fresh tensor: no copy needed
            gb = coef[1] if (has_b and ctx.needs_input_grad[2]) else None
        return (dx, gw, gb) + (None,) * 12


def _supported(x: torch.Tensor) -> bool:
    C = x.size(-1)
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.size(0) > 0
            and ((C % 4 == 0 and C <= 1024) or C <= 256) and not torch.is_autocast_enabled())


def batch_norm_rows(x, weight, bias, running_mean, running_var, num_batches, training: bool, momentum: float,
                    eps: float, relu: bool = False, drop: DropSpec = None, stats=None, sync=None, skip: bool = False):
    """BatchNorm1d over the rows of ``x`` (rows, C) [+ ReLU] [+ dropout]; ``training`` selects batch statistics (and
    updates the running buffers in place when given).  ``stats``: (parts, 2, C) partial sums of x and x^2 that the
    producer of ``x`` already computed (``rows_linear(..., want_stats=True)``): the statistics pass is skipped."""
    return _BatchNormRows.apply(x, weight, bias, running_mean, running_var, num_batches, bool(training),
                                float(momentum), float(eps), bool(relu), torch.is_grad_enabled(), drop,
                                stats if training else None, sync if training else None, bool(skip))


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

    def forward(self, x, fuse_relu: bool = False, drop: DropSpec = None, stats=None, skip: bool = False):
        """skip: return ``(y, x)`` -- the second output carries the caller's skip connection (see _BatchNormRows)."""
        sync = _active_sync() if self.training else None      # node-partitioned graph: statistics over all ranks' rows
        a = self._hip_args(x)
        if a is None:
            if sync is not None and x.dim() == 2:
                y = _sync_bn_torch(self, x, fuse_relu, drop, sync)
            else:
                y = _stock_tail(super().forward(x), fuse_relu, drop)
            return (y, x) if skip else y
        use_batch, momentum, nb, rm, rv = a
        return batch_norm_rows(x, self.weight, self.bias, rm, rv, nb, use_batch, momentum, self.eps, relu=fuse_relu,
                               drop=drop, stats=stats if use_batch else None, sync=sync if use_batch else None,
                               skip=skip)


def _active_sync():
    import sys
    d = sys.modules.get(__package__ + ".dist")         # only a process that imported dist can have an active partition
    ctx = d.active_partition() if d is not None else None
    return None if ctx is None else ctx.sync


def _sync_bn_torch(bn, x, relu, drop, sync):
    """Training-mode BatchNorm1d over all ranks' rows for inputs the row kernels do not take (CPU tensors: the gloo
    tests): dist._SyncBatchNormTorch."""
    from . import dist as _d
    factors = None
    if drop is not None and drop.mode == DROP_MASK:
        factors = drop.mask
    elif drop is not None and drop.mode == DROP_HASH:
        factors = hash_keep_factors(x.size(0), x.size(1), drop.s0, drop.s1, drop.thr).to(x.device)
    track = bn.track_running_stats
    if bn.momentum is not None:
        momentum = bn.momentum
    elif track and bn.num_batches_tracked is not None:     # nn.BatchNorm1d: cumulative moving average
        momentum = 1.0 / float(int(bn.num_batches_tracked) + 1)
    else:
        momentum = 0.0
    return _d._SyncBatchNormTorch.apply(x, bn.weight, bn.bias, bn.running_mean if track else None,
                                        bn.running_var if track else None, bn.num_batches_tracked if track else None,
                                        momentum, bn.eps, relu, factors, sync)


def _stock_tail(y, relu: bool, drop: DropSpec):
    """ReLU / dropout after a stock norm (inputs the row kernels do not take: CPU tensors, other dtypes, 3-D)."""
    if relu:
        y = torch.relu(y)
    if drop is not None and drop.mode == DROP_MASK:
        y = y * drop.mask
    elif drop is not None and drop.mode == DROP_HASH:
        f = hash_keep_factors(y.numel() // y.size(-1), y.size(-1), drop.s0, drop.s1, drop.thr).to(y.device)
        y = y * f.view(y.shape).to(y.dtype)
    return y


class _LayerNormRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps: float, relu: bool, track: bool, drop: DropSpec = None, skip: bool = False):
        lib = _lib.load()
        dev = _lib.require_device(x)
        stream = _lib.current_stream_handle(dev)
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        if x2.stride(1) != 1 or x2.stride(0) % 4 != 0 or x2.data_ptr() % 16 != 0:
            x2 = x2.contiguous()
        rows, C = x2.shape
        ld = x2.stride(0) if rows > 1 else C
        drop = drop or DropSpec.none()
        if drop.mode == DROP_MASK and drop.mask.dim() != 2:
            drop = DropSpec.shared(drop.mask.reshape(rows, C))
        drop = _mask_rows(drop, rows, C, dev)
        y = torch.empty(rows, C, device=dev, dtype=torch.float32)
        mean = torch.empty(rows, device=dev, dtype=torch.float32)
        rstd = torch.empty(rows, device=dev, dtype=torch.float32)
        w = None if weight is None else weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        with _lib.device_ctx(dev):
            _lib.check(lib.dgcn_rows_ln_act_fwd_f32(x2.data_ptr(), ld, _lib.ptr(w), _lib.ptr(b), float(eps),
                                                    1 if relu else 0, *drop.args(), y.data_ptr(), mean.data_ptr(),
                                                    rstd.data_ptr(), rows, C, stream), "dgcn_rows_ln_act_fwd_f32")
        if track and any(ctx.needs_input_grad[:3]):
            ctx.save_for_backward(x2, w, b, mean, rstd, drop.mask)
            ctx.cfg = (weight is not None, bias is not None, tuple(shape), relu, drop)
        if skip:
            return y.view(shape), x.view_as(x)
        return y.view(shape)

    @staticmethod
    def backward(ctx, g, g_skip=None):
        lib = _lib.load()
        x2, w, b, mean, rstd, mask = ctx.saved_tensors
        has_w, has_b, shape, relu, drop = ctx.cfg
        dev = x2.device
        stream = _lib.current_stream_handle(dev)
        rows, C = x2.shape
        ld = x2.stride(0) if rows > 1 else C
        if g is None:
            g = torch.zeros(rows, C, device=dev, dtype=torch.float32)
        g2 = g.reshape(rows, C).float().contiguous()
        gs2 = None if g_skip is None else g_skip.reshape(rows, C).float().contiguous()
        need_dx = ctx.needs_input_grad[0]
        need_p = (has_w and ctx.needs_input_grad[1]) or (has_b and ctx.needs_input_grad[2])
        dx = torch.empty(rows, C, device=dev, dtype=torch.float32) if need_dx else None
        partial = psum = None
        with _lib.device_ctx(dev):
            if need_p:
                nparts = lib.dgcn_rows_ln_num_partials(rows, C)
                partial = torch.empty(nparts, 2, C, device=dev, dtype=torch.float32)
            _lib.check(lib.dgcn_rows_ln_act_bwd_f32(g2.data_ptr(), x2.data_ptr(), ld, _lib.ptr(w), _lib.ptr(b),
                                                    mean.data_ptr(), rstd.data_ptr(), 1 if relu else 0, *drop.args(),
                                                    _lib.ptr(gs2) if need_dx else None, _lib.ptr(dx), _lib.ptr(partial),
                                                    rows, C, stream),
                       "dgcn_rows_ln_act_bwd_f32")
            if need_p:
                psum = _lib.sum_partials(partial)     # (2, C): sum g' | sum g' xhat over <= 1024 workgroup partials, one launch
        gw = psum[1] if (has_w and ctx.needs_input_grad[1]) else None
        gb = psum[0] if (has_b and ctx.needs_input_grad[2]) else None
        return (dx.view(shape) if dx is not None else None), gw, gb, None, None, None, None, None


def _ln_supported(x: torch.Tensor, C: int) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 2 and x.size(-1) == C and C % 4 == 0 and C <= 1024
            and x.numel() > 0 and not torch.is_autocast_enabled())


def layer_norm_rows(x, weight, bias, eps: float = 1e-5, relu: bool = False, drop: DropSpec = None, skip: bool = False):
    """LayerNorm over the last dimension of ``x`` (..., C) [+ ReLU] [+ dropout]."""
    return _LayerNormRows.apply(x, weight, bias, float(eps), bool(relu), torch.is_grad_enabled(), drop, bool(skip))


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm (normalised over the last dimension only) whose fp32 device inputs run on the HIP row kernels; other
    shapes / dtypes take the stock implementation.  Same parameters and ``state_dict`` keys."""

    def forward(self, x, fuse_relu: bool = False, drop: DropSpec = None, stats=None, skip: bool = False):
        if len(self.normalized_shape) == 1 and _ln_supported(x, self.normalized_shape[0]):
            return layer_norm_rows(x, self.weight, self.bias, self.eps, relu=fuse_relu, drop=drop, skip=skip)
        y = _stock_tail(super().forward(x), fuse_relu, drop)
        return (y, x) if skip else y


def pre_activation(norm: nn.Module, x: torch.Tensor, p: float = 0.0, training: bool = True, mask: torch.Tensor = None,
                   stats=None, skip: bool = False):
    """``dropout(relu(norm(x)))`` -- the run in front of every convolution of the 'res+' models
    (examples/ogb/ogbn_arxiv/model.py:96-99: ``norm -> F.relu -> F.dropout(p, training)``) and of the reversible
    BasicBlock (eff_gcn_modules/rev/rev_layer.py:38-46: ``norm -> relu -> x * shared mask``) in ONE row kernel when
    ``norm`` is this package's BatchNorm1d / LayerNorm; any other module runs the three steps.
    ``mask``: the shared dropout mask (already scaled) -- used as is, ``p`` is ignored.  ``stats``: see batch_norm_rows.
    ``skip``: return ``(y, x_skip)`` with ``x_skip`` = ``x`` for the skip connection AROUND the block that follows
    (``conv(y) + x_skip``): the gradient of that connection is added inside this op's backward kernel instead of by a
    separate accumulation pass over the (N, C) gradient."""
    if mask is not None:
        drop = DropSpec.shared(mask) if training else None
    else:
        drop = DropSpec.hashed(p) if (training and p > 0.0) else None
    if isinstance(norm, (BatchNorm1d, LayerNorm)):
        return norm(x, fuse_relu=True, drop=drop, stats=stats, skip=skip)
    y = torch.relu(norm(x))
    if mask is not None:
        y = y * mask if training else y
    else:
        y = torch.nn.functional.dropout(y, p=p, training=training)
    return (y, x) if skip else y


# ---- Linear over node rows ------------------------------------------------------------------------------------------
ROWS_LINEAR_MIN_ROWS = 2048     # below this the library GEMM is as good (launch-bound either way)

_WARNED = set()


def warn_library_gemm(kind: str, rows: int, a: int, b: int, why: str) -> None:
    """Say ONCE per shape that a node-row product left the row kernels for the library GEMM (same results, slower):
    a user must not lose the kernel without notice.  Small inputs (< ROWS_LINEAR_MIN_ROWS rows) are silent:
    there the library call is the intended path."""
    key = (kind, a, b)
    if key in _WARNED or rows < ROWS_LINEAR_MIN_ROWS:
        return
    _WARNED.add(key)
    import warnings
    warnings.warn(f"deep_gcns_torch_amd: {kind} of {rows} rows x ({a}, {b}) runs on the library GEMM, not on the row "
                  f"kernel ({why}); results are the same", RuntimeWarning, stacklevel=3)


def rows_linear_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32
            and x.size(0) >= ROWS_LINEAR_MIN_ROWS and not torch.is_autocast_enabled()):
        return False
    C, K = weight.shape
    return x.size(1) == K and bool(_lib.load().dgcn_rows_linear_supported(K, C))


def _rl_launch(x, w, w_trans, bias, res, relu, want_stats, want_xsum, out=None, negate=False, late=False):
    """y = x @ (w^T | w) + bias + res on the matrix pipe; returns (y, stats or None, xsum partials or None).
    ``out``: write the result there (fp32 (rows, C), unit column stride); ``out is res`` accumulates in place -- every
    element is read (one 16-row batch ahead) and written by the same wave, rows of different waves are disjoint.
    ``late``: the residual joins behind the product chain (rounded once at its magnitude, as a separate add would);
    ``negate``: y = res - (x @ w^T + bias), likewise."""
    lib = _lib.load()
    dev = x.device
    if x.stride(1) != 1 or x.stride(0) % 4 != 0 or x.data_ptr() % 16 != 0:
        x = x.contiguous()
    rows, K = x.shape
    C = w.size(1) if w_trans else w.size(0)
    if w.stride(1) != 1:
        w = w.contiguous()
    if res is not None and (res.stride(1) != 1 or res.dtype != torch.float32):
        res = res.float().contiguous()
    if out is None:
        y = torch.empty(rows, C, device=dev, dtype=torch.float32)
    else:
        if out.shape != (rows, C) or out.dtype != torch.float32 or out.stride(1) != 1:
            raise ValueError("rows_linear: out must be fp32 (rows, C) with unit column stride")
        y = out
    nparts = lib.dgcn_rows_linear_num_partials(rows, K, C)
    stats = torch.empty(nparts, 2, C, device=dev, dtype=torch.float32) if want_stats else None
    xsum = torch.empty(nparts, K, device=dev, dtype=torch.float32) if want_xsum else None
    with _lib.device_ctx(dev):
        _lib.check(lib.dgcn_rows_linear_f32(x.data_ptr(), x.stride(0), rows, w.data_ptr(), w.stride(0), 1 if w_trans else 0,
                                            _lib.ptr(bias), _lib.ptr(res), res.stride(0) if res is not None else 0,
                                            y.data_ptr(), y.stride(0), K, C, (1 if relu else 0) | (2 if negate else 0) | (4 if late else 0),
                                            _lib.ptr(stats), _lib.ptr(xsum),
                                            _lib.current_stream_handle(dev)), "dgcn_rows_linear_f32")
    return y, stats, xsum


class _ResidualGradView(threading.local):
    on = False


_RES_GRAD_VIEW = _ResidualGradView()


class residual_gradient_is_last_use:
    """``with residual_gradient_is_last_use():`` -- the caller promises that the ``residual`` tensors of the
    ``rows_linear`` calls inside have NO other consumer in the graph being recorded (``blocks.res_plus_layer`` inside
    its checkpoint: ``h`` enters the recomputed function for the skip connection only).  Their gradient -- the upstream
    gradient itself -- is then handed on as a fresh view object, which a leaf can take over instead of cloning it: the
    reentrant ``torch.utils.checkpoint`` copied the (N, C) skip gradient once per layer (26 x 87 MB per DeeperGCN-28
    step).  Without the promise the gradient is returned as it came: a view stolen by a leaf that a second consumer
    then accumulates into in place would overwrite the upstream gradient."""

    def __enter__(self):
        self.prev, _RES_GRAD_VIEW.on = _RES_GRAD_VIEW.on, True
        return self

    def __exit__(self, *exc):
        _RES_GRAD_VIEW.on = self.prev
        return False


class CouplingResidual:
    """What the additive coupling of the reversible layers (eff_gcn_modules/rev/memgcn.py) hands down as ``residual`` to
    the block it wraps: ``out = res + F(.)`` (forward, ``y_i = x_i + F_i``) or ``out = res - F(.)`` (``negate``: the
    inverse, ``x_i = y_i - F_i``) is to be written straight into ``out`` -- a column block of the tensor the coupling
    assembles -- by the epilogue of F's LAST Linear instead of by an elementwise pass behind it.  Whoever folds it sets
    ``used`` and returns a tensor that aliases ``out``; a level that cannot (library GEMM, other shapes) ignores it and
    the coupling does the add itself.  ``res`` takes no gradient here.  Under autograd the returned tensor carries the
    VALUES ``res -/+ F`` and the graph of ``F``: the coupling differentiates F through it with the gradient that
    reaches F's output (memgcn.fused_backward)."""
    __slots__ = ("res", "out", "negate", "used")

    def __init__(self, res, out, negate=False):
        self.res, self.out, self.negate, self.used = res, out, bool(negate), False

    def fits(self, rows, C, dev) -> bool:
        r, o = self.res, self.out
        return (not self.used and r.shape == (rows, C) and o.shape == (rows, C) and r.dtype == torch.float32
                and o.dtype == torch.float32 and r.device == dev and o.device == dev and r.stride(1) == 1
                and o.stride(1) == 1 and not r.requires_grad)


def _alias_with_own_version(t: torch.Tensor) -> torch.Tensor:
    """A tensor over the same memory whose version counter is its own: writes to OTHER column blocks of the buffer ``t`` is
    a view of do not invalidate what autograd saved of this one (the blocks are disjoint)."""
    return torch.empty(0, device=t.device, dtype=t.dtype).set_(t.untyped_storage(), t.storage_offset(), t.size(), t.stride())


class _RowsLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, want_stats: bool, coupling=None):
        ctx.res_view = _RES_GRAD_VIEW.on
        ctx.set_materialize_grads(False)          # no zero-filled gradient for the (non-differentiable) statistics output
        w = weight.detach()
        b = None if bias is None else bias.detach().float().contiguous()
        if coupling is not None:
            _rl_launch(x, w, False, b, coupling.res, False, False, False, out=coupling.out, negate=coupling.negate,
                       late=True)
            coupling.used = True
            ctx.save_for_backward(x, weight)
            ctx.has_bias = bias is not None
            return _alias_with_own_version(coupling.out), None
        y, stats, _ = _rl_launch(x, w, False, b, residual, False, want_stats, False)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y, None

    @staticmethod
    def backward(ctx, g, _gstats):
        from .nn_util import splitk_xt_g
        if g is None:
            return None, None, None, None, None, None
        x, weight = ctx.saved_tensors
        g = g.float()
        if g.stride(1) != 1 or g.stride(0) % 4 != 0 or g.data_ptr() % 16 != 0:
            g = g.contiguous()
        C, K = weight.shape
        gx = gw = gb = None
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        # the bias gradient g.sum(0) leaves the weight gradient's pass over g (rows_tn's loading waves)
        tn_b = need_b and ctx.needs_input_grad[1] and g.dim() == 2
        if ctx.needs_input_grad[0]:
            # dX = G W on the same kernel (reduction over C); without a weight gradient its sweep over G yields the bias's
            fuse_b = need_b and not tn_b and C <= 128 and bool(_lib.load().dgcn_rows_linear_supported(C, K))
            if _lib.load().dgcn_rows_linear_supported(C, K):
                gx, _, xsum = _rl_launch(g, weight.detach(), True, None, None, False, False, fuse_b)
                if fuse_b:
                    gb = _lib.sum_partials(xsum) if xsum.is_contiguous() else xsum.sum(0)
            else:
                gx = g @ weight
        if ctx.needs_input_grad[1]:
            if tn_b:
                gw, gb = rows_tn(g, x, with_colsum=True)
            else:
                gw = rows_tn(g, x)
        if need_b and gb is None:
            gb = g.sum(0)
        gres = None
        if ctx.needs_input_grad[3]:
            gres = g.view_as(g) if ctx.res_view else g      # (see residual_gradient_is_last_use)
        return gx, gw, gb, gres, None, None


def rows_matmul_accumulate_(acc: torch.Tensor, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """``acc += x @ w`` in place for (rows, K) x (K, C) with rows >> K, C (no autograd): the running sum of the shared
    edge-embedding gradient in the reversible backward (``dz @ W`` of every layer, ops._GenAggregate.backward)."""
    if (x.is_cuda and x.dtype == torch.float32 and acc.dtype == torch.float32 and w.dtype == torch.float32
            and x.size(0) >= ROWS_LINEAR_MIN_ROWS and _lib.load().dgcn_rows_linear_supported(x.size(1), w.size(1))):
        _rl_launch(x, w, True, None, acc, False, False, False, out=acc)
        return acc
    return torch.addmm(acc, x, w, out=acc)


def rows_matmul(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """``x @ w`` for (rows, K) x (K, C) with rows >> K, C (no autograd)."""
    if (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.size(0) >= ROWS_LINEAR_MIN_ROWS
            and _lib.load().dgcn_rows_linear_supported(x.size(1), w.size(1))):
        return _rl_launch(x, w, True, None, None, False, False, False)[0]
    return x @ w


ROWS_TN_KERNEL = True      # False: library split-K GEMM for the weight gradients (A/B measurements)


def _rows_tn_kernel_ok(g: torch.Tensor, x: torch.Tensor) -> bool:
    return bool(ROWS_TN_KERNEL and g.is_cuda and g.dtype == torch.float32 and x.dtype == torch.float32 and g.dim() == 2
                and x.dim() == 2 and g.size(0) == x.size(0) and g.size(0) >= ROWS_LINEAR_MIN_ROWS and g.stride(1) == 1
                and x.stride(1) == 1 and max(g.stride(0), x.stride(0)) < (1 << 22)
                and g.stride(0) % 4 == 0 and x.stride(0) % 4 == 0 and g.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
                and _lib.load().dgcn_rows_tn_supported(g.size(1), x.size(1)))


def rows_tn(g: torch.Tensor, x: torch.Tensor, with_colsum: bool = False):
    """``g.T @ x`` for (rows, C), (rows, K) with rows >> C, K: the weight gradient of a row-wise Linear (no autograd).
    csrc/rows_tn.hip on device rows (min(C, K) <= 128, max(C, K) <= 256, both multiples of 4, unit column strides,
    16-byte aligned rows); anything else: the split-K library GEMM.
    ``with_colsum``: returns ``(g.T @ x, g.sum(0))`` -- the Linear's bias gradient leaves the same pass over the rows (the
    kernel's loading waves hold every element of ``g`` once), no launch of its own."""
    from .nn_util import splitk_xt_g
    C, K = g.size(1), x.size(1)
    # the kernel's first operand is the narrow one: a wide g (> 128 columns) goes second and the result is written
    # transposed by the partial-sum launch, (x^T g)^T
    swap = C > 128 and K <= 128 and g.dim() == 2 and x.dim() == 2
    a, b = (x, g) if swap else (g, x)
    if not _rows_tn_kernel_ok(a, b):
        if ROWS_TN_KERNEL and g.is_cuda and g.dim() == 2 and g.dtype == torch.float32:
            warn_library_gemm("the weight gradient g^T x", g.size(0), C, K,
                              "it takes min(C, K) <= 128, max(C, K) <= 256, both multiples of 4, 16-byte aligned fp32 rows")
        out = splitk_xt_g(g.contiguous(), x.contiguous())
        return (out, g.sum(0)) if with_colsum else out
    lib = _lib.load()
    dev = g.device
    rows = g.size(0)
    Ca, Kb = a.size(1), b.size(1)
    nparts = lib.dgcn_rows_tn_num_partials(rows, Ca, Kb)
    parts = torch.empty(nparts, Ca * Kb + (C if with_colsum else 0), device=dev, dtype=torch.float32)
    out = torch.empty(C, K, device=dev, dtype=torch.float32)
    colsum = torch.empty(C, device=dev, dtype=torch.float32) if with_colsum else None
    with _lib.device_ctx(dev):
        stream = _lib.current_stream_handle(dev)
        _lib.check(lib.dgcn_rows_tn_colsum_f32(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), rows, Ca, Kb,
                                               parts.data_ptr(), out.data_ptr(), K, 1 if swap else 0,
                                               (2 if swap else 1) if with_colsum else 0, _lib.ptr(colsum), stream),
                   "dgcn_rows_tn_colsum_f32")
    return (out, colsum) if with_colsum else out


def rows_linear(x, weight, bias=None, residual=None, want_stats: bool = False):
    """``x @ weight.T + bias [+ residual]`` for (rows, K) node features; with ``want_stats`` also the per-workgroup
    partial sums (parts, 2, C) of the result and its square (hand them to the following BatchNorm1d as ``stats``).
    Returns ``y`` or ``(y, stats)``."""
    if isinstance(residual, CouplingResidual):
        cr = residual if (not want_stats and x.dim() == 2 and residual.fits(x.size(0), weight.size(0), x.device)) else None
        y, stats = _RowsLinear.apply(x, weight, bias, None, bool(want_stats), cr)
        return (y, stats) if want_stats else y
    y, stats = _RowsLinear.apply(x, weight, bias, residual, bool(want_stats))
    return (y, stats) if want_stats else y


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
