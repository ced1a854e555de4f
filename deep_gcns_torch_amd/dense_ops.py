"""Autograd-aware wrappers of the dense (B,C,N,1) kernels of libdgcn:

  knn_edge_index / knn_indices  fused distance + top-K + dilation (no (B,N,N) matrix)
  vertex_gemm                   per-vertex EdgeConv GEMM on fp32 MFMA  -> point-major (B,N,M)
  edge_reduce                   gather + act + neighbourhood max/min + BatchNorm sums

Replaces gcn_lib/dense/torch_edge.py:32-58, gcn_lib/dense/torch_nn.py:75-96 and the hot part
of gcn_lib/dense/torch_vertex.py:16-20,31-35 of the reference.
"""
from __future__ import annotations

import weakref

import torch
import torch.nn.functional as F

from . import _lib

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
USE_INVERSE_LISTS = True      # dense edge backward: dQ through inverse neighbour lists (False: LDS-privatised atomics)
EDGECONV_BWD_KERNELS = True   # EdgeConv2d's dx and [dW | db] from dP | dQ in csrc/edgeconv_bwd.hip (False: rounds 1 - 4's
                              # library calls -- sub, cat, baddbmm, permute-copy, split-K bmm, three sums: A/B measurements)
USE_KNN_FILTER = True   # candidate-filter kNN fast path for N >= 1024 (exact fallback inside the library)
KNN_BF16_PIPE = True    # distance tiles of the filter pass on the bf16 matrix pipe (C in {32, 64}); False: fp32 MFMA (A/B)
KNN_GLOBAL_LISTS = True  # round 6: 32 query rows per workgroup, candidate lists in the workspace (knn_filter2_kernel);
                         # False: rounds 4 - 5's 16-row kernel with LDS lists (A/B; same ids)
KNN_LIST_ENTRIES = 1024  # (key, id) pairs per point behind the planes of the workspace (csrc/knn_dense.hip kF2Cap)


# ----------------------------------------------------------------------------------------
# kNN
# ----------------------------------------------------------------------------------------
KNN_MAX_POINTS = 4096        # one sample's distance strip lives in LDS (csrc/knn_dense.hip)
KNN_MAX_NEIGHBOURS = 1024    # k * dilation; ResGCN-28's deepest block needs 16 * 27 = 432 (candidate-filter kernel up to
                             # 512), ResGCN-56's 16 * 55 = 880 (exact kernel)


def _knn_launch(x3: torch.Tensor, K: int, dilation: int, nn_out: torch.Tensor, ctr_out, exclude_self: bool = False):
    """x3: (B, C, N) fp32 view (any strides)."""
    lib = _lib.load()
    dev = _lib.require_device(x3)
    B, C, N = x3.shape
    if K > N - (1 if exclude_self else 0):
        raise ValueError(f"k*dilation = {K} neighbours asked of clouds with {N} points")
    if N > KNN_MAX_POINTS or K > KNN_MAX_NEIGHBOURS:
        return _knn_beyond_kernel_limits(x3, K, dilation, nn_out, ctr_out, exclude_self)
    ws_bytes = lib.dgcn_knn_dense_workspace_bytes(B, N, C if KNN_BF16_PIPE else 0) if (N >= 1024 and USE_KNN_FILTER) else 0
    if not KNN_GLOBAL_LISTS and KNN_BF16_PIPE and C in (32, 64) and ws_bytes:
        ws_bytes -= B * N * KNN_LIST_ENTRIES * 8            # a workspace without room for the lists selects the 16-row kernel
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8) if ws_bytes else None
    with _lib.device_ctx(dev):
        rc = lib.dgcn_knn_dense_f32(x3.data_ptr(), x3.stride(0), x3.stride(1), x3.stride(2), B, C, N, K,
                                    dilation, 1 if exclude_self else 0, nn_out.data_ptr(), _lib.ptr(ctr_out),
                                    _lib.ptr(ws), ws_bytes, _lib.current_stream_handle(dev))
    _lib.check(rc, "dgcn_knn_dense_f32")


def _knn_beyond_kernel_limits(x3, K, dilation, nn_out, ctr_out, exclude_self):
    """Clouds larger than the kernel serves (N > 4096 points or k*dilation > 1024): the reference's own formulation
    -- pairwise_distance, then the k*dilation smallest per row (gcn_lib/dense/torch_edge.py:27-76) -- on library ops,
    one sample at a time with the reference's (N, N) distance matrix.  Ties go to the lower index (stable sort), as in
    the kernel.  No reference configuration gets here (sem_seg_dense: 4096 points, k*d <= 432); it exists so that a
    bigger cloud runs instead of raising."""
    B, C, N = x3.shape
    centre = torch.arange(N, device=x3.device).view(N, 1)
    for b in range(B):
        p = x3[b].t().contiguous()                                  # (N, C)
        sq = (p * p).sum(1, keepdim=True)
        d = (sq + (-2.0) * (p @ p.t())) + sq.t()                    # the reference's association
        if exclude_self:
            d.fill_diagonal_(float("inf"))
        nn_out[b] = torch.sort(d, dim=1, stable=True).indices[:, :K:dilation]
        if ctr_out is not None:
            ctr_out[b] = centre
        del d


def knn_edge_index(x: torch.Tensor, k: int, dilation: int = 1, exclude_self: bool = False) -> torch.Tensor:
    """x (B,C,N,1) -> edge_index (2,B,N,k) int64: [0] = every `dilation`-th of the k*dilation nearest
    neighbours (ascending distance; self included unless ``exclude_self``), [1] = centre ids."""
    with torch.no_grad():
        x3 = x.detach()
        if x3.dtype != torch.float32:
            x3 = x3.float()
        x3 = x3.squeeze(-1) if x3.dim() == 4 else x3
        B, C, N = x3.shape
        K = k * dilation
        kout = (K + dilation - 1) // dilation
        ei = torch.empty(2, B, N, kout, dtype=torch.int64, device=x.device)
        _knn_launch(x3, K, dilation, ei[0], ei[1], exclude_self)
    _register_centres(ei)                                  # written by the kernel: edge_index[1][b, n, :] == n
    return ei


# ---- centre-id contract of the dense convolutions ------------------------------------------------------------
# The reference gathers x_i = batched_index_select(x, edge_index[1]) (gcn_lib/dense/torch_vertex.py:17,32); every
# graph builder of the library writes edge_index[1][b, n, :] = n, which is what the fused kernels assume (row n of
# the neighbour list belongs to point n).  Edge lists built here are known to satisfy it (their storage is
# registered, views such as edge_index[..., ::d] share it); a caller-supplied edge_index is verified ONCE per
# storage (one device reduction + host read) and rejected loudly if its centres differ.  An entry is keyed by
# (storage address, version counter) -- an in-place edit of a verified tensor is a new key -- and dies with the tensor
# object it was registered for, so a later tensor that the caching allocator places at the same address is checked again.
class _StorageSet:
    def __init__(self, cap=256):
        self._d, self._cap = {}, cap

    def add(self, key):
        if len(self._d) >= self._cap:
            self._d.pop(next(iter(self._d)))
        self._d[key] = self._d.get(key, 0) + 1

    def discard(self, key):
        n = self._d.get(key, 0)
        if n <= 1:
            self._d.pop(key, None)
        else:
            self._d[key] = n - 1

    def __contains__(self, key):
        return key in self._d


_CENTRES_OK = _StorageSet()


def _centres_key(t: torch.Tensor):
    return (t.untyped_storage().data_ptr(), t._version)


def _register_centres(t: torch.Tensor) -> None:
    key = _centres_key(t)
    _CENTRES_OK.add(key)
    weakref.finalize(t, _CENTRES_OK.discard, key)


def check_centres(edge_index: torch.Tensor) -> None:
    if _centres_key(edge_index) in _CENTRES_OK:
        return
    centre = edge_index[1]
    n = centre.size(-2)
    want = torch.arange(n, device=centre.device, dtype=centre.dtype).view(1, n, 1)
    if not bool((centre == want).all()):
        raise NotImplementedError(
            "dense EdgeConv2d / MRConv2d: edge_index[1][b, n, :] must equal n (the centre of neighbour row n is "
            "point n, as every gcn_lib.dense graph builder produces); general centre ids are not supported by the "
            "fused kernels")
    _register_centres(edge_index)


def knn_indices(pts: torch.Tensor, k: int, dilation: int = 1, exclude_self: bool = False) -> torch.Tensor:
    """pts (B,N,C) point-major features -> (B,N,k) int64 neighbour ids (sparse-layout callers)."""
    with torch.no_grad():
        p = pts.detach().float()
        B, N, C = p.shape
        K = k * dilation
        out = torch.empty(B, N, (K + dilation - 1) // dilation, dtype=torch.int64, device=pts.device)
        _knn_launch(p.permute(0, 2, 1), K, dilation, out, None, exclude_self)
    return out


# ----------------------------------------------------------------------------------------
# per-vertex GEMM (fp32 MFMA forward; the backward GEMMs are plain library matmuls)
# ----------------------------------------------------------------------------------------
class _VertexGemm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, bias):
        lib = _lib.load()
        dev = _lib.require_device(x, W, bias)
        x3 = x.squeeze(-1) if x.dim() == 4 else x
        if x3.dtype != torch.float32:
            x3 = x3.float()
        B, C, N = x3.shape
        Wc = W.float().contiguous()
        M = Wc.size(1)
        bc = None if bias is None else bias.float().contiguous()
        out = torch.empty(B, N, M, device=dev, dtype=torch.float32)
        with _lib.device_ctx(dev):
            rc = lib.dgcn_vertex_gemm_f32(x3.data_ptr(), x3.stride(0), x3.stride(1), x3.stride(2), B, C, N,
                                          Wc.data_ptr(), _lib.ptr(bc), M, out.data_ptr(),
                                          _lib.current_stream_handle(dev))
        _lib.check(rc, "dgcn_vertex_gemm_f32")
        ctx.save_for_backward(x3, Wc)
        ctx.x_shape = x.shape
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, gout):
        x3, Wc = ctx.saved_tensors
        gx = gW = gb = None
        g2 = gout.reshape(-1, gout.size(-1))                               # (B*N, M)
        if ctx.needs_input_grad[0]:
            gx = torch.matmul(gout, Wc.t()).permute(0, 2, 1).reshape(ctx.x_shape)   # (B,C,N[,1])
        if ctx.needs_input_grad[1]:
            xt = x3.permute(1, 0, 2).reshape(x3.size(1), -1)               # (C, B*N)
            gW = torch.matmul(xt, g2)                                       # (C, M)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum(0)
        return gx, gW, gb


def vertex_gemm(x: torch.Tensor, W: torch.Tensor, bias=None) -> torch.Tensor:
    """out[b,n,m] = sum_c x[b,c,n] * W[c,m] + bias[m];  x (B,C,N,1) or (B,C,N), W (C,M)."""
    return _VertexGemm.apply(x, W, bias)


# ----------------------------------------------------------------------------------------
# neighbourhood reduction
# ----------------------------------------------------------------------------------------
class _EdgeReduce(torch.autograd.Function):
    """(vmax, vmin, sum_a, sum_a2) of a_{bnl} = act(P[b,n] + Q[b, idx[b,n,l]]).
    PQ is (B,N,2C) [P | Q] when has_p else (B,N,C) = Q.  sum_* are float64 (C,) tensors."""

    @staticmethod
    def forward(ctx, PQ, idx, has_p: bool, act: int, slope: float, need_min: bool, need_stats: bool, track: bool):
        lib = _lib.load()
        dev = _lib.require_device(PQ, idx)
        PQ = PQ.float().contiguous()
        B, N, W = PQ.shape
        C = W // 2 if has_p else W
        k = idx.size(-1)
        if idx.dtype != torch.int64:
            idx = idx.long()
        need_bwd = track and ctx.needs_input_grad[0]
        vmax = torch.empty(B, N, C, device=dev, dtype=torch.float32)
        vmin = torch.empty_like(vmax) if need_min else None
        amax = torch.empty(B, N, C, device=dev, dtype=torch.uint8) if need_bwd else None
        amin = torch.empty(B, N, C, device=dev, dtype=torch.uint8) if (need_bwd and need_min) else None
        stats = None
        if need_stats:
            nparts = lib.dgcn_dense_edge_reduce_num_partials(B, N, C)
            stats = torch.empty(nparts, 2, C, device=dev, dtype=torch.float32)
        p_ptr = PQ.data_ptr() if has_p else None
        q_ptr = PQ.data_ptr() + (C * 4 if has_p else 0)
        with _lib.device_ctx(dev):
            rc = lib.dgcn_dense_edge_reduce_fwd_f32(
                p_ptr, W, q_ptr, W, idx.data_ptr(), idx.stride(0), idx.stride(1), idx.stride(2),
                B, N, C, k, act, slope, vmax.data_ptr(), _lib.ptr(vmin), _lib.ptr(amax), _lib.ptr(amin),
                _lib.ptr(stats), _lib.current_stream_handle(dev))
        _lib.check(rc, "dgcn_dense_edge_reduce_fwd_f32")
        if need_stats:
            tot = stats.double().sum(0)
            s1, s2 = tot[0], tot[1]
        else:
            s1 = s2 = torch.zeros(C, device=dev, dtype=torch.float64)
        if vmin is None:
            vmin = vmax.new_zeros(())
        if need_bwd:
            ctx.save_for_backward(PQ, idx, amax, amin)
            ctx.cfg = (has_p, act, slope, need_min, need_stats, C, k)
        ctx.mark_non_differentiable(*([] if need_min else [vmin]))
        return vmax, vmin, s1, s2

    @staticmethod
    def backward(ctx, gmax, gmin, gs1, gs2):
        lib = _lib.load()
        PQ, idx, amax, amin = ctx.saved_tensors
        has_p, act, slope, need_min, need_stats, C, k = ctx.cfg
        dev = PQ.device
        B, N, W = PQ.shape
        gmax = gmax.float().contiguous()
        gmin_c = gmin.float().contiguous() if (need_min and gmin is not None) else None
        gs = gs1.float().contiguous() if (need_stats and gs1 is not None) else None
        gq = gs2.float().contiguous() if (need_stats and gs2 is not None) else None
        p_ptr = PQ.data_ptr() if has_p else None
        q_ptr = PQ.data_ptr() + (C * 4 if has_p else 0)
        if USE_INVERSE_LISTS:                           # dQ through inverse neighbour lists (no float atomics)
            dPQ = torch.empty_like(PQ)
            ws_bytes = lib.dgcn_dense_edge_reduce_bwd_inv_workspace_bytes(B, N, C, k)
            ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
            with _lib.device_ctx(dev):
                rc = lib.dgcn_dense_edge_reduce_bwd_inv_f32(
                    p_ptr, W, q_ptr, W, idx.data_ptr(), idx.stride(0), idx.stride(1), idx.stride(2),
                    B, N, C, k, act, slope, amax.data_ptr(), _lib.ptr(amin if gmin_c is not None else None),
                    gmax.data_ptr(), _lib.ptr(gmin_c), _lib.ptr(gs), _lib.ptr(gq), None,
                    dPQ.data_ptr() if has_p else None, dPQ.data_ptr() + (C * 4 if has_p else 0),
                    ws.data_ptr(), ws_bytes, _lib.current_stream_handle(dev))
            _lib.check(rc, "dgcn_dense_edge_reduce_bwd_inv_f32")
            return dPQ, None, None, None, None, None, None, None
        nsplit = lib.dgcn_dense_edge_reduce_bwd_nsplit(B, N, C)
        parts = None
        if nsplit > 0:                                  # LDS-accumulated dQ partials, no global atomics
            parts = torch.empty(nsplit, B, N, C, device=dev, dtype=torch.float32)
            dPQ = torch.empty_like(PQ)
        else:
            dPQ = torch.zeros_like(PQ)                  # dQ half is accumulated with global atomics
        dp_ptr = dPQ.data_ptr() if has_p else None
        dq_ptr = dPQ.data_ptr() + (C * 4 if has_p else 0)
        with _lib.device_ctx(dev):
            rc = lib.dgcn_dense_edge_reduce_bwd_f32(
                p_ptr, W, q_ptr, W, idx.data_ptr(), idx.stride(0), idx.stride(1), idx.stride(2),
                B, N, C, k, act, slope, amax.data_ptr(), _lib.ptr(amin if gmin_c is not None else None),
                gmax.data_ptr(), _lib.ptr(gmin_c), _lib.ptr(gs), _lib.ptr(gq), None, dp_ptr, dq_ptr,
                _lib.ptr(parts), nsplit, _lib.current_stream_handle(dev))
        _lib.check(rc, "dgcn_dense_edge_reduce_bwd_f32")
        if parts is not None:
            dq = parts[0] if nsplit == 1 else parts.sum(0)
            if has_p:
                dPQ[..., C:] = dq
            else:
                dPQ = dq
        return dPQ, None, None, None, None, None, None, None


def edge_reduce(PQ: torch.Tensor, idx: torch.Tensor, has_p: bool, act: int = ACT_NONE, slope: float = 0.2,
                need_min: bool = False, need_stats: bool = False):
    """Returns (vmax, vmin|None, sum_a, sum_a2).  Channel counts that are not a multiple of 4 are
    zero-padded for the kernel's float4 lanes and sliced back."""
    B, N, W = PQ.shape
    C = W // 2 if has_p else W
    pad = (-C) % 4
    if pad:
        if has_p:
            P, Q = PQ[..., :C], PQ[..., C:]
            PQ = torch.cat([F.pad(P, (0, pad)), F.pad(Q, (0, pad))], dim=-1)
        else:
            PQ = F.pad(PQ, (0, pad))
    vmax, vmin, s1, s2 = _EdgeReduce.apply(PQ, idx, has_p, act, float(slope), need_min, need_stats,
                                           torch.is_grad_enabled())
    if pad:
        vmax = vmax[..., :C]
        vmin = vmin[..., :C] if need_min else vmin
        s1, s2 = s1[:C], s2[:C]
    return vmax, (vmin if need_min else None), s1, s2


# ----------------------------------------------------------------------------------------
# fused dense EdgeConv2d (conv -> act -> [BatchNorm2d] -> max over neighbours) in a handful of launches
# ----------------------------------------------------------------------------------------
BN_NONE, BN_TRAIN, BN_EVAL = 0, 1, 2


from .nn_util import splitk_xt_g as _splitk_xt_g  # noqa: E402  (tall-skinny g^T x as batched split-K)


class _EdgeConv2dFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, idx, act, slope, bn_mode, running_mean, running_var,
                num_batches, momentum, eps, track, res_scale=None):
        lib = _lib.load()
        dev = _lib.require_device(x, weight, idx)
        stream = _lib.current_stream_handle(dev)
        x3 = x.squeeze(-1) if x.dim() == 4 else x
        if x3.dtype != torch.float32:
            x3 = x3.float()
        B, C, N = x3.shape
        Cout = weight.size(0)
        W2 = weight.reshape(Cout, 2 * C).float().contiguous()
        bc = None if bias is None else bias.float().contiguous()
        if idx.dtype != torch.int64:
            idx = idx.long()
        k = idx.size(-1)
        need_bwd = track and any(ctx.needs_input_grad[:5])
        has_bn = bn_mode != BN_NONE
        pq = torch.empty(B, N, 2 * Cout, device=dev, dtype=torch.float32)
        vmax = torch.empty(B, N, Cout, device=dev, dtype=torch.float32)
        vmin = torch.empty_like(vmax) if has_bn else None
        amax = torch.empty(B, N, Cout, device=dev, dtype=torch.uint8) if need_bwd else None
        amin = torch.empty(B, N, Cout, device=dev, dtype=torch.uint8) if (need_bwd and has_bn) else None
        stats = None
        nparts = 0
        if bn_mode == BN_TRAIN:
            nparts = lib.dgcn_dense_edge_reduce_num_partials(B, N, Cout)
            stats = torch.empty(nparts, 2, Cout, device=dev, dtype=torch.float32)
        out = torch.empty(B, Cout, N, 1, device=dev, dtype=torch.float32)
        bnbuf = torch.empty(4, Cout, device=dev, dtype=torch.float32) if has_bn else None
        count = float(B) * N * k
        with _lib.device_ctx(dev):
            _lib.check(lib.dgcn_edgeconv_pq_f32(x3.data_ptr(), x3.stride(0), x3.stride(1), x3.stride(2), B, C, N,
                                                W2.data_ptr(), _lib.ptr(bc), Cout, pq.data_ptr(), stream),
                       "dgcn_edgeconv_pq_f32")
            _lib.check(lib.dgcn_dense_edge_reduce_fwd_f32(
                pq.data_ptr(), 2 * Cout, pq.data_ptr() + 4 * Cout, 2 * Cout, idx.data_ptr(), idx.stride(0),
                idx.stride(1), idx.stride(2), B, N, Cout, k, act, slope, vmax.data_ptr(), _lib.ptr(vmin),
                _lib.ptr(amax), _lib.ptr(amin), _lib.ptr(stats), stream), "dgcn_dense_edge_reduce_fwd_f32")
            if has_bn:
                _lib.check(lib.dgcn_bn_finalize_f32(
                    _lib.ptr(stats), nparts, Cout, count, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean),
                    _lib.ptr(running_var), _lib.ptr(num_batches), 1 if bn_mode == BN_TRAIN else 0,
                    float(momentum), float(eps), bnbuf.data_ptr(), stream), "dgcn_bn_finalize_f32")
            if res_scale is None:
                _lib.check(lib.dgcn_bn_apply_f32(vmax.data_ptr(), _lib.ptr(vmin), _lib.ptr(bnbuf), out.data_ptr(),
                                                 B, N, Cout, stream), "dgcn_bn_apply_f32")
            else:
                # the block's skip connection rides in the transposing store: out = conv(x) + res_scale * x
                if Cout != C:
                    raise ValueError("residual needs in_channels == out_channels")
                _lib.check(lib.dgcn_bn_apply_res_f32(vmax.data_ptr(), _lib.ptr(vmin), _lib.ptr(bnbuf), x3.data_ptr(),
                                                     x3.stride(0), x3.stride(1), x3.stride(2), float(res_scale),
                                                     out.data_ptr(), B, N, Cout, stream), "dgcn_bn_apply_res_f32")
        if need_bwd:
            ctx.save_for_backward(x3, W2, pq, idx, amax, amin, vmax, vmin, bnbuf, gamma)
            ctx.cfg = (act, slope, bn_mode, count, tuple(x.shape), tuple(weight.shape), bias is not None)
            ctx.res_scale = res_scale
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x3, W2, pq, idx, amax, amin, vmax, vmin, bnbuf, gamma = ctx.saved_tensors
        act, slope, bn_mode, count, x_shape, w_shape, has_bias = ctx.cfg
        dev = x3.device
        stream = _lib.current_stream_handle(dev)
        B, C, N = x3.shape
        Cout = W2.size(0)
        k = idx.size(-1)
        has_bn = bn_mode != BN_NONE
        g3 = g.squeeze(-1) if g.dim() == 4 else g
        if g3.dtype != torch.float32:
            g3 = g3.float()
        gsel = torch.empty(B, N, Cout, device=dev, dtype=torch.float32)
        nparts = lib.dgcn_bn_bwd_num_partials(B, N)
        partial = torch.empty(nparts, 2, Cout, device=dev, dtype=torch.float32) if has_bn else None
        coef = torch.empty(4, Cout, device=dev, dtype=torch.float32) if has_bn else None
        dPQ = torch.empty(B, N, 2 * Cout, device=dev, dtype=torch.float32)
        inv_ws = None
        nsplit, parts = 0, None
        if USE_INVERSE_LISTS:
            inv_bytes = lib.dgcn_dense_edge_reduce_bwd_inv_workspace_bytes(B, N, Cout, k)
            inv_ws = torch.empty(inv_bytes, device=dev, dtype=torch.uint8)
        else:
            nsplit = lib.dgcn_dense_edge_reduce_bwd_nsplit(B, N, Cout)
            parts = torch.empty(nsplit, B, N, Cout, device=dev, dtype=torch.float32) if nsplit > 0 else None
            if parts is None:
                dPQ[..., Cout:].zero_()
        train_stats = bn_mode == BN_TRAIN
        with _lib.device_ctx(dev):
            _lib.check(lib.dgcn_bn_bwd_prep_f32(g3.data_ptr(), g3.stride(0), g3.stride(1), g3.stride(2), vmax.data_ptr(),
                                                _lib.ptr(vmin), _lib.ptr(bnbuf), gsel.data_ptr(), _lib.ptr(partial),
                                                B, N, Cout, stream), "dgcn_bn_bwd_prep_f32")
            if has_bn:
                _lib.check(lib.dgcn_bn_bwd_finalize_f32(partial.data_ptr(), nparts, Cout, count, _lib.ptr(gamma),
                                                        bnbuf.data_ptr(), 1 if train_stats else 0, coef.data_ptr(),
                                                        stream), "dgcn_bn_bwd_finalize_f32")
            gs_ptr = coef[2].data_ptr() if train_stats else None
            gq_ptr = coef[3].data_ptr() if train_stats else None
            if inv_ws is not None:
                _lib.check(lib.dgcn_dense_edge_reduce_bwd_inv_f32(
                    pq.data_ptr(), 2 * Cout, pq.data_ptr() + 4 * Cout, 2 * Cout, idx.data_ptr(), idx.stride(0),
                    idx.stride(1), idx.stride(2), B, N, Cout, k, act, slope, amax.data_ptr(), _lib.ptr(amin),
                    gsel.data_ptr(), None, gs_ptr, gq_ptr, bnbuf.data_ptr() if has_bn else None,
                    dPQ.data_ptr(), dPQ.data_ptr() + 4 * Cout, inv_ws.data_ptr(), inv_bytes, stream),
                    "dgcn_dense_edge_reduce_bwd_inv_f32")
            else:
                _lib.check(lib.dgcn_dense_edge_reduce_bwd_f32(
                    pq.data_ptr(), 2 * Cout, pq.data_ptr() + 4 * Cout, 2 * Cout, idx.data_ptr(), idx.stride(0),
                    idx.stride(1), idx.stride(2), B, N, Cout, k, act, slope, amax.data_ptr(), _lib.ptr(amin),
                    gsel.data_ptr(), None, gs_ptr, gq_ptr, bnbuf.data_ptr() if has_bn else None,
                    dPQ.data_ptr(), dPQ.data_ptr() + 4 * Cout, _lib.ptr(parts), nsplit, stream),
                    "dgcn_dense_edge_reduce_bwd_f32")
            if parts is not None:
                _lib.check(lib.dgcn_reduce_parts_f32(parts.data_ptr(), nsplit, B * N, Cout,
                                                     dPQ.data_ptr() + 4 * Cout, 2 * Cout, stream),
                           "dgcn_reduce_parts_f32")
        gx = gW = gb = ggamma = gbeta = None
        if EDGECONV_BWD_KERNELS:
            want_w = ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])
            with _lib.device_ctx(dev):
                if ctx.needs_input_grad[0]:
                    gx3 = torch.empty(B, C, N, device=dev, dtype=torch.float32)
                    res = ctx.res_scale is not None
                    _lib.check(lib.dgcn_edgeconv_bwd_input_f32(
                        dPQ.data_ptr(), W2.data_ptr(), g3.data_ptr() if res else None, g3.stride(0), g3.stride(1),
                        g3.stride(2), float(ctx.res_scale) if res else 0.0, B, C, N, Cout, gx3.data_ptr(), stream),
                        "dgcn_edgeconv_bwd_input_f32")
                    gx = gx3.reshape(x_shape)
                if want_w:
                    wparts = torch.empty(lib.dgcn_edgeconv_bwd_weight_num_partials(B, N), Cout * 2 * C + Cout,
                                         device=dev, dtype=torch.float32)
                    _lib.check(lib.dgcn_edgeconv_bwd_weight_f32(
                        dPQ.data_ptr(), x3.data_ptr(), x3.stride(0), x3.stride(1), x3.stride(2), B, C, N, Cout,
                        wparts.data_ptr(), stream), "dgcn_edgeconv_bwd_weight_f32")
            if want_w:
                wsum = _lib.sum_partials(wparts)             # [dW1 | dW2] in the weight's layout, then db
                if ctx.needs_input_grad[1]:
                    gW = wsum[:Cout * 2 * C].view(w_shape)
                if has_bias and ctx.needs_input_grad[2]:
                    gb = wsum[Cout * 2 * C:]
            if has_bn and gamma is not None:
                if ctx.needs_input_grad[3]:
                    ggamma = coef[0]
                if ctx.needs_input_grad[4]:
                    gbeta = coef[1]
            return (gx, gW, gb, ggamma, gbeta) + (None,) * 11
        w1, w2h = W2[:, :C], W2[:, C:]
        if ctx.needs_input_grad[0]:
            wc = torch.cat([w1 - w2h, w2h], dim=0)                          # (2Cout, C)
            # (B, C, N) straight out of the GEMM (transposed operands, no permute + copy); the skip connection's
            # gradient res_scale * g is the GEMM's beta term
            wct = wc.t().unsqueeze(0).expand(B, C, 2 * Cout)
            if ctx.res_scale is None:
                gx = torch.bmm(wct, dPQ.transpose(1, 2)).reshape(x_shape)
            else:
                gx = torch.baddbmm(g3, wct, dPQ.transpose(1, 2), beta=float(ctx.res_scale)).reshape(x_shape)
        if ctx.needs_input_grad[1]:
            xr = x3.permute(0, 2, 1).reshape(B * N, C)
            dwc = _splitk_xt_g(dPQ.view(B * N, 2 * Cout), xr)               # (2Cout, C)
            gW = torch.cat([dwc[:Cout], dwc[Cout:] - dwc[:Cout]], dim=1).reshape(w_shape)
        if has_bias and ctx.needs_input_grad[2]:
            gb = dPQ[..., :Cout].sum((0, 1))
        if has_bn and gamma is not None:
            if ctx.needs_input_grad[3]:
                ggamma = coef[0]
            if ctx.needs_input_grad[4]:
                gbeta = coef[1]
        return (gx, gW, gb, ggamma, gbeta) + (None,) * 11


def edgeconv2d_fused(x, weight, bias, idx, act, slope, bn=None, res_scale=None):
    """max_l BN(act(W [x_i ; x_j - x_i] + b)) for x (B,C,N,1), idx (B,N,k) -> (B,Cout,N,1).
    ``bn``: a BatchNorm2d module (training or eval semantics, running statistics updated in place) or None.
    ``res_scale``: add ``res_scale * x`` (ResDynBlock2d's skip connection) in the final store; its gradient joins the
    input-gradient GEMM."""
    if bn is None:
        return _EdgeConv2dFused.apply(x, weight, bias, None, None, idx, act, float(slope), BN_NONE, None, None,
                                      None, 0.0, 0.0, torch.is_grad_enabled(), res_scale)
    use_batch = bn.training or not bn.track_running_stats
    momentum = bn.momentum
    nbt = bn.num_batches_tracked if (use_batch and bn.track_running_stats and bn.training) else None
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    if use_batch and not bn.training:
        rm = rv = None                                   # batch statistics without touching the buffers
    if nbt is not None and momentum is None:             # cumulative moving average: needs the count (host sync,
        bn.num_batches_tracked += 1                      # exactly like torch.nn.BatchNorm2d itself)
        momentum = 1.0 / float(bn.num_batches_tracked)
        nbt = None
    return _EdgeConv2dFused.apply(x, weight, bias, bn.weight, bn.bias, idx, act, float(slope),
                                  BN_TRAIN if use_batch else BN_EVAL, rm, rv, nbt,
                                  0.0 if momentum is None else float(momentum), float(bn.eps),
                                  torch.is_grad_enabled(), res_scale)
