"""Autograd-aware wrappers around the libdgcn C ABI (sparse aggregation).

``gen_aggregate`` is the fused replacement of the reference's
``propagate -> message -> aggregate`` chain
(gcn_lib/sparse/torch_vertex.py:68,78-85; gcn_lib/sparse/torch_message.py:44-85).
All arithmetic on E-sized data happens inside the HIP kernels; the few N-sized
coefficient tensors (gradient pre-scaling, d/dt, d/dp reductions) are plain torch ops.
"""
from __future__ import annotations

from typing import Optional, Union

import torch

from . import _lib
from .graph import Graph, graph_of

_MODES = {
    "add": _lib.AGGR_ADD, "sum": _lib.AGGR_ADD, "mean": _lib.AGGR_MEAN, "max": _lib.AGGR_MAX,
    "softmax": _lib.AGGR_SOFTMAX, "softmax_sg": _lib.AGGR_SOFTMAX, "softmax_sum": _lib.AGGR_SOFTMAX,
    "power": _lib.AGGR_POWER, "power_sum": _lib.AGGR_POWER,
}
POW_LO, POW_HI = 1e-7, 1e1  # gcn_lib/sparse/torch_message.py:69
ENC_FEATURES = 8                   # raw edge features of the per-edge encoder (kEncF in csrc/gen_aggr_common.h)
SINGLE_GATHER_SOFTMAX_BWD = True   # halves the backward's gather traffic when the log-sum-exp range allows
SHIFT_SAFE_ABS_L = 80.0            # the forward kernel flags |L_i| >= 80 (kShiftSafe in csrc/gen_aggr_common.h)
ENC_STATIC_ITEMS = False           # per-edge encoder kernels: True = work items dealt by wave index (DGCN_FLAG_STATIC_ITEMS):
                                   # bit-reproducible dW | db at 1.3 - 1.6x the launch time; default = items claimed from
                                   # device-side counters (outputs and grad_x identical, dW | db equal to rounding)
FUSED_EDGE_GEMM = True             # wide edge features (Linear(hidden -> C) per layer): GEMM + aggregation in one kernel
ENC_MAX_WINNER_BWD = True          # per-edge encoder under max: dW' | db' from the (row, channel) arg-max winners (n_dst * C gathers
                                   # of 32 bytes) + the plain CSC walk for grad_x, instead of the per-edge encoder walk: 0.296 ->
                                   # 0.254 ms per layer.  Round 4 shipped it OFF: a replayed hipGraph faulted with it.  Root cause
                                   # (round 5): the work-item counters were re-armed by a memset NODE that a replay
                                   # ran before the previous launch had drained -- items 0 .. 2 were lost, their arg-max ids stale,
                                   # and this route used them as addresses.  Counters are now zeroed by a kernel, ids range-checked
EGEMM_MAX_WINNER_BWD = True        # its backward under max: walk the (row, channel) winners (csrc/egemm_max_bwd.hip) instead of
                                   # writing dz (E, C) and running dz @ W, dz^T F over it; False = that dense route (A/B)
                                   # (csrc/gen_aggr_egemm.hip); False = stock GEMM + (E, C) embedding (A/B benchmarks)
MAX_MASK_MIN_TABLE_BYTES = 128 << 20   # max backward through per-edge arg-max bit masks (two launches) when the (n_dst, C)
                                   # arg-max table is at least this big, i.e. falls out of the 256 MiB Infinity Cache
                                   # (products: 28.3 -> 23.7 ms per step); cache-resident graphs keep the one-launch
                                   # walk over gathered arg-max rows (arxiv shape with locality: 0.44 vs 0.53 ms)


_OPTION_NAMES = ("SINGLE_GATHER_SOFTMAX_BWD", "ENC_STATIC_ITEMS", "FUSED_EDGE_GEMM", "ENC_MAX_WINNER_BWD",
                 "EGEMM_MAX_WINNER_BWD", "MAX_MASK_MIN_TABLE_BYTES")


class options:
    """``with ops.options(enc_max_winner_bwd=True, ...):`` -- the switches above for the aggregation calls of THIS thread.

    The module-level names are the process defaults (set them before the model runs).  A call reads its switches ONCE,
    in its forward, on the calling thread -- the defaults overlaid by the innermost ``options`` block of that thread --
    and its backward (which autograd runs on another thread, possibly while another replica's host thread is inside its
    own block: nn.DataParallel) uses that snapshot, never the globals."""

    def __init__(self, **kw):
        self._kw = {}
        for k, v in kw.items():
            name = k.upper()
            if name not in _OPTION_NAMES:
                raise TypeError(f"unknown option {k!r} (known: {', '.join(n.lower() for n in _OPTION_NAMES)})")
            self._kw[name] = v

    def __enter__(self):
        # the previous overrides go on a PER-THREAD stack, not on this object: one captured_options() object is re-entered
        # from autograd's thread (checkpoint recomputation, reversible layers), possibly nested or concurrently
        # (retain_graph double backward, DataParallel replicas sharing a closure) -- ADVICE r5
        prev = getattr(_TLS, "options", None)
        stack = getattr(_TLS, "opt_stack", None)
        if stack is None:
            stack = _TLS.opt_stack = []
        stack.append(prev)
        merged = dict(prev or {})
        merged.update(self._kw)
        _TLS.options = merged
        return self

    def __exit__(self, *exc):
        _TLS.options = _TLS.opt_stack.pop()
        return False


def captured_options() -> "options":
    """The calling thread's overrides as a context object that can be re-entered later / on another thread (the
    recomputation of a checkpointed or reversible layer runs inside the backward pass, on autograd's thread)."""
    o = options()
    o._kw = dict(getattr(_TLS, "options", None) or {})
    return o


def current_options() -> dict:
    """Snapshot of the switches for a call made now on this thread."""
    g = globals()
    snap = {n: g[n] for n in _OPTION_NAMES}
    over = getattr(_TLS, "options", None)
    if over:
        snap.update(over)
    return snap


def _scalar_arg(v):
    """(host float, device pointer or None) for a python float or a 1-element device tensor."""
    if isinstance(v, torch.Tensor):
        return 0.0, v
    return float(v), None


def _rows_f32(x: torch.Tensor) -> torch.Tensor:
    """fp32, unit channel stride (row stride may be anything >= C)."""
    if x.dtype != torch.float32:
        x = x.float()  # autocast / half inputs are computed in fp32
    if x.dim() != 2:
        raise ValueError("expected a (rows, channels) tensor")
    if x.stride(1) != 1 or x.stride(0) < x.size(1):
        x = x.contiguous()
    return x


def _scalar_f32(v: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if v is None or (v.dtype == torch.float32 and v.is_contiguous()):
        return v
    return v.detach().float().contiguous()


def _feat_rows(f: torch.Tensor) -> torch.Tensor:
    """Edge-feature rows for the fused edge GEMM: fp32, unit column stride, 16-byte aligned rows.  A torch.chunk view
    of the model-level embedding ((E, hidden) inside (E, hidden * group), model_rev.py:98-99) is consumed in place."""
    if f.dtype != torch.float32:
        f = f.float()
    if f.stride(1) != 1 or f.stride(0) % 4 != 0 or f.data_ptr() % 16 != 0 or f.stride(0) < f.size(1):
        f = f.contiguous()
    return f


# ---- in-place accumulation of edge-feature gradients ------------------------------------------------------------------
# The reference's deep models hand ONE (E, hidden) edge embedding to every layer (ogbn_proteins/model.py:116-127,
# ogb_eff/ogbn_proteins/model_rev.py:98-107); autograd then materialises an (E, hidden) gradient per layer and adds
# them up.  A caller that owns a running sum can attach it to the feature tensor it is about to push through a layer:
# the fused edge-GEMM backward of that call then ADDS its ``dz @ W`` into the buffer (one GEMM with beta = 1) and reports
# no gradient.  The buffer travels on the tensor OBJECT and is captured by the autograd context of the call: no
# module-level state, safe with one host thread per GPU.
_SINK_ATTR = "_dgcn_grad_sink"


class AggregationStash:
    """Outputs of the aggregation launches of one checkpointed function (blocks.res_plus_layer).

    torch.utils.checkpoint runs a function under no_grad and again, with grad enabled, inside the backward.  The
    aggregation is by far the most expensive part of that function and its results are a few (N, C) arrays: recorded in
    the first pass (``mode = "record"``: the launch also writes what the backward needs -- log-sum-exp, arg-max ids,
    pre-activations), handed out again in the second (``"replay"``: no launch).  The node-wise part is recomputed as
    torch.utils.checkpoint intends.  The recorded tensors are the ones the first pass returned: they must not be written
    in place by the caller (GENConv does not)."""

    def __init__(self, node_sized_only: bool = False):
        self.items = []
        self.extra = {}        # small by-products of the first pass that the second reuses by key (blocks._ComposeEncoder)
        self.pos = 0
        self.mode = None
        # True: an aggregation whose backward needs an (E, C) array of the forward (the fused edge encoder's
        # pre-activations under softmax / power) is not kept -- its slot says "launch again"
        self.node_sized_only = node_sized_only


# The active stash belongs to the THREAD that entered ``stash_aggregation``: the record pass runs on the caller's
# thread, the replay on the device's autograd thread (reentrant checkpoint / reversible backward), and with one host
# thread per GPU (nn.DataParallel) several passes are in flight at once -- a module-level variable would let one
# thread's ``with`` block install or restore another thread's stash.
import threading

_TLS = threading.local()


def _active_stash() -> Optional[AggregationStash]:
    return getattr(_TLS, "stash", None)


class stash_aggregation:
    """``with stash_aggregation(stash, "record" | "replay"):`` around the function a checkpoint runs twice."""

    def __init__(self, stash: AggregationStash, mode: str):
        if mode not in ("record", "replay"):
            raise ValueError("mode must be 'record' or 'replay'")
        self.stash, self.mode = stash, mode

    def __enter__(self):
        self._prev = _active_stash()
        self.stash.mode = self.mode
        if self.mode == "replay":
            self.stash.pos = 0
        else:
            self.stash.items.clear()
            self.stash.extra.clear()
        _TLS.stash = self.stash
        return self.stash

    def __exit__(self, *exc):
        _TLS.stash = self._prev
        self.stash.mode = None
        return False


class edge_grad_sink:
    """``with edge_grad_sink(feat, buffer): out = layer(..., feat, ...); autograd.grad(out, ...)``: gradients of ``feat``
    (the very tensor object handed to the layer, an (E, F) edge-feature tensor) produced by the fused edge-GEMM backward
    are accumulated into ``buffer`` (same shape, fp32) instead of being returned to autograd."""

    def __init__(self, feat: torch.Tensor, buffer: torch.Tensor):
        if buffer.shape != feat.shape or buffer.dtype != torch.float32 or not buffer.is_contiguous():
            raise ValueError("edge_grad_sink: buffer must be a contiguous fp32 tensor with the shape of the feature tensor")
        self._feat, self._buf = feat, buffer

    def __enter__(self):
        setattr(self._feat, _SINK_ATTR, self._buf)
        return self

    def __exit__(self, *exc):
        if hasattr(self._feat, _SINK_ATTR):
            delattr(self._feat, _SINK_ATTR)
        return False


_ZEROS = {}


def _zeros_cached(dev, n):
    """A read-only zero vector per (device, length): the per-channel shift of the single-gather softmax backward."""
    key = (dev.index, n)
    z = _ZEROS.get(key)
    if z is None:
        z = _ZEROS[key] = torch.zeros(n, device=dev, dtype=torch.float32)
    return z


class _GenAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, edge_attr, t_param, p_param, graph: Graph, mode: int, msg: int,
                eps: float, t_val: float, p_val: float, learn_t: bool, learn_p: bool, track: bool,
                add_root: bool = False, enc_feat=None, enc_w=None, enc_b=None):
        lib = _lib.load()
        dev = _lib.require_device(x, edge_attr, t_param, p_param, enc_feat, enc_w, enc_b)
        # the kernels read t / p as fp32 scalars through raw pointers: a model cast with .half() / .bfloat16()
        # keeps its Parameter dtype, the kernels get an fp32 copy
        t_dtype = None if t_param is None else t_param.dtype
        p_dtype = None if p_param is None else p_param.dtype
        t_param = _scalar_f32(t_param)
        p_param = _scalar_f32(p_param)
        if dev != graph.device:
            raise RuntimeError("graph and features live on different devices")
        x = _rows_f32(x)
        C = x.size(1)
        if x.size(0) != graph.n_src:
            raise ValueError(f"x has {x.size(0)} rows, graph expects {graph.n_src}")
        if edge_attr is not None:
            edge_attr = edge_attr.float().contiguous()
            if edge_attr.shape != (graph.n_edges, C):
                raise ValueError("edge_attr must be (E, C) matching x's channels")
        enc = enc_feat is not None
        egemm = False
        grad_sink = getattr(enc_feat, _SINK_ATTR, None) if enc else None     # see edge_grad_sink
        if enc:
            if edge_attr is not None:
                raise ValueError("pass either edge_attr (E, C) or the raw features + encoder, not both")
            n_feat = enc_feat.size(1)
            enc_w = enc_w.float().contiguous()
            enc_b = None if enc_b is None else enc_b.float().contiguous()
            if enc_feat.dim() != 2 or enc_feat.size(0) != graph.n_edges or enc_w.shape != (C, n_feat):
                raise ValueError("fused edge encoder: features (E, F), weight (C, F)")
            if n_feat == ENC_FEATURES and C % 4 == 0 and C <= 256:
                # narrow features: every edge recomputes W f_e + b from its 32 bytes (csrc/gen_aggr_common.h, EA == 2)
                if track and ctx.needs_input_grad[14]:
                    #  the per-edge kernels produce dW | db only: refuse rather than return no gradient
                    raise ValueError("the per-edge encoder path has no gradient w.r.t. the raw edge features: pass them "
                                     "detached (blocks.ComposedEdgeEmbedding does) or encode them first")
                enc_feat = enc_feat.float().contiguous()
            elif graph.n_edges > 0 and lib.dgcn_gen_aggr_egemm_supported(n_feat, C):
                egemm = True
                enc_feat = _feat_rows(enc_feat)
            else:
                raise ValueError(f"fused edge encoder: unsupported shape F={n_feat}, C={C} (see encoder_fusable)")
        opts = current_options()
        ctx.opts = opts
        need_grad = track and (any(ctx.needs_input_grad[:4]) or any(ctx.needs_input_grad[14:17]))
        # (no_grad / inverse passes skip the saved aux)
        stash = _active_stash()
        record = stash is not None and stash.mode == "record"
        replay = False
        # slot identity: position in the recorded sequence + everything that shapes the launch (two layers of one width
        # differ by position only, so EVERY aggregation call of the replayed function consumes its slot, whether or not
        # it needs a gradient this time)
        slot_key = (graph.n_dst, graph.n_src, graph.n_edges, C, mode, msg, egemm, enc, learn_t, learn_p, add_root)
        if stash is not None and stash.mode == "replay":
            if stash.pos >= len(stash.items):
                raise RuntimeError("aggregation stash: the recomputation runs more aggregations than the recorded pass")
            key, kept = stash.items[stash.pos]
            stash.pos += 1
            if key != slot_key:
                raise RuntimeError("aggregation stash: the recomputation does not repeat the recorded pass")
            replay = kept is not None
            if replay and not need_grad:
                return kept[0]                        # same inputs, deterministic kernels: the recorded output
        if replay:
            out, aux1, aux2, range_flag, z_save = kept
            ctx.range_flag = range_flag
            ctx.enc = (enc_feat, enc_w, enc_b) if enc else None
            ctx.egemm = egemm
            ctx.grad_sink = grad_sink
            ctx.save_for_backward(x, z_save if egemm else edge_attr, t_param, p_param, aux1, aux2, out)
            ctx.graph, ctx.mode, ctx.msg, ctx.eps = graph, mode, msg, eps
            ctx.t_val, ctx.p_val = t_val, p_val
            ctx.flags = ((_lib.FLAG_LEARN_T if learn_t else 0) | (_lib.FLAG_LEARN_P if learn_p else 0)
                         | (_lib.FLAG_ADD_ROOT if add_root else 0))
            ctx.learn_t, ctx.learn_p = learn_t, learn_p
            ctx.t_dtype, ctx.p_dtype = t_dtype, p_dtype
            ctx.add_root = add_root
            return out
        if record and stash.node_sized_only and egemm and mode != _lib.AGGR_MAX:
            stash.items.append((slot_key, None))
            record = False
        want_aux = need_grad or record
        out = torch.empty(graph.n_dst, C, device=dev, dtype=torch.float32)
        aux1 = aux2 = None
        if want_aux:
            if mode == _lib.AGGR_MAX:
                aux1 = torch.empty(graph.n_dst, C, device=dev, dtype=torch.int32)
            elif mode in (_lib.AGGR_SOFTMAX, _lib.AGGR_POWER):
                aux1 = torch.empty(graph.n_dst, C, device=dev, dtype=torch.float32)
            if (mode == _lib.AGGR_SOFTMAX and learn_t) or (mode == _lib.AGGR_POWER and learn_p):
                aux2 = torch.empty(graph.n_dst, C, device=dev, dtype=torch.float32)
        flags = (_lib.FLAG_LEARN_T if learn_t else 0) | (_lib.FLAG_LEARN_P if learn_p else 0)
        if add_root:
            if learn_t or learn_p or graph.n_dst != graph.n_src:
                raise ValueError("add_root needs a square graph and non-learnable t / p")
            flags |= _lib.FLAG_ADD_ROOT
        range_flag = None
        if want_aux and mode == _lib.AGGR_SOFTMAX and not learn_t and C % 4 == 0 and opts["SINGLE_GATHER_SOFTMAX_BWD"]:
            range_flag = torch.zeros(1, device=dev, dtype=torch.int32)   # set by the kernel if some |L| >= 80
        ws_bytes = lib.dgcn_gen_aggr_fwd_workspace_bytes(graph.c_struct, C)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8) if ws_bytes else None
        z_save = None
        if egemm:
            if want_aux and mode != _lib.AGGR_MAX:    # max: the arg-max ids carry all the backward needs
                z_save = torch.empty(graph.n_edges, C, device=dev, dtype=torch.float32)   # z_e, original edge order
            ws_bytes = lib.dgcn_gen_aggr_egemm_fwd_workspace_bytes(graph.n_edges, graph.n_src, n_feat, C)
            ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        with _lib.device_ctx(dev):
            if egemm:
                rc = lib.dgcn_gen_aggr_egemm_fwd_f32(
                    graph.c_struct, graph.erow.data_ptr(), x.data_ptr(), x.stride(0), enc_feat.data_ptr(),
                    enc_feat.stride(0), enc_w.data_ptr(), _lib.ptr(enc_b), n_feat, C, mode, msg, flags, t_val, p_val,
                    eps, _lib.ptr(t_param), _lib.ptr(p_param), out.data_ptr(), _lib.ptr(aux1), _lib.ptr(aux2),
                    _lib.ptr(range_flag), _lib.ptr(z_save), ws.data_ptr(), ws_bytes, _lib.current_stream_handle(dev))
            elif enc:
                rc = lib.dgcn_gen_aggr_enc_fwd_f32(
                    graph.c_struct, x.data_ptr(), x.stride(0), enc_feat.data_ptr(), enc_w.data_ptr(), _lib.ptr(enc_b),
                    ENC_FEATURES, C, mode, msg, flags | (_lib.FLAG_STATIC_ITEMS if opts["ENC_STATIC_ITEMS"] else 0), t_val, p_val, eps, _lib.ptr(t_param), _lib.ptr(p_param),
                    out.data_ptr(), _lib.ptr(aux1), _lib.ptr(aux2), _lib.ptr(range_flag), _lib.ptr(ws), ws_bytes,
                    _lib.current_stream_handle(dev))
            else:
                rc = lib.dgcn_gen_aggr_fwd_f32(
                    graph.c_struct, x.data_ptr(), x.stride(0), _lib.ptr(edge_attr), C, mode, msg, flags,
                    t_val, p_val, eps, _lib.ptr(t_param), _lib.ptr(p_param), out.data_ptr(),
                    _lib.ptr(aux1), _lib.ptr(aux2), _lib.ptr(range_flag), _lib.ptr(ws), ws_bytes,
                    _lib.current_stream_handle(dev))
        _lib.check(rc, "dgcn_gen_aggr_egemm_fwd_f32" if egemm else
                   ("dgcn_gen_aggr_enc_fwd_f32" if enc else "dgcn_gen_aggr_fwd_f32"))
        if record:
            stash.items.append((slot_key, (out, aux1, aux2, range_flag, z_save)))
        if need_grad:
            ctx.range_flag = range_flag
            ctx.enc = (enc_feat, enc_w, enc_b) if enc else None
            ctx.egemm = egemm
            ctx.grad_sink = grad_sink
            # fused edge GEMM: the backward reads the saved pre-activations z_e instead of x[src] + edge rows
            ctx.save_for_backward(x, z_save if egemm else edge_attr, t_param, p_param, aux1, aux2, out)
            ctx.graph, ctx.mode, ctx.msg, ctx.eps = graph, mode, msg, eps
            ctx.t_val, ctx.p_val, ctx.flags = t_val, p_val, flags
            ctx.learn_t, ctx.learn_p = learn_t, learn_p
            ctx.t_dtype, ctx.p_dtype = t_dtype, p_dtype
            ctx.add_root = add_root
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        opts = ctx.opts                                   # the forward's snapshot (see ``options``)
        x, edge_attr, t_param, p_param, aux1, aux2, out = ctx.saved_tensors
        graph, mode = ctx.graph, ctx.mode
        dev = x.device
        C = x.size(1)
        g = grad_out.float().contiguous()
        p = p_param if p_param is not None else ctx.p_val

        grad_t = grad_p = None
        if mode in (_lib.AGGR_MEAN, _lib.AGGR_POWER):
            # per-destination coefficient g * r^(1/p-1) * [q in range] / max(deg, 1) (MEAN: g / max(deg, 1)): one pass
            gcoef = torch.empty_like(g)
            with _lib.device_ctx(dev):
                _lib.check(lib.dgcn_power_bwd_prep_f32(
                    graph.c_struct, g.data_ptr(), aux1.data_ptr() if mode == _lib.AGGR_POWER else None,
                    _lib.ptr(p_param), ctx.p_val, gcoef.data_ptr(), C, _lib.current_stream_handle(dev)),
                    "dgcn_power_bwd_prep_f32")
            if mode == _lib.AGGR_POWER and ctx.learn_p and ctx.needs_input_grad[3]:
                # d o/d p = o * ( -ln r / p^2 + 1[q in range] * S2 / (p * deg * r) ),  S2 = sum u^p ln u
                q = aux1
                deg1 = graph.deg.clamp(min=1.0).unsqueeze(1)
                r = q.clamp(POW_LO, POW_HI)
                inr = ((q >= POW_LO) & (q <= POW_HI)).to(g.dtype)
                dodp = out * (-torch.log(r) / (p * p) + inr * aux2 / (p * deg1 * r))
                grad_p = (g * dodp).sum().reshape(p_param.shape).to(ctx.p_dtype)
        else:
            gcoef = g
        if mode == _lib.AGGR_SOFTMAX and ctx.learn_t and ctx.needs_input_grad[2]:
            # d L/d t = sum g * (sum_e w m^2 - out^2)      
            grad_t = (g * (aux2 - out * out)).sum().reshape(t_param.shape).to(ctx.t_dtype)

        grad_x = grad_ea = grad_w = grad_b = grad_feat = None
        egemm = ctx.egemm
        need_dz = egemm and any(ctx.needs_input_grad[14:17])
        # max over the fused edge GEMM: dz has one non-zero per (row, channel); the winners kernel needs no (E, C) array
        winners = (egemm and mode == _lib.AGGR_MAX and opts["EGEMM_MAX_WINNER_BWD"] and need_dz and C <= 128
                   and ctx.enc[0].size(1) <= 256)
        if winners:
            need_dz = False
        enc = None if egemm else ctx.enc                  # narrow per-edge encoder: dW | db partials, no (E, C) array
        # ... under max the forward's arg-max ids (-1 = relu floor) are all the backward needs: grad_x from the plain
        # walk (no encoder recomputation per edge), dW' | db' from the winners
        enc_winners = enc is not None and mode == _lib.AGGR_MAX and opts["ENC_MAX_WINNER_BWD"] and ctx.msg == _lib.MSG_RELU_EPS
        if enc_winners:
            enc_w_args, enc = enc, None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or need_dz or \
                (winners and ctx.enc[2] is not None and ctx.needs_input_grad[16]) or \
                (enc is not None and any(ctx.needs_input_grad[15:17])):
            gcoef = gcoef.contiguous()
            grad_x = torch.empty(graph.n_src, C, device=dev, dtype=torch.float32)
            if (edge_attr is not None or egemm) and (ctx.needs_input_grad[1] or need_dz):
                grad_ea = torch.empty(graph.n_edges, C, device=dev, dtype=torch.float32)
            ws_bytes = lib.dgcn_gen_aggr_bwd_workspace_bytes(graph.c_struct, C)
            ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8) if ws_bytes else None
            gshift = kshift = shift_ok = None
            bwd_flags = ctx.flags | (_lib.FLAG_EA_IS_Z if (egemm or enc_winners) else 0)
            if mode == _lib.AGGR_SOFTMAX and not ctx.learn_t and C % 4 == 0 and ctx.range_flag is not None:
                # g_i exp(t m - L_i) = [g_i exp(K_c - L_i)] exp(t m - K_c): one gathered row per edge.  K_c = 0
                # is safe whenever every |L_i| < 80, which the FORWARD kernel checked on the fly (range_flag);
                # the decision stays on the device (no host sync) and the kernel falls back to two gathers.
                kshift = _zeros_cached(dev, C)
                shift_ok = ctx.range_flag            # the kernel reads the forward's flag directly (0 = safe)
                bwd_flags |= _lib.FLAG_SHIFT_FLAG_IS_RANGE
                gshift = torch.empty_like(gcoef)
                with _lib.device_ctx(dev):
                    rc = lib.dgcn_softmax_bwd_prep_f32(gcoef.data_ptr(), aux1.data_ptr(), kshift.data_ptr(),
                                                       gshift.data_ptr(), gcoef.size(0), C,
                                                       _lib.current_stream_handle(dev))
                _lib.check(rc, "dgcn_softmax_bwd_prep_f32")
            with _lib.device_ctx(dev):
                if enc is not None:
                    feat, w_enc, b_enc = enc
                    nparts = lib.dgcn_gen_aggr_enc_bwd_num_partials(graph.c_struct, C)
                    gpart = torch.empty(nparts, C, ENC_FEATURES + 1, device=dev, dtype=torch.float32)
                    rc = lib.dgcn_gen_aggr_enc_bwd_f32(
                        graph.c_struct, x.data_ptr(), x.stride(0), feat.data_ptr(), w_enc.data_ptr(), _lib.ptr(b_enc),
                        ENC_FEATURES, C, mode, ctx.msg, bwd_flags | (_lib.FLAG_STATIC_ITEMS if opts["ENC_STATIC_ITEMS"] else 0),
                        ctx.t_val, ctx.p_val, ctx.eps, _lib.ptr(t_param),
                        _lib.ptr(p_param), gcoef.data_ptr(), _lib.ptr(aux1), _lib.ptr(out), _lib.ptr(gshift),
                        _lib.ptr(kshift), _lib.ptr(shift_ok), g.data_ptr() if ctx.add_root else None,
                        grad_x.data_ptr(), gpart.data_ptr(), _lib.ptr(ws), ws_bytes, _lib.current_stream_handle(dev))
                elif mode == _lib.AGGR_MAX and edge_attr is None and not egemm and not enc_winners and C <= 256 and \
                        graph.n_dst * C * 4 >= opts["MAX_MASK_MIN_TABLE_BYTES"] and graph.n_edges > 0:
                    # arg-max bit masks per edge instead of gathered arg-max rows (big graphs: the table misses the caches)
                    mbytes = lib.dgcn_gen_aggr_max_mask_bytes(graph.n_edges, C)
                    mask = torch.empty(mbytes, device=dev, dtype=torch.uint8)
                    rc = lib.dgcn_gen_aggr_max_bwd_f32(
                        graph.c_struct, graph.t_cpos.data_ptr(), x.data_ptr(), x.stride(0), C,
                        ctx.msg, bwd_flags, ctx.eps, gcoef.data_ptr(), aux1.data_ptr(),
                        g.data_ptr() if ctx.add_root else None, grad_x.data_ptr(), mask.data_ptr(), mbytes,
                        _lib.ptr(ws), ws_bytes, _lib.current_stream_handle(dev))
                else:
                    rc = lib.dgcn_gen_aggr_bwd_f32(
                        graph.c_struct, x.data_ptr(), x.stride(0), _lib.ptr(edge_attr), C, mode, ctx.msg,
                        bwd_flags, ctx.t_val, ctx.p_val, ctx.eps, _lib.ptr(t_param), _lib.ptr(p_param),
                        gcoef.data_ptr(), _lib.ptr(aux1), _lib.ptr(out), _lib.ptr(gshift), _lib.ptr(kshift),
                        _lib.ptr(shift_ok), g.data_ptr() if ctx.add_root else None, grad_x.data_ptr(),
                        _lib.ptr(grad_ea), _lib.ptr(ws), ws_bytes, _lib.current_stream_handle(dev))
            _lib.check(rc, "dgcn_gen_aggr_enc_bwd_f32" if enc is not None else "dgcn_gen_aggr_bwd_f32")
            if enc is not None:
                gw_sum, gb_sum = _lib.sum_partials_split(gpart)     # fixed-order partials -> dW (C, 8) and db (C,), one launch
                if ctx.needs_input_grad[15]:
                    grad_w = gw_sum
                if b_enc is not None and ctx.needs_input_grad[16]:
                    grad_b = gb_sum
            if winners:
                feat, w_enc, b_enc = ctx.enc
                if b_enc is not None and ctx.needs_input_grad[16]:
                    grad_b = grad_x.sum(0) - g.sum(0) if ctx.add_root else grad_x.sum(0)
            elif egemm:
                # dz = dL/dz_e (E, C), original edge order = the gradient of the never-materialised edge embedding
                feat, w_enc, b_enc = ctx.enc
                dz, grad_ea = grad_ea, None
                if dz is not None:
                    if ctx.needs_input_grad[14]:
                        sink = ctx.grad_sink
                        from . import node_ops
                        if sink is not None:
                            node_ops.rows_matmul_accumulate_(sink, dz, w_enc)   # running sum owned by the caller
                        else:
                            grad_feat = node_ops.rows_matmul(dz, w_enc)
                    if ctx.needs_input_grad[15]:
                        from . import node_ops
                        grad_w = node_ops.rows_tn(dz, feat)        # dz^T F on the matrix pipe (strided feature view in place)
                    if b_enc is not None and ctx.needs_input_grad[16]:
                        # sum_e dz_e = sum_s grad_x[s] (every edge lands in exactly one source row): an (N, C)
                        # reduction instead of an (E, C) one; the fused root term adds g to every row
                        grad_b = grad_x.sum(0) - g.sum(0) if ctx.add_root else grad_x.sum(0)
            if not ctx.needs_input_grad[0]:
                grad_x = None
        if enc_winners and any(ctx.needs_input_grad[15:17]):
            feat, w_enc, b_enc = enc_w_args
            gpart = torch.empty(lib.dgcn_enc_max_bwd_num_partials(graph.n_dst), C, ENC_FEATURES + 1, device=dev,
                                dtype=torch.float32)
            gc_rows = gcoef.contiguous()
            with _lib.device_ctx(dev):
                _lib.check(lib.dgcn_enc_max_bwd_weight_f32(gc_rows.data_ptr(), aux1.data_ptr(), graph.n_dst, graph.n_edges,
                                                           feat.data_ptr(),
                                                           ENC_FEATURES, C, gpart.data_ptr(),
                                                           _lib.current_stream_handle(dev)), "dgcn_enc_max_bwd_weight_f32")
            gw_sum, gb_sum = _lib.sum_partials_split(gpart)
            if ctx.needs_input_grad[15]:
                grad_w = gw_sum
            if b_enc is not None and ctx.needs_input_grad[16]:
                grad_b = gb_sum
        if winners and (ctx.needs_input_grad[14] or ctx.needs_input_grad[15]):
            feat, w_enc, b_enc = ctx.enc
            n_feat = feat.size(1)
            gf = None
            if ctx.needs_input_grad[14]:
                gf = ctx.grad_sink                         # running sum owned by the caller (edge_grad_sink), or a fresh one
                if gf is None:
                    gf = grad_feat = torch.zeros(graph.n_edges, n_feat, device=dev, dtype=torch.float32)
            wpart = None
            if ctx.needs_input_grad[15]:
                wpart = torch.empty(lib.dgcn_egemm_max_bwd_num_partials(graph.n_dst), C, n_feat, device=dev,
                                    dtype=torch.float32)
            gc_rows = gcoef.contiguous()
            with _lib.device_ctx(dev):
                _lib.check(lib.dgcn_egemm_max_bwd_f32(
                    gc_rows.data_ptr(), aux1.data_ptr(), graph.n_dst, graph.n_edges, feat.data_ptr(),
                    feat.stride(0), w_enc.data_ptr(), n_feat, C, _lib.ptr(gf), gf.stride(0) if gf is not None else 0,
                    _lib.ptr(wpart), _lib.current_stream_handle(dev)), "dgcn_egemm_max_bwd_f32")
            if wpart is not None:
                grad_w = _lib.sum_partials(wpart)
        return (grad_x, grad_ea, grad_t, grad_p) + (None,) * 10 + (grad_feat, grad_w, grad_b)


def gen_aggregate(x: torch.Tensor, edge_index: Union[torch.Tensor, Graph],
                  edge_attr: Optional[torch.Tensor] = None, aggr: str = "softmax",
                  t: Union[float, torch.Tensor] = 1.0, p: Union[float, torch.Tensor] = 1.0,
                  learn_t: bool = False, learn_p: bool = False, relu_eps: bool = True,
                  eps: float = 1e-7, dim_size: Optional[int] = None, add_root: bool = False,
                  edge_encoder=None) -> torch.Tensor:
    """out_i = AGGR_{e: dst(e)=i} m_e with m_e = relu(x[src(e)] (+edge_attr_e)) + eps.

    ``aggr`` in {add, mean, max, softmax, softmax_sg, softmax_sum, power, power_sum}; the
    ``*_sum`` degree scaling (torch_message.py:60-63,77-80) is applied by the caller.
    ``t`` / ``p`` may be python floats or 1-element device tensors (learnable parameters are
    read on the device, no host synchronisation).  ``relu_eps=False`` aggregates raw rows.
    ``add_root`` returns ``x + out`` (the ``h = x + m`` of GENConv.forward) from the same kernel: the root row is
    added in the epilogue and the upstream gradient in the backward's, saving two elementwise passes per layer.
    ``edge_encoder=(weight, bias)`` with ``edge_attr`` = the (E, hidden) features the layer's ``edge_encoder`` would be
    applied to fuses GENConv's ``Linear(edge_feat_dim -> C)`` into the aggregation (``encoder_fusable``): no (E, C)
    tensor at all.
    """
    if aggr not in _MODES:
        raise NotImplementedError("To be implemented")  # torch_message.py:85
    graph = graph_of(edge_index, x.size(0) if dim_size is None else dim_size)
    mode = _MODES[aggr]
    t_val, t_param = _scalar_arg(t)
    p_val, p_param = _scalar_arg(p)
    learn_t = bool(learn_t and t_param is not None and mode == _lib.AGGR_SOFTMAX)
    learn_p = bool(learn_p and p_param is not None and mode == _lib.AGGR_POWER)
    if t_param is not None and not learn_t:
        t_param = t_param.detach()
    if p_param is not None and not learn_p:
        p_param = p_param.detach()
    msg = _lib.MSG_RELU_EPS if relu_eps else _lib.MSG_IDENTITY
    if edge_encoder is not None:
        w_enc, b_enc = edge_encoder
        return _GenAggregate.apply(x, None, t_param, p_param, graph, mode, msg, float(eps), t_val, p_val, learn_t,
                                   learn_p, torch.is_grad_enabled(), bool(add_root), edge_attr, w_enc, b_enc)
    return _GenAggregate.apply(x, edge_attr, t_param, p_param, graph, mode, msg, float(eps),
                               t_val, p_val, learn_t, learn_p, torch.is_grad_enabled(), bool(add_root))


def softmax_state_forward(x: torch.Tensor, graph: Graph, t: float = 1.0, relu_eps: bool = True, eps: float = 1e-7):
    """``(out, L)`` of the softmax aggregation over ``graph`` (no autograd): ``L[i, c]`` = log sum_e exp(t m_e) over the
    row's edges (0 for a row without edges, whose ``out`` is 0).  Two partial aggregations over disjoint edge sets of the
    same destination rows merge exactly from these two arrays (``dist.SplitGraph``: the local-source edges are aggregated
    while the remote rows are still in flight)."""
    lib = _lib.load()
    dev = _lib.require_device(x)
    x = _rows_f32(x)
    C = x.size(1)
    if x.size(0) != graph.n_src:
        raise ValueError(f"x has {x.size(0)} rows, graph expects {graph.n_src}")
    out = torch.empty(graph.n_dst, C, device=dev, dtype=torch.float32)
    L = torch.empty(graph.n_dst, C, device=dev, dtype=torch.float32)
    ws_bytes = lib.dgcn_gen_aggr_fwd_workspace_bytes(graph.c_struct, C)
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8) if ws_bytes else None
    msg = _lib.MSG_RELU_EPS if relu_eps else _lib.MSG_IDENTITY
    with _lib.device_ctx(dev):
        rc = lib.dgcn_gen_aggr_fwd_f32(graph.c_struct, x.data_ptr(), x.stride(0), None, C, _lib.AGGR_SOFTMAX, msg, 0,
                                       float(t), 1.0, float(eps), None, None, out.data_ptr(), L.data_ptr(), None, None,
                                       _lib.ptr(ws), ws_bytes, _lib.current_stream_handle(dev))
    _lib.check(rc, "dgcn_gen_aggr_fwd_f32")
    return out, L


def softmax_state_merge(out_a: torch.Tensor, lse_a: torch.Tensor, graph_a: Graph, out_b: torch.Tensor, lse_b: torch.Tensor,
                        graph_b: Graph):
    """``(out, L)`` of the softmax aggregation over the UNION of two disjoint edge sets of the same destination rows from
    the two partial ``softmax_state_forward`` results, one launch (``dgcn_softmax_state_merge_f32``); written into
    ``out_a`` / ``lse_a``."""
    lib = _lib.load()
    dev = _lib.require_device(out_a, lse_a, out_b, lse_b)
    n, C = out_a.shape
    if graph_a.n_dst != n or graph_b.n_dst != n or out_b.shape != out_a.shape:
        raise ValueError("the two partial states must cover the same destination rows")
    if C % 4 != 0:                                   # (the aggregation kernels serve such widths; the merge does not)
        both = ((graph_a.deg > 0) & (graph_b.deg > 0)).unsqueeze(1)
        only_a = (graph_a.deg > 0).unsqueeze(1)
        w = torch.where(both, torch.sigmoid(lse_a - lse_b), only_a.to(out_a.dtype).expand_as(out_a))
        return w * out_a + (1.0 - w) * out_b, torch.where(both, torch.logaddexp(lse_a, lse_b), torch.where(only_a, lse_a, lse_b))
    with _lib.device_ctx(dev):
        rc = lib.dgcn_softmax_state_merge_f32(out_a.data_ptr(), lse_a.data_ptr(), graph_a.rowptr.data_ptr(), out_b.data_ptr(),
                                              lse_b.data_ptr(), graph_b.rowptr.data_ptr(), out_a.data_ptr(), lse_a.data_ptr(),
                                              n, C, _lib.current_stream_handle(dev))
    _lib.check(rc, "dgcn_softmax_state_merge_f32")
    return out_a, lse_a


def softmax_state_prepare(g: torch.Tensor, L: torch.Tensor):
    """The node-wise prologue of the single-gather softmax backward for a given log-sum-exp array: ``(g exp(-L), zeros,
    range flag)``; shared by the backward launches of a split aggregation."""
    lib = _lib.load()
    dev = _lib.require_device(g, L)
    g = g.float().contiguous()
    L = L.float().contiguous()
    C = g.size(1)
    if C % 4 != 0:
        return None
    kshift = _zeros_cached(dev, C)
    flag = (L.abs() >= SHIFT_SAFE_ABS_L).any().to(torch.int32).reshape(1)       # 1 = NOT safe (the forward's convention)
    gshift = torch.empty_like(g)
    with _lib.device_ctx(dev):
        _lib.check(lib.dgcn_softmax_bwd_prep_f32(g.data_ptr(), L.data_ptr(), kshift.data_ptr(), gshift.data_ptr(), g.size(0), C,
                                                 _lib.current_stream_handle(dev)), "dgcn_softmax_bwd_prep_f32")
    return gshift, kshift, flag


def softmax_state_backward(x: torch.Tensor, graph: Graph, g: torch.Tensor, L: torch.Tensor, t: float = 1.0,
                           relu_eps: bool = True, eps: float = 1e-7, prep=None) -> torch.Tensor:
    """grad_x (graph.n_src, C) of  sum_i g_i . sum_e w_e m_e  with the weights w_e = exp(t m_e - L_i) held constant
    (softmax_sg / softmax without learnable t, gcn_lib/sparse/torch_message.py:54-58) for a GIVEN log-sum-exp array --
    the merged one of a split aggregation.  ``prep = softmax_state_prepare(g, L)``: the single-gather form (one gathered
    row per edge; the kernel falls back to gathering g and L when some |L| >= 80)."""
    lib = _lib.load()
    dev = _lib.require_device(x, g, L)
    x = _rows_f32(x)
    C = x.size(1)
    g = g.float().contiguous()
    L = L.float().contiguous()
    grad_x = torch.empty(graph.n_src, C, device=dev, dtype=torch.float32)
    ws_bytes = lib.dgcn_gen_aggr_bwd_workspace_bytes(graph.c_struct, C)
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8) if ws_bytes else None
    msg = _lib.MSG_RELU_EPS if relu_eps else _lib.MSG_IDENTITY
    gshift, kshift, flag = prep if prep is not None else (None, None, None)
    flags = _lib.FLAG_SHIFT_FLAG_IS_RANGE if prep is not None else 0
    with _lib.device_ctx(dev):
        rc = lib.dgcn_gen_aggr_bwd_f32(graph.c_struct, x.data_ptr(), x.stride(0), None, C, _lib.AGGR_SOFTMAX, msg, flags,
                                       float(t), 1.0, float(eps), None, None, g.data_ptr(), L.data_ptr(), None,
                                       _lib.ptr(gshift), _lib.ptr(kshift), _lib.ptr(flag), None, grad_x.data_ptr(), None,
                                       _lib.ptr(ws), ws_bytes, _lib.current_stream_handle(dev))
    _lib.check(rc, "dgcn_gen_aggr_bwd_f32")
    return grad_x


def power_state_forward(x: torch.Tensor, graph: Graph, p: float = 1.0, relu_eps: bool = True, eps: float = 1e-7):
    """``(out, q)`` of the power-mean aggregation over ``graph`` (no autograd): ``q[i, c]`` = the PRE-CLAMP mean of m^p over
    the row's edges (what the forward kernel saves for its backward; 0 for a row without edges).  Two partial
    aggregations over disjoint edge sets of the same destination rows merge exactly from q and the two degrees
    (``dist.SplitGraph``); the partial OUTPUTS do not (the reference clamps the mean before the root,
    gcn_lib/sparse/torch_message.py:70-74)."""
    lib = _lib.load()
    dev = _lib.require_device(x)
    x = _rows_f32(x)
    C = x.size(1)
    if x.size(0) != graph.n_src:
        raise ValueError(f"x has {x.size(0)} rows, graph expects {graph.n_src}")
    out = torch.empty(graph.n_dst, C, device=dev, dtype=torch.float32)
    q = torch.empty(graph.n_dst, C, device=dev, dtype=torch.float32)
    ws_bytes = lib.dgcn_gen_aggr_fwd_workspace_bytes(graph.c_struct, C)
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8) if ws_bytes else None
    msg = _lib.MSG_RELU_EPS if relu_eps else _lib.MSG_IDENTITY
    with _lib.device_ctx(dev):
        rc = lib.dgcn_gen_aggr_fwd_f32(graph.c_struct, x.data_ptr(), x.stride(0), None, C, _lib.AGGR_POWER, msg, 0,
                                       1.0, float(p), float(eps), None, None, out.data_ptr(), q.data_ptr(), None, None,
                                       _lib.ptr(ws), ws_bytes, _lib.current_stream_handle(dev))
    _lib.check(rc, "dgcn_gen_aggr_fwd_f32")
    return out, q


def power_state_backward(x: torch.Tensor, graph: Graph, coef: torch.Tensor, q: torch.Tensor, p: float = 1.0,
                         relu_eps: bool = True, eps: float = 1e-7) -> torch.Tensor:
    """grad_x (graph.n_src, C) of the power-mean aggregation for a GIVEN per-destination coefficient
    ``coef = g r^(1/p - 1) 1[lo <= q <= hi] / max(deg, 1)`` (r = clamp(q)): the caller forms it from the MERGED mean and
    the total degree of a split aggregation; the edge walk is the one of ``_GenAggregate.backward``."""
    lib = _lib.load()
    dev = _lib.require_device(x, coef, q)
    x = _rows_f32(x)
    C = x.size(1)
    coef = coef.float().contiguous()
    q = q.float().contiguous()
    grad_x = torch.empty(graph.n_src, C, device=dev, dtype=torch.float32)
    ws_bytes = lib.dgcn_gen_aggr_bwd_workspace_bytes(graph.c_struct, C)
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8) if ws_bytes else None
    msg = _lib.MSG_RELU_EPS if relu_eps else _lib.MSG_IDENTITY
    with _lib.device_ctx(dev):
        rc = lib.dgcn_gen_aggr_bwd_f32(graph.c_struct, x.data_ptr(), x.stride(0), None, C, _lib.AGGR_POWER, msg, 0,
                                       1.0, float(p), float(eps), None, None, coef.data_ptr(), q.data_ptr(), None,
                                       None, None, None, None, grad_x.data_ptr(), None,
                                       _lib.ptr(ws), ws_bytes, _lib.current_stream_handle(dev))
    _lib.check(rc, "dgcn_gen_aggr_bwd_f32")
    return grad_x


def encoder_fusable(x: torch.Tensor, edge_feat: torch.Tensor, weight: Optional[torch.Tensor], narrow: bool = False) -> bool:
    """Whether ``Linear(edge_feat)`` can be folded into the aggregation kernels.

    * wide features, F % 16 == 0, F <= 256, C % 4 == 0, C <= 128 -- the ``Linear(hidden -> C)`` every reference model
      with edge features puts in its GENConv layers (ogbn-proteins / ogbg-ppa / RevGCN: ``edge_feat_dim =
      hidden_channels``): an E x F x C GEMM on the matrix cores inside the aggregation;
    * ``narrow=True`` (blocks.ComposedEdgeEmbedding: the model-level and the per-layer encoder composed into one
      Linear(8 -> C)): F == 8, C % 4 == 0, C <= 256, evaluated per edge in registers.  ``weight`` may be None here (the
      composed weight is formed by the caller)."""
    if edge_feat is None or edge_feat.dim() != 2 or x.dim() != 2 or not edge_feat.is_floating_point():
        return False
    C, F = x.size(-1), edge_feat.size(1)
    if torch.is_autocast_enabled() or edge_feat.size(0) == 0:
        return False
    if narrow:
        return (F == ENC_FEATURES and C % 4 == 0 and C <= 256 and not edge_feat.requires_grad
                and (weight is None or tuple(weight.shape) == (C, F)))
    if weight is None or tuple(weight.shape) != (C, F) or not current_options()["FUSED_EDGE_GEMM"]:
        return False
    return bool(_lib.load().dgcn_gen_aggr_egemm_supported(F, C))


def selftest(device="cuda:0") -> None:
    """Prove libdgcn launches on torch's stream and sees torch's allocations."""
    lib = _lib.load()
    dev = torch.device(device)
    x = torch.arange(1000, device=dev, dtype=torch.float32)
    y = torch.ones(1000, device=dev, dtype=torch.float32)
    with _lib.device_ctx(dev):
        rc = lib.dgcn_selftest_axpy_f32(2.0, x.data_ptr(), y.data_ptr(), x.numel(),
                                        _lib.current_stream_handle(dev))
    _lib.check(rc, "dgcn_selftest_axpy_f32")
    exp = 2.0 * torch.arange(1000, dtype=torch.float32) + 1.0
    if not torch.equal(y.cpu(), exp):
        raise RuntimeError("libdgcn selftest produced wrong values: HIP runtime mismatch?")
