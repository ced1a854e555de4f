"""Reversible (memory-free) module wrapper: public surface of the reference's eff_gcn_modules/rev/gcn_revop.py
(InvertibleCheckpointFunction :17-157, InvertibleModuleWrapper :160-268, get/set_device_states :271-293).

Semantics kept: the forward runs under ``no_grad`` on detached inputs and the storage of the FIRST input (the node
features) is released; the backward re-creates it from the output with ``fn_inverse``, re-runs ``fn`` with grad
enabled and differentiates that.  Same ``keep_input`` / ``num_bwd_passes`` / ``preserve_rng_state`` / ``disable``
switches, same gradient values.

What is different ("reversible-aware fusion"):

* **shared inverse + recompute.**  For an additive coupling (``memgcn.GroupAdditiveCoupling``) the inverse and the
  grad-enabled recompute evaluate every ``Fm_i`` on IDENTICAL inputs (inverse: ``x_i = y_i - Fm_i(y_{i-1})``,
  recompute: ``y_i = x_i + Fm_i(y_{i-1})``).  A wrapped module that offers ``fused_backward`` gets ONE grad-enabled
  evaluation of each ``Fm_i`` per backward step: the input is reconstructed from it and the gradients are taken
  through the same graph.  RevGCN then runs the message-passing kernels twice per layer and step instead of three
  times; values are identical because the kernels are deterministic.
* **shared-argument gradients accumulate in place.**  Every layer of RevGCN receives the same ``edge_emb`` tensor
  (examples/ogb_eff/ogbn_proteins/model_rev.py:98-107); the reference returns an (E, hidden*group) gradient per
  layer and lets autograd add them up (three passes over 1.4 GB per layer at the ogbn-proteins cluster shape).
  Here the layers of one backward pass share one accumulation buffer per such tensor (GENConv adds its
  ``dz @ W`` straight into the right column block, ``ops.edge_grad_sink``).  No forward-side bookkeeping: the first
  layer whose backward runs in a backward pass (identified by autograd's graph-task id) creates the buffer and returns
  a memory-free placeholder gradient, every later layer adds in place and returns nothing, and a tensor hook on the
  shared argument -- which autograd runs once ALL its users have run -- swaps the placeholder for the finished sum
  (or adds the sum to whatever other consumers contributed).  Forward passes under ``no_grad``, forwards that are
  never followed by a backward, aborted backward passes and leaf (Parameter) arguments therefore cannot leave stale
  state behind.
* The fused path is taken only for wrapped modules without training-mode BatchNorm: the reference evaluates every
  ``Fm_i`` three times per step (forward, inverse, recompute), this path twice, so BatchNorm running statistics would
  receive two momentum updates instead of three.  Such modules take the generic path (= the reference's algorithm).
"""
import threading
import weakref

import numpy as np
import torch
import torch.nn as nn

from ... import ops

__all__ = ["InvertibleCheckpointFunction", "InvertibleModuleWrapper", "get_device_states", "set_device_states"]

_STATE = "_dgcn_rev_state"      # attribute on a shared argument tensor: its _SharedArgState
# The fused backward re-evaluates every coupling function on the input the forward gave it.  True: the aggregation
# launches of that forward keep their node-sized results for it (ops.AggregationStash; per GENBlock of the ogbn-proteins
# RevGCN: the (N, C) output + arg-max ids, 12 MB against 0.3 ms of edge kernel); False: launch again (memory as in the
# reference's reversible scheme, to the byte); "edge": also the aggregations whose backward needs an (E, C) array of
# the forward (softmax / power with the fused edge encoder: the pre-activations, 354 MB per GENBlock at the
# ogbn-proteins cluster shape -- 79 GB for RevGCN-112, which a 288 GB device holds; the reversible scheme's memory
# argument is then gone and only its arithmetic remains).
#
# "auto" (default): as True while the arrays kept by all live reversible layers of the device stay under
# KEEP_AGGREGATION_BUDGET_FRACTION of its memory (KEEP_AGGREGATION_BUDGET_BYTES overrides), as False beyond: a stack
# keeps 2 - 3 (N, C/group) arrays per coupling function, i.e. memory that GROWS WITH DEPTH (which the reference's
# reversible scheme exists to avoid) -- bounded here by the budget, so that the 112- / 1001-layer configurations keep
# their depth-independent activation memory once the budget is spent.
KEEP_AGGREGATION = "auto"
KEEP_AGGREGATION_BUDGET_FRACTION = 0.125
KEEP_AGGREGATION_BUDGET_BYTES = None

_KEPT_BYTES = {}                # device index -> bytes held by live stashes
_KEPT_LOCK = threading.Lock()


class _KeptStashes(list):
    """The stashes of one reversible layer (a list that can carry a finalizer: the ledger entry goes when it goes)."""
    __slots__ = ("__weakref__",)


def _ledger_add(dev_index, n):
    with _KEPT_LOCK:
        _KEPT_BYTES[dev_index] = _KEPT_BYTES.get(dev_index, 0) + n


def kept_aggregation_bytes(device=None) -> int:
    """Bytes of aggregation results currently kept for the backward by the reversible layers of ``device``."""
    idx = torch.device(device).index if device is not None else None
    if idx is None:
        idx = torch.cuda.current_device() if torch.cuda.is_available() else -1
    return _KEPT_BYTES.get(idx, 0)


def _budget_bytes(dev) -> int:
    if KEEP_AGGREGATION_BUDGET_BYTES is not None:
        return int(KEEP_AGGREGATION_BUDGET_BYTES)
    if dev.type != "cuda":
        return 0
    return int(torch.cuda.get_device_properties(dev).total_memory * KEEP_AGGREGATION_BUDGET_FRACTION)


def _keep_mode(dev):
    """KEEP_AGGREGATION resolved for one forward: False, True or "edge"."""
    if KEEP_AGGREGATION != "auto":
        return KEEP_AGGREGATION
    idx = dev.index if dev.type == "cuda" else -1
    return _KEPT_BYTES.get(idx, 0) < _budget_bytes(dev)


def _stash_bytes(stashes) -> int:
    n = 0
    for st in stashes:
        for _, kept in st.items:
            if kept is not None:
                n += sum(t.numel() * t.element_size() for t in kept if isinstance(t, torch.Tensor))
    return n


class _SharedArgState:
    """Running gradient sum of a tensor every reversible layer receives, valid for ONE backward pass (graph task)."""
    __slots__ = ("task", "acc", "finish", "ph_ptr")

    def __init__(self):
        self.task, self.acc, self.finish, self.ph_ptr = -1, None, None, 0


class _ShapeOf:
    """shape / dtype of a tensor without the tensor (what ``finish_arg_sink`` reads)."""
    __slots__ = ("shape", "dtype")

    def __init__(self, t):
        self.shape, self.dtype = t.shape, t.dtype


def _graph_task_id() -> int:
    return torch._C._current_graph_task_id()


def _deliver_hook(st: _SharedArgState):
    def hook(grad):
        if st.acc is None or st.task != _graph_task_id():
            return None                               # nothing accumulated in this backward pass
        total = st.finish(st.acc)
        st.acc, st.task, st.finish = None, -1, None   # (finish refers to the module and, weakly, to the tensor)
        if grad.data_ptr() == st.ph_ptr and all(s == 0 for s in grad.stride()):
            return total                              # only the placeholder arrived: hand over the sum itself
        return grad + total                           # other consumers of the tensor contributed as well
    return hook


def _ensure_state(t) -> _SharedArgState:
    st = getattr(t, _STATE, None)
    if st is None:
        st = _SharedArgState()
        setattr(t, _STATE, st)
        t.register_hook(_deliver_hook(st))            # runs when autograd has the TOTAL gradient of t
    return st


def get_device_states(*args):
    devices = sorted({a.get_device() for a in args if isinstance(a, torch.Tensor) and a.is_cuda})
    states = []
    for d in devices:
        with torch.cuda.device(d):
            states.append(torch.cuda.get_rng_state())
    return devices, states


def set_device_states(devices, states):
    for d, s in zip(devices, states):
        with torch.cuda.device(d):
            torch.cuda.set_rng_state(s)


def _detached(seq):
    return [t.detach() if isinstance(t, torch.Tensor) else t for t in seq]


def _release(t):
    t.untyped_storage().resize_(0)


def _restore(t, value):
    t.untyped_storage().resize_(int(np.prod(t.size())) * t.element_size())
    t.set_(value)


def _is_shared_arg(t):
    return isinstance(t, torch.Tensor) and t.requires_grad and t.is_floating_point()


class InvertibleCheckpointFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fn, fn_inverse, keep_input, num_bwd_passes, preserve_rng_state, num_inputs, *inputs_and_weights):
        ctx.fn, ctx.fn_inverse = fn, fn_inverse
        ctx.keep_input = keep_input
        ctx.weights = inputs_and_weights[num_inputs:]
        ctx.num_bwd_passes = num_bwd_passes
        ctx.preserve_rng_state = preserve_rng_state
        ctx.num_inputs = num_inputs
        inputs = inputs_and_weights[:num_inputs]
        if preserve_rng_state:
            ctx.fwd_cpu_state = torch.get_rng_state()
            ctx.had_cuda_in_fwd = torch.cuda._initialized
            if ctx.had_cuda_in_fwd:
                ctx.fwd_gpu_devices, ctx.fwd_gpu_states = get_device_states(*inputs)
        ctx.input_requires_grad = [isinstance(t, torch.Tensor) and t.requires_grad for t in inputs]
        ctx.options = ops.captured_options()       # the caller's ``ops.options`` block: the backward re-evaluates inside it
        # tensors every layer receives (edge embedding): count the layers whose backward will contribute
        module = getattr(fn, "__self__", None)
        ctx.fused = (hasattr(module, "fused_backward") and getattr(fn, "__name__", "") == "forward"
                     and not keep_input and not preserve_rng_state
                     and all(hasattr(t, _STATE) and module.arg_sink_ok(t) for t in inputs[2:] if _is_shared_arg(t))
                     and not any(isinstance(m, nn.modules.batchnorm._BatchNorm) and m.training
                                 for m in module.modules()))
        # the backward's grad-enabled evaluation of every F_i repeats this pass's: keep the aggregations' (N, C) results
        # (max / add / mean and the unfused softmax / power forms: 2 - 3 node-sized arrays per coupling function) instead
        # of launching the edge kernels again -- KEEP_AGGREGATION = False restores the pure recomputation
        keep = (_keep_mode(inputs[0].device) if ctx.fused and hasattr(module, "new_stashes") and any(ctx.needs_input_grad)
                else False)
        ctx.stashes = _KeptStashes(module.new_stashes(node_sized_only=keep != "edge")) if keep else None
        with torch.no_grad():
            if ctx.stashes is not None:
                outputs = fn(*_detached(inputs), _stashes=ctx.stashes)
                held = _stash_bytes(ctx.stashes)
                idx = inputs[0].device.index if inputs[0].device.type == "cuda" else -1
                _ledger_add(idx, held)
                weakref.finalize(ctx.stashes, _ledger_add, idx, -held)      # released with the layer's context
            else:
                outputs = fn(*_detached(inputs))
        if not isinstance(outputs, tuple):
            outputs = (outputs,)
        outputs = tuple(o.detach_() for o in outputs)
        if not keep_input:
            _release(inputs[0])                    # only the node features are dropped (reference :62-67)
        ctx.inputs = [inputs] * num_bwd_passes
        ctx.outputs = [outputs] * num_bwd_passes
        return outputs

    @staticmethod
    def backward(ctx, *grad_outputs):
        if not torch.autograd._is_checkpoint_valid():
            raise RuntimeError("InvertibleCheckpointFunction is not compatible with .grad(), please use .backward() "
                               "if possible")
        if len(ctx.outputs) == 0:
            raise RuntimeError("Trying to perform backward on the InvertibleCheckpointFunction for more than {} "
                               "times! Try raising `num_bwd_passes` by one.".format(ctx.num_bwd_passes))
        inputs = ctx.inputs.pop()
        outputs = ctx.outputs.pop()
        with ctx.options:
            return InvertibleCheckpointFunction._backward(ctx, inputs, outputs, grad_outputs)

    @staticmethod
    def _backward(ctx, inputs, outputs, grad_outputs):
        if ctx.fused:
            return (None,) * 6 + InvertibleCheckpointFunction._backward_fused(ctx, inputs, outputs, grad_outputs)

        # ---- generic path (any invertible module): inverse, then a grad-enabled recompute ----
        if not ctx.keep_input:
            devices = ctx.fwd_gpu_devices if (ctx.preserve_rng_state and ctx.had_cuda_in_fwd) else []
            with torch.random.fork_rng(devices=devices, enabled=ctx.preserve_rng_state):
                if ctx.preserve_rng_state:
                    torch.set_rng_state(ctx.fwd_cpu_state)
                    if ctx.had_cuda_in_fwd:
                        set_device_states(ctx.fwd_gpu_devices, ctx.fwd_gpu_states)
                with torch.no_grad():
                    rebuilt = ctx.fn_inverse(*(outputs + inputs[1:]))
                    for o in outputs:
                        _release(o)
                    if not isinstance(rebuilt, tuple):
                        rebuilt = (rebuilt,)
                    for original, value in zip(inputs, rebuilt):
                        _restore(original, value)
        with torch.enable_grad():
            leaves = []
            for t, req in zip(inputs, ctx.input_requires_grad):
                if isinstance(t, torch.Tensor):
                    t = t.detach()
                    t.requires_grad = req
                leaves.append(t)
            recomputed = ctx.fn(*leaves)
        if not isinstance(recomputed, tuple):
            recomputed = (recomputed,)
        wrt = tuple(t for t in leaves if isinstance(t, torch.Tensor) and t.requires_grad)
        grads = torch.autograd.grad(outputs=recomputed, inputs=wrt + ctx.weights, grad_outputs=grad_outputs)
        it = iter(grads[:len(wrt)])
        input_grads = tuple(next(it) if req else None for req in ctx.input_requires_grad)
        return (None,) * 6 + input_grads + tuple(grads[len(wrt):])

    @staticmethod
    def _backward_fused(ctx, inputs, outputs, grad_outputs):
        """One grad-enabled evaluation per coupling function: rebuilds the input AND yields the gradients."""
        module = ctx.fn.__self__
        y = outputs[0]
        tid = _graph_task_id()
        sinks, first = [], []
        for t in inputs[2:]:
            if not _is_shared_arg(t):
                continue
            st = getattr(t, _STATE)
            fresh = st.acc is None or st.task != tid          # first layer of THIS backward pass (stale sums are dropped)
            if fresh:
                st.acc, st.task = module.make_arg_sink(t), tid
                # only the tensor's geometry goes into the closure: the state hangs on the tensor itself (setattr), a
                # reference back would close a cycle tensor -> state -> closure -> tensor and the (E, hidden * group)
                # embedding of every step would wait for the cyclic collector (measured: +1.4 GB per step until it ran,
                # 22 -> 53 ms per step); a weak reference does not survive the hand-over of the tensor's Python object to
                # its C++ owner
                st.finish = (lambda buf, like=_ShapeOf(t), m=module: m.finish_arg_sink(buf, like))
            sinks.append(st.acc)
            first.append(fresh)
        x, grad_x, weight_grads = module.fused_backward(y, grad_outputs[0], inputs[1], inputs[2:], ctx.weights,
                                                        sinks, ops.edge_grad_sink, stashes=ctx.stashes)
        if len(ctx.outputs) == 0:
            ctx.stashes = None                     # last backward pass over this layer: the kept arrays go
        _release(y)
        _restore(inputs[0], x)
        arg_grads = []
        k = 0
        for t in inputs[2:]:
            if not _is_shared_arg(t):
                arg_grads.append(None)
                continue
            if first[k]:
                # a stride-0 zero "gradient": autograd then waits for every user of t before it runs t's hook, which
                # replaces it by the finished sum (the layers that follow add into the buffer in place)
                st = getattr(t, _STATE)
                ph = torch.zeros((), device=t.device, dtype=t.dtype).expand(t.shape)
                st.ph_ptr = ph.data_ptr()
                arg_grads.append(ph)
            else:
                arg_grads.append(None)
            k += 1
        first_in = grad_x if ctx.input_requires_grad[0] else None
        return (first_in, None) + tuple(arg_grads) + tuple(weight_grads)


class InvertibleModuleWrapper(nn.Module):
    """``y = fn(x, ...)`` without keeping ``x`` (it is rebuilt by ``fn.inverse`` in the backward pass).

    fn: module with ``forward`` and ``inverse`` (``x == fn.inverse(fn.forward(x))``).  keep_input / keep_input_inverse
    keep the input of forward / inverse alive; num_bwd_passes: how many backward passes may use the stored output;
    disable: plain ``fn(x)``; preserve_rng_state: replay the forward's RNG state during the reconstruction."""

    def __init__(self, fn, keep_input=False, keep_input_inverse=False, num_bwd_passes=1, disable=False,
                 preserve_rng_state=False):
        super().__init__()
        self.disable = disable
        self.keep_input = keep_input
        self.keep_input_inverse = keep_input_inverse
        self.num_bwd_passes = num_bwd_passes
        self.preserve_rng_state = preserve_rng_state
        self._fn = fn

    def _apply_reversible(self, f, f_inv, keep, args):
        weights = tuple(p for p in self._fn.parameters() if p.requires_grad)
        # an argument that stands for a function of parameters (blocks.ComposedEdgeEmbedding: the model-level edge
        # encoder) brings them along: their gradients leave through the same route as the block's own weights
        for a in args:
            if not isinstance(a, torch.Tensor) and hasattr(a, "parameters") and hasattr(a, "group_view"):
                known = {id(w) for w in weights}
                weights = weights + tuple(p for p in a.parameters() if id(p) not in known)
        if torch.is_grad_enabled() and hasattr(self._fn, "fused_backward"):
            for t in args[2:]:                        # tensors every layer receives: one delivery hook per tensor
                if _is_shared_arg(t):
                    _ensure_state(t)
        out = InvertibleCheckpointFunction.apply(f, f_inv, keep, self.num_bwd_passes, self.preserve_rng_state,
                                                 len(args), *(args + weights))
        return out[0] if isinstance(out, tuple) and len(out) == 1 else out

    def forward(self, *xin):
        if self.disable:
            y = self._fn(*xin)
            return y[0] if isinstance(y, tuple) and len(y) == 1 else y
        return self._apply_reversible(self._fn.forward, self._fn.inverse, self.keep_input, xin)

    def inverse(self, *yin):
        if self.disable:
            x = self._fn.inverse(*yin)
            return x[0] if isinstance(x, tuple) and len(x) == 1 else x
        return self._apply_reversible(self._fn.inverse, self._fn.forward, self.keep_input_inverse, yin)
