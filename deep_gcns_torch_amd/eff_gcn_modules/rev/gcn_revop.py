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
  ``dz @ W`` straight into the right column block, ``ops.edge_grad_sink``); only the layer whose backward runs last
  hands the buffer to autograd.
"""
import numpy as np
import torch
import torch.nn as nn

from ... import ops

__all__ = ["InvertibleCheckpointFunction", "InvertibleModuleWrapper", "get_device_states", "set_device_states"]

_USES = "_dgcn_rev_uses"        # attribute on a shared argument tensor: layers whose backward is still to come
_ACC = "_dgcn_rev_grad"         # attribute on a shared argument tensor: the running gradient sum


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
        with torch.no_grad():
            outputs = fn(*_detached(inputs))
        if not isinstance(outputs, tuple):
            outputs = (outputs,)
        outputs = tuple(o.detach_() for o in outputs)
        if not keep_input:
            _release(inputs[0])                    # only the node features are dropped (reference :62-67)
        # tensors every layer receives (edge embedding): count the layers whose backward will contribute
        module = getattr(fn, "__self__", None)
        ctx.fused = (hasattr(module, "fused_backward") and getattr(fn, "__name__", "") == "forward"
                     and not keep_input and not preserve_rng_state)
        if ctx.fused:
            for t in inputs[2:]:
                if _is_shared_arg(t):
                    setattr(t, _USES, getattr(t, _USES, 0) + 1)
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
        shared = [t for t in inputs[2:] if _is_shared_arg(t)]
        sinks = []
        for t in shared:                              # running gradient sums of the tensors all layers share
            acc = getattr(t, _ACC, None)
            if acc is None:
                acc = module.make_arg_sink(t)
                setattr(t, _ACC, acc)
            sinks.append(acc)
        x, grad_x, weight_grads = module.fused_backward(y, grad_outputs[0], inputs[1], inputs[2:], ctx.weights,
                                                        sinks, ops.edge_grad_sink)
        _release(y)
        _restore(inputs[0], x)
        arg_grads = []
        k = 0
        for t in inputs[2:]:
            if not _is_shared_arg(t):
                arg_grads.append(None)
                continue
            left = getattr(t, _USES, 1) - 1
            if left > 0:
                setattr(t, _USES, left)
                arg_grads.append(None)                # a layer further down the backward pass hands the sum over
            else:
                arg_grads.append(module.finish_arg_sink(sinks[k], t))
                for name in (_USES, _ACC):
                    if hasattr(t, name):
                        delattr(t, name)
            k += 1
        first = grad_x if ctx.input_requires_grad[0] else None
        return (first, None) + tuple(arg_grads) + tuple(weight_grads)


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
