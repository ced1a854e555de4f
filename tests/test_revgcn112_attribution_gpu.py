"""BASELINE config 5 at depth 112, MAX aggregation: every parameter gradient of the device step against float64 ALONG THE
DEVICE'S OWN DECISIONS.

tests/test_revgcn112_gpu.py holds the device to the reference's own float32-vs-float64 error (two correct float32 runs of
this model differ per parameter by whole gradient terms: a ReLU within rounding of zero, two arg-max candidates within
rounding of each other).  Here those decisions are taken out of the comparison instead: a recording pass of the device
model stores the on / off mask of all 449 ReLU sites (13,253 x 112 / 224 each) and the arg-max edge id of all 224
aggregation launches (13,253 x 112 each), and the float64 host evaluation (tests/attribution.revgcn_max_backward_along)
is forced through the same branches -- what is left between the two gradients is fp32 rounding through 224 coupling
functions, and it has to be SMALL for every parameter, not just as small as the reference's own float32 noise.

Measured (round 6): all 2,248 parameter gradients of the 112-layer model within 3.8e-6 of their scale (median 1.2e-6),
last_norm's output within 9.2e-7 of its max, along 0.5 G decisions.  One subtlety decides the result: the gradient is
formed by the BACKWARD's evaluation of every coupling function on the REBUILT input (x_i = y_i - F_i, off the forward's
by 4e-7 relative at this depth), and 10 of the 0.33 G ReLU pre-activations take the other branch there than in the forward
pass -- replaying the FORWARD's masks leaves 2.7e-3 on `gcns.69._fn.Fms.0.norm.bias`; the reference's own reversible
scheme (eff_gcn_modules/rev/gcn_revop.py:98-140) recomputes on rebuilt inputs in the same way.
Host memory: ~34 GB for the float64 evaluation at 112 layers (every coupling function checkpointed).
``power`` (8 layers): on the default route the ReLU sites are forced and the aggregation is the oracle's in float64 with its
own per-edge message ReLU (2.1e-4 -- see the comment at its gate); on the edge-GEMM route the kernels keep the (E, C)
pre-activations for their backward, their signs are forced too (1.5 G decisions at 8 layers) and the strict tolerance
holds: 1.0e-5 at worst (the learnable exponent ``p``), median 7.4e-7.  DGCN_LONG_TESTS=1 adds that route at 112 layers
(15 minutes of float64 edge-level work on a 256-core host; profiles/r06_revgcn112_power_attribution.log: 20.9 G decisions,
worst gradient 4.8e-5 of its scale on the model-level `edge_encoder.bias`, median 1.3e-6 -- the number that
tests/test_revgcn112_gpu.py holds to 4.2e-3 against the reference's OWN decisions).
"""
import os

import pytest
import torch

import attribution
import config_replays as cr
import rev_restated

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("identity_dropout_mask")]

GRAD_TOL = 2e-5          # max |device - float64 along the device's decisions| / scale, EVERY parameter (BASELINE's fp32 tolerance is
                         # 1e-4; measured: 2.2e-6 at 8 layers, 3.8e-6 at 112 -- profiles/r06_test_gates.json)


_CASES = [("max", 8, "fused"), ("max", 112, "fused"), ("max", 112, "product"), ("power", 8, "fused"), ("power", 8, "product")]
if os.environ.get("DGCN_LONG_TESTS") == "1":      # ~1 h of float64 edge-level work on the host, ~50 GB of host memory
    _CASES.append(("power", 112, "product"))


@pytest.mark.parametrize("aggr,layers,route", _CASES)
def test_revgcn_gradients_along_the_device_decisions(aggr, layers, route):
    """route: ``fused`` = the default install (composed per-edge encoders); ``product`` = the model file's own forward on this
    package's eff_gcn_modules.rev (the (E, 448) edge embedding exists, every layer's encoder is the fused edge GEMM)."""
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from conftest import gate
    from deep_gcns_torch_amd import fuse, ops
    from deep_gcns_torch_amd.eff_gcn_modules.rev import gcn_revop
    dev = torch.device("cuda:0")
    inp = cr.revgcn_inputs()
    kw = dict(num_layers=layers, hidden=224, aggr=aggr, dropout=0.0, node_table=inp["table"], learn_p=aggr == "power", p=1.0)
    host = rev_restated.RevGCNModelFile(impl="restated", **kw)
    cr.formula_init(host, seed=5)
    m = rev_restated.RevGCNModelFile(impl="product", **kw)
    m.load_state_dict(host.state_dict())
    m.node_features = inp["table"].to(dev)
    m = m.to(dev).train()
    if route == "fused":
        m = fuse.fuse_model(m)
    x, nidx, ei, ea = (inp[k].to(dev) for k in ("x", "node_index", "edge_index", "edge_attr"))
    probe = inp["probe"].to(dev)

    ids = None
    if aggr == "max":
        # ---- the arg-max ids: a pass without graph and without kept aggregations (the wrapper's own stashes would take the
        #      launches); the step below keeps and replays exactly these (deterministic kernels, same inputs) ----
        stash = ops.AggregationStash(node_sized_only=True)
        saved = gcn_revop.KEEP_AGGREGATION
        gcn_revop.KEEP_AGGREGATION = False
        try:
            with torch.no_grad(), ops.stash_aggregation(stash, "record"):
                m(x, nidx, ei, ea)
        finally:
            gcn_revop.KEEP_AGGREGATION = saved
        ids = [kept[1].cpu() for _, kept in stash.items]
        assert len(ids) == 2 * layers and all(t.dtype == torch.int32 and t.shape == (inp["n"], 112) for t in ids)
        stash.items.clear()

    # ---- the step as shipped, every ReLU site recorded.  The gradient is formed by the BACKWARD's grad-enabled evaluation
    #      of every coupling function, whose input is the REBUILT one (x_i = y_i - F_i: off the forward's by the
    #      reconstruction's rounding, 4e-7 relative at this depth) -- a pre-activation within that distance of zero takes
    #      the other branch there, and it is that branch the gradient follows (the LayerNorm backward recomputes its ReLU
    #      mask from the input it saved).  Sites in call order: the forward's 4 x layers + 1, then the backward's, last layer
    #      first, last group first, within a block norm -> ReLU then the MLP's ----
    # power on the edge-GEMM route: the kernels keep the (E, C) pre-activations z_e for their backward -- their signs are the
    # message ReLU's decisions in the kernels' own arithmetic.  Taken from the launches of the BACKWARD's grad-enabled
    # evaluations (the forward's run without a graph and keep nothing), bit-packed on the device
    edge_bits = []
    orig_fwd = ops._GenAggregate.forward
    if aggr == "power" and route == "product":
        def fwd(ctx, *a, **k):
            out = orig_fwd(ctx, *a, **k)
            saved = getattr(ctx, "to_save", None)
            if saved is not None and getattr(ctx, "egemm", False) and saved[1] is not None:
                z = saved[1]
                bits = (z > 0).view(-1)
                pad = (-bits.numel()) % 8
                if pad:
                    bits = torch.cat([bits, bits.new_zeros(pad)])
                w = (bits.view(-1, 8).to(torch.uint8) << torch.arange(8, device=z.device, dtype=torch.uint8)).sum(1, dtype=torch.uint8)
                edge_bits.append((w.cpu(), tuple(z.shape)))
            return out
        ops._GenAggregate.forward = staticmethod(fwd)

    dec = attribution.ReluDecisions()
    keep = {}
    hook = m.last_norm.register_forward_hook(lambda mod, i, o: keep.__setitem__("hn", o))
    try:
        with dec.recording():
            m(x, nidx, ei, ea)
            hook.remove()
            (keep["hn"] * probe).sum().backward()
            torch.cuda.synchronize()
    finally:
        ops._GenAggregate.forward = orig_fwd
    group = 2
    assert len(dec.masks) == 2 * (2 * group * layers) + 1, len(dec.masks)
    fwd_masks, bwd_masks = dec.masks[:2 * group * layers], dec.masks[2 * group * layers + 1:]
    masks = [None] * (2 * group * layers) + [dec.masks[2 * group * layers]]
    v = 0
    for L in range(layers - 1, -1, -1):
        for g in range(group - 1, -1, -1):
            k = L * group + g
            masks[2 * k], masks[2 * k + 1] = bwd_masks[v], bwd_masks[v + 1]
            v += 2
    flipped = sum(int((a != b).sum()) for a, b in zip(fwd_masks, masks[:-1]))
    edge_masks = None
    if edge_bits:
        assert len(edge_bits) == group * layers, len(edge_bits)
        edge_masks = [None] * (group * layers)
        v = 0
        for L in range(layers - 1, -1, -1):                 # the backward's launches: last layer first, last group first
            for g in range(group - 1, -1, -1):
                w, shape = edge_bits[v]

                def unpack(w=w, shape=shape):
                    bits = ((w.unsqueeze(1) >> torch.arange(8, dtype=torch.uint8)) & 1).bool().view(-1)
                    return bits[:shape[0] * shape[1]].view(shape)
                edge_masks[L * group + g] = unpack
                v += 1

    # ---- float64 along the same branches ----
    host = host.double().train()
    hn64 = attribution.revgcn_max_backward_along(host, masks, ids, inp, inp["probe"], aggr=aggr, edge_masks=edge_masks)
    hn_err = float((keep["hn"].detach().cpu().double() - hn64).abs().max() / hn64.abs().max())
    errs = attribution.gradient_errors(m, host)
    worst = max(errs.items(), key=lambda kv: kv[1])
    n_dec = sum(int(t.numel()) for t in masks) + sum(int(t.numel()) for t in (ids or [])) + sum(
        sh[0] * sh[1] for _, sh in edge_bits)
    print(f"[revgcn{layers} {aggr} {route}] {n_dec} decisions replayed ({flipped} ReLU sites where the backward's evaluation on the "
          f"rebuilt input took the other branch than the forward); last_norm output {hn_err:.2e} of its max; worst parameter gradient "
          f"{worst[1]:.2e} of its scale ({worst[0]}); median {sorted(errs.values())[len(errs) // 2]:.2e}")
    gate(f"revgcn{layers} {aggr}, {route} route: last_norm output vs float64 along the device's decisions (max error / max)", hn_err, 1e-4)
    # power: the aggregation's per-edge message ReLU (E x C = 177 M decisions per launch) is NOT forced -- the host's float64
    # pass decides for itself, and ~300 of the 2.8 G pre-activations of 8 layers lie within fp32 rounding of zero.  One such
    # flip removes one term from a sum of 1.6 M signed terms (the encoder bias' gradient): ~1e-3 of the sum, not a rounding
    # error and not a kernel bug; measured 2.1e-4.  Under max every decision is forced and the strict tolerance applies.
    # power on the edge-GEMM route: the per-edge decisions ARE forced (the kernels' own z_e) and the strict tolerance applies
    tol = GRAD_TOL if (aggr == "max" or edge_masks is not None) else 1e-3
    if aggr == "power" and layers > 8:
        # 224 launches x 177 M edge terms: the model-level edge encoder's bias collects a signed sum over every edge of every
        # layer; measured 4.8e-5 of its scale (median over the 2,472 parameters 1.3e-6) -- BASELINE's fp32 tolerance
        tol = 1e-4
    gate(f"revgcn{layers} {aggr}, {route} route: worst parameter gradient vs float64 along the device's decisions (max error / scale), "
         f"all {len(errs)} parameters", worst[1], tol, what=worst[0])
