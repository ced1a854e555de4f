"""BASELINE config 5 at the depth it names: RevGCN, 112 layers, hidden 224, group 2 (examples/ogb_eff/ogbn_proteins/
args.py:40-54) at the cluster shape (N = 13,253, E = 791,225), max and power aggregation, dropout 0.

Fixtures: tests/golden/config_revgcn112_{max,power}.pt, produced in the build container by the reference's REAL
model_rev.RevGCN on its REAL eff_gcn_modules/rev/* (tests/golden/make_revgcn112_golden.py), in float32 -- what the
reference computes -- and in float64 -- what it approximates.  At depth 112 two correct float32 evaluations of this model
do not agree to 1e-4 elementwise: the forward is continuous but long (224 coupling functions, each a LayerNorm -> GENConv
-> MLP chain), and the gradients are discontinuous at every relu of the messages / MLPs and at every arg-max of the
aggregation, so a pre-activation within rounding of zero moves a whole gradient term.  The yardstick is therefore the
reference's own float32 error against float64, recorded in the fixture per quantity: the device result has to be as
close to the float64 values as the reference's float32 run is, up to the factor written next to each gate.

Routes (all through the C ABI): the model file's class on this package's eff_gcn_modules.rev with the fused edge-GEMM
kernels (``product``: install(fuse_models=False)), the same class fused from outside (``fused``: the default install --
composed per-edge encoders), and the fused step captured and replayed as one hipGraph (``graphed``).
Also measured and bounded: the drift of the reversible reconstruction -- the input of layer 0 as the backward rebuilds it
through 112 inverse couplings against what the forward saw (fixture: the reference's own drift, both precisions)."""
import json
import os

import pytest
import torch

import config_replays as cr
import rev_restated

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("identity_dropout_mask")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HN_FACTOR = 4.0        # device error vs float64 <= max(this x the reference's float32 error vs float64, 1e-4 of the scale)
GRAD_FACTOR = 2.0      # every kept parameter gradient (max error / max |gradient|) <= this x the WORST such error of the
                       # reference's own float32 run over the kept parameters (which parameter a flipped relu / arg-max
                       # lands in differs between two float32 evaluations; the size of the worst hit does not -- SHOWN
                       # since round 6 on the reference itself: a second float32 run with the edge order permuted is off
                       # by up to 45x the first run's error on individual parameters, its worst hit by 2x:
                       # tests/golden/make_revgcn112_perm.py, tests/test_oracle_golden.py::test_two_float32_runs...)
                       # The STRICT statement for max aggregation is tests/test_revgcn112_attribution_gpu.py: along the
                       # device's own ReLU / arg-max decisions every one of the 2,248 parameter gradients is within 3.8e-6
                       # of its scale of the float64 evaluation -- what this factor absorbs is decisions, not arithmetic.
DRIFT_FACTOR = 8.0     # rebuilt layer-0 input: relative L2 error <= this x the reference's float32 drift


def _fixture(aggr):
    path = cr.revgcn_fixture_path(aggr)
    assert os.path.exists(path), f"{path} missing: python tests/golden/make_revgcn112_golden.py {aggr}"
    return torch.load(path, map_location="cpu", weights_only=False)


def _record(name, data):
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"revgcn112_{name}.json"), "w") as f:
            json.dump(data, f, indent=1)


@pytest.mark.parametrize("route", ["fused", "product", "graphed"])
@pytest.mark.parametrize("aggr", ["max", "power"])
def test_revgcn112_full_depth_against_the_reference(aggr, route):
    # (dropout 0: conftest.identity_dropout_mask pins the identity mask the model draws with bernoulli_(1.0))
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from deep_gcns_torch_amd import fuse
    from deep_gcns_torch_amd.graphs import GraphedStep
    fix = _fixture(aggr)
    dev = torch.device("cuda:0")
    inp = cr.revgcn_inputs()
    c = fix["ctor"]
    m = rev_restated.RevGCNModelFile(num_layers=c["num_layers"], hidden=c["hidden"], aggr=aggr, dropout=0.0,
                                     learn_p=c["learn_p"], p=c["p"], node_table=inp["table"].to(dev), impl="product")
    assert list(m.state_dict().keys()) == fix["param_keys"]
    cr.formula_init(m, seed=5)
    got = cr.checksums(inp["x"], inp["edge_index"], m.state_dict())
    for k, v in fix["checksums"].items():                       # same seeded inputs and parameters as the generator's
        assert abs(got[k] - v) <= 1e-9 * max(1.0, abs(v)), (k, got[k], v)
    m = m.to(dev).train()
    if route != "product":
        fuse.fuse_model(m)
    x, nidx, ei, ea = (inp[k].to(dev) for k in ("x", "node_index", "edge_index", "edge_attr"))
    probe = inp["probe"].to(dev)
    keep = {}
    m.node_features_encoder.register_forward_hook(lambda mod, i, o: keep.update(h0_obj=o, h0_true=o.detach().clone()))
    m.last_norm.register_forward_hook(lambda mod, i, o: keep.__setitem__("hn", o))
    hn_static = torch.empty(inp["n"], c["hidden"], device=dev)
    h0_rebuilt = torch.empty(inp["n"], c["hidden"], device=dev)
    h0_true = torch.empty(inp["n"], c["hidden"], device=dev)

    def step():
        for p in m.parameters():
            p.grad = None
        pred = m(x, nidx, ei, ea)
        assert tuple(pred.shape) == tuple(fix["pred_shape"])
        hn = keep["hn"]
        (hn * probe).sum().backward()
        with torch.no_grad():
            hn_static.copy_(hn)
            h0_rebuilt.copy_(keep["h0_obj"])
            h0_true.copy_(keep["h0_true"])

    if route == "graphed":
        g = GraphedStep(step, warmup=1)
        g()
        g()
    else:
        step()
    torch.cuda.synchronize()
    hn = hn_static.cpu().double()
    rows = fix["rows"]
    rec = dict(aggr=aggr, route=route)

    # ---- forward: sampled rows, column sums and norm of ALL rows against the float64 reference ----
    ref64 = fix["hn_rows64"].double()
    ref_err = float((fix["hn_rows32"].double() - ref64).abs().max())
    dev_err = float((hn[rows] - ref64).abs().max())
    scale = float(ref64.abs().max())
    rec.update(hn_max_err_device=dev_err, hn_max_err_reference_float32=ref_err, hn_scale=scale)
    col_ref_err = float((fix["hn_colsum32"] - fix["hn_colsum64"]).abs().max())
    col_dev_err = float((hn.sum(0) - fix["hn_colsum64"]).abs().max())
    rec.update(colsum_err_device=col_dev_err, colsum_err_reference_float32=col_ref_err,
               norm_device=float(hn.norm()), norm64=fix["hn_norm64"])

    # ---- backward: gradients of the first / last layers, the encoders and last_norm ----
    gerr = {}
    for k, g64 in fix["grads64"].items():
        p = dict(m.named_parameters())[k]
        assert p.grad is not None, k
        gd = p.grad.detach().cpu().double()
        den = float(g64.abs().max()) + 1e-300
        gerr[k] = (float((gd - g64.double()).abs().max()) / den, fix["grad_err32_vs_64"][k])
    worst = max(gerr.items(), key=lambda kv: kv[1][0] / max(kv[1][1], 1e-4))
    rec.update(worst_grad=dict(name=worst[0], device=worst[1][0], reference_float32=worst[1][1]),
               grad_err_device_max=max(v[0] for v in gerr.values()),
               grad_err_reference_float32_max=max(v[1] for v in gerr.values()))
    # a SECOND float32 run of the reference (edge order permuted: tests/golden/make_revgcn112_perm.py) -- recorded next to
    # the first: per parameter the two correct float32 evaluations differ by up to 45x (power), their worst hits by 2x
    perm_path = cr.revgcn_fixture_path(aggr).replace(".pt", "_perm.pt")
    if os.path.exists(perm_path):
        perm = torch.load(perm_path, map_location="cpu", weights_only=False)["grad_err_perm_vs_64"]
        rec["worst_grad"]["reference_float32_permuted_edges"] = perm[worst[0]]
        rec["grad_err_reference_float32_permuted_edges_max"] = max(perm.values())
        rec["parameters_where_the_two_reference_runs_differ_10x"] = sum(
            1 for k in perm if max(perm[k], 1e-12) / max(fix["grad_err32_vs_64"][k], 1e-12) > 10
            or max(perm[k], 1e-12) / max(fix["grad_err32_vs_64"][k], 1e-12) < 0.1)

    # ---- reversible reconstruction drift over 112 inverse couplings ----
    drift = float((h0_rebuilt - h0_true).double().norm() / h0_true.double().norm())
    rec.update(drift_rel_l2_device=drift, drift_rel_l2_reference_float32=fix["drift32"]["rel_l2"],
               drift_rel_l2_reference_float64=fix["drift64"]["rel_l2"])
    _record(f"{aggr}_{route}", rec)

    assert dev_err <= max(HN_FACTOR * ref_err, 1e-4 * scale), rec        # (1e-4 relative: BASELINE.json's fp32 tolerance)
    # column sums over all 13,253 rows: every element may be off by the forward's error in the same direction
    assert col_dev_err <= max(HN_FACTOR * col_ref_err, 1e-6 * inp["n"] * scale), rec
    assert abs(float(hn.norm()) - fix["hn_norm64"]) <= 1e-4 * fix["hn_norm64"], rec
    ref_worst = max(max(v[1] for v in gerr.values()), 1e-4)
    for k, (e_dev, e_ref) in gerr.items():
        assert e_dev <= GRAD_FACTOR * ref_worst, (k, e_dev, e_ref, ref_worst)
    assert drift <= DRIFT_FACTOR * max(fix["drift32"]["rel_l2"], 1e-7), rec
