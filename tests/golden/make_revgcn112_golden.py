"""TEST INFRASTRUCTURE.  Generates tests/golden/config_revgcn112_{max,power}.pt: BASELINE config 5 at the depth it names
(RevGCN, 112 layers, hidden 224, group 2, conv_encode_edge, LayerNorm, mlp_layers 2; examples/ogb_eff/ogbn_proteins/
args.py:40-54) at the cluster shape (N = 13,253, E = 791,225), dropout 0, executed by the reference's REAL
examples/ogb_eff/ogbn_proteins/model_rev.py:85-112 on its REAL eff_gcn_modules/rev/{gcn_revop,memgcn,rev_layer}.py and
gcn_lib.sparse (third-party scatter primitives from oracle/thirdparty.py via oracle/refshim.py), once in float32 (what
the reference computes) and once in float64 (what it approximates).  Build container only:

    python tests/golden/make_revgcn112_golden.py max power            (about an hour of host time per aggregator)
    python tests/golden/make_revgcn112_golden.py --layers 8 --scale 0.05 max     (a quick variant for the CPU suite)

Recorded per aggregator: checksums of the seeded inputs and formula-initialised parameters, last_norm's output on 1,024
sampled rows + float64 column sums and norm of all rows (both precisions), the gradients of L = sum(last_norm_out * probe)
w.r.t. the first and last layers' parameters, the node / edge encoders and last_norm (both precisions), and the drift of
the reversible reconstruction: the input of layer 0 as the backward rebuilds it (inverse of 112 couplings) against the
value the forward saw."""
import argparse
import io
import os
import sys
import tempfile
import time
from contextlib import redirect_stdout

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)
sys.dont_write_bytecode = True


def kept_param(name, layers):
    return (name.startswith("gcns.0.") or name.startswith(f"gcns.{layers - 1}.") or not name.startswith("gcns."))


def run(aggr, layers, scale, dtype, inp):
    from oracle import refshim
    refshim.import_reference()            # the reference's gcn_lib / utils + restated third-party primitives
    import config_replays as cr
    import ref_models
    with redirect_stdout(io.StringIO()), tempfile.TemporaryDirectory() as tmp:
        m = ref_models.proteins_revgcn(tmp, num_layers=layers, hidden=224, aggr=aggr, n_table=inp["n"], dropout=0.0,
                                       learn_p=(aggr == "power"), p=1.0)
    assert type(m).__module__ == "ref_proteins_model_rev"
    import eff_gcn_modules.rev.gcn_revop as ref_revop
    assert ref_revop.__file__.startswith("/root/reference")
    cr.formula_init(m, seed=5)
    m.node_features = inp["table"].clone()
    if dtype == torch.float64:
        m = m.double()
        m.node_features = m.node_features.double()
    m.train()
    cast = lambda t: t.to(dtype) if t.is_floating_point() else t
    keep = {}
    h1 = m.node_features_encoder.register_forward_hook(
        lambda mod, i, o: keep.update(h0_obj=o, h0_true=o.detach().clone()))
    h2 = m.last_norm.register_forward_hook(lambda mod, i, o: keep.__setitem__("hn", o))
    t0 = time.time()
    pred = m(cast(inp["x"]), inp["node_index"], inp["edge_index"], cast(inp["edge_attr"]))
    t_fwd = time.time() - t0
    h1.remove(); h2.remove()
    hn = keep["hn"]
    (hn * cast(inp["probe"])).sum().backward()
    t_all = time.time() - t0
    h0_rebuilt, h0_true = keep["h0_obj"].detach(), keep["h0_true"]
    drift = dict(rel_l2=float((h0_rebuilt - h0_true).double().norm() / h0_true.double().norm()),
                 max_abs=float((h0_rebuilt - h0_true).abs().max()), h0_abs_max=float(h0_true.abs().max()))
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None and kept_param(k, layers)}
    gnorm = {k: float(p.grad.double().norm()) for k, p in m.named_parameters() if p.grad is not None}
    sd = {k: v.detach().float() for k, v in m.state_dict().items()}
    return dict(hn=hn.detach(), grads=grads, grad_norms=gnorm, drift=drift, seconds=(t_fwd, t_all),
                pred_shape=tuple(pred.shape), state_dict=sd, learn_p=m.learn_p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("aggrs", nargs="*", default=["max", "power"])
    ap.add_argument("--layers", type=int, default=112)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    import config_replays as cr
    inp = cr.revgcn_inputs(args.scale)
    rows = cr.sample_rows(inp["n"], cr.REVGCN_OUT_ROWS, 303)
    for aggr in args.aggrs:
        r32 = run(aggr, args.layers, args.scale, torch.float32, inp)
        print(aggr, "float32 done", r32["seconds"], r32["drift"], flush=True)
        r64 = run(aggr, args.layers, args.scale, torch.float64, inp)
        print(aggr, "float64 done", r64["seconds"], r64["drift"], flush=True)
        hn32, hn64 = r32["hn"], r64["hn"]
        fix = dict(aggr=aggr, layers=args.layers, scale=args.scale, n=inp["n"], n_edges=int(inp["edge_index"].size(1)),
                   ctor=dict(num_layers=args.layers, hidden=224, aggr=aggr, dropout=0.0, learn_p=(aggr == "power"), p=1.0),
                   checksums=cr.checksums(inp["x"], inp["edge_index"], r32["state_dict"]),
                   param_keys=list(r32["state_dict"].keys()),
                   rows=rows, hn_rows32=hn32[rows].clone(), hn_rows64=hn64[rows].clone(),
                   hn_colsum32=hn32.double().sum(0), hn_colsum64=hn64.sum(0),
                   hn_norm32=float(hn32.double().norm()), hn_norm64=float(hn64.norm()),
                   hn_err32_vs_64=dict(max_abs=float((hn32.double() - hn64).abs().max()),
                                       rel_l2=float((hn32.double() - hn64).norm() / hn64.norm())),
                   grads32=r32["grads"], grads64={k: v.float() for k, v in r64["grads"].items()},
                   grad_err32_vs_64={k: float((r32["grads"][k].double() - r64["grads"][k]).abs().max()
                                              / (r64["grads"][k].abs().max() + 1e-300)) for k in r64["grads"]},
                   grad_norms32=r32["grad_norms"], grad_norms64=r64["grad_norms"],
                   drift32=r32["drift"], drift64=r64["drift"], pred_shape=r32["pred_shape"],
                   seconds32=r32["seconds"], seconds64=r64["seconds"], torch_version=torch.__version__,
                   threads=torch.get_num_threads())
        path = cr.revgcn_fixture_path(aggr, args.layers, args.scale)
        torch.save(fix, path)
        worst = max(fix["grad_err32_vs_64"].items(), key=lambda kv: kv[1])
        print(aggr, "->", path, f"{os.path.getsize(path) / 1e6:.1f} MB; hn float32 vs float64:", fix["hn_err32_vs_64"],
              "; worst kept gradient (max error / max):", worst, flush=True)


if __name__ == "__main__":
    main()
