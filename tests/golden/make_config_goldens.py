"""TEST INFRASTRUCTURE.  Generates tests/golden/config_deepergcn28_*.pt: the reference's own model file at the BASELINE
configuration sizes (see tests/config_replays.py).  Run in the build container, from the repository root:

    python tests/golden/make_config_goldens.py [quarter_powerlaw full_arxiv]

Minutes of host time per fixture (28 layers of the reference's scatter_softmax chain on the host cores)."""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)


def deepergcn(size):
    """The reference's REAL examples/ogb/ogbn_arxiv/model.py:DeeperGCN (28 layers, 'res+', softmax_sg t = 0.1, BatchNorm,
    mlp_layers 1, dropout 0) on the REFERENCE gcn_lib.sparse, third-party scatter primitives from oracle/thirdparty.py
    (oracle/refshim.py) -- rounds 3 - 4 replayed the restated class (tests/arch_restated.py) with the oracle's aggregation
    instead.  Parameters: config_replays.formula_init(seed 33), which the test repeats on its own
    classes."""
    from oracle import refshim
    refshim.import_reference()
    import config_replays as cr
    import ref_models
    n, ei, x = cr.deepergcn_inputs(size)
    kw = cr.DEEPERGCN_KW
    mc = ref_models.arxiv_deepergcn(kw["num_layers"], dropout=0.0, in_channels=kw["in_channels"],
                                    hidden_channels=kw["hidden"], num_tasks=kw["num_tasks"], gcn_aggr=kw["aggr"],
                                    t=kw["t"], norm=kw["norm"], mlp_layers=kw["mlp_layers"])
    assert type(mc).__module__ == "ref_ogbn_arxiv_model"
    import gcn_lib.sparse.torch_vertex as tv
    assert tv.__file__.startswith("/root/reference")
    cr.formula_init(mc, seed=33)
    sd = {k: v.clone() for k, v in mc.state_dict().items()}
    rows = cr.sample_rows(n, cr.N_OUT_ROWS, 101)
    hrows = cr.sample_rows(n, cr.N_HID_ROWS, 202)
    hidden, hooks = [], []
    for nm in mc.norms:
        hooks.append(nm.register_forward_pre_hook(lambda mod, inp: hidden.append(inp[0].detach()[hrows].clone())))
    mc.train()
    t0 = time.time()
    with torch.no_grad():
        ref = mc(x, ei)
    for h in hooks:
        h.remove()
    fix = dict(size=size, n=n, n_edges=int(ei.size(1)), kw=kw, checksums=cr.checksums(x, ei, sd), param_keys=list(sd.keys()),
               rows=rows, out_rows=ref[rows].clone(), out_colsum64=ref.double().sum(0), out_norm64=float(ref.double().norm()),
               hidden_rows=hrows, hidden=torch.stack(hidden), torch_version=torch.__version__,
               source="reference examples/ogb/ogbn_arxiv/model.py on the reference gcn_lib (oracle/refshim.py)",
               oracle_seconds=time.time() - t0, threads=torch.get_num_threads())
    torch.save(fix, cr.fixture_path(size))
    print(size, "->", cr.fixture_path(size), f"{fix['oracle_seconds']:.0f} s", flush=True)


if __name__ == "__main__":
    for s in (sys.argv[1:] or ["quarter_powerlaw", "full_arxiv"]):
        deepergcn(s)
