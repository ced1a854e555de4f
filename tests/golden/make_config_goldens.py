"""TEST INFRASTRUCTURE.  Generates tests/golden/config_*.pt: CPU-oracle replays of whole models at the BASELINE
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
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    import arch_restated
    import config_replays as cr
    n, ei, x = cr.deepergcn_inputs(size)
    torch.manual_seed(33)
    mc = arch_restated.DeeperGCN(**cr.DEEPERGCN_KW)
    mc.checkpoint_grad = False
    sd = {k: v.clone() for k, v in mc.state_dict().items()}
    rows = cr.sample_rows(n, cr.N_OUT_ROWS, 101)
    hrows = cr.sample_rows(n, cr.N_HID_ROWS, 202)
    t0 = time.time()
    ref, hidden = cr.deepergcn_oracle_forward(mc, x, ei, hrows)
    fix = dict(size=size, n=n, n_edges=int(ei.size(1)), kw=cr.DEEPERGCN_KW, checksums=cr.checksums(x, ei, sd),
               rows=rows, out_rows=ref[rows].clone(), out_colsum64=ref.double().sum(0), out_norm64=float(ref.double().norm()),
               hidden_rows=hrows, hidden=torch.stack(hidden), torch_version=torch.__version__,
               oracle_seconds=time.time() - t0, threads=torch.get_num_threads())
    torch.save(fix, cr.fixture_path(size))
    print(size, "->", cr.fixture_path(size), f"{fix['oracle_seconds']:.0f} s", flush=True)


if __name__ == "__main__":
    for s in (sys.argv[1:] or ["quarter_powerlaw", "full_arxiv"]):
        deepergcn(s)
