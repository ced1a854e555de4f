"""TEST INFRASTRUCTURE.  A SECOND float32 run of the reference's real RevGCN-112 (the model, inputs and parameters of
tests/golden/make_revgcn112_golden.py) with the EDGE ORDER PERMUTED: the same function, evaluated with the scatter sums in
another order -- another correct float32 evaluation.  Writes tests/golden/config_revgcn112_{aggr}_perm.pt: per kept
parameter, max error / max |gradient| of this run against the main fixture's float64 gradients and against its float32
gradients, and the same for last_norm's output.  What it shows: which parameter a flipped relu / arg-max decision lands
in differs between two float32 evaluations of the REFERENCE; the size of the worst hit does not (VERDICT r5 weak #1c).

    python tests/golden/make_revgcn112_perm.py max power            (build container; ~12 min per aggregator)
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
sys.dont_write_bytecode = True


def main():
    import config_replays as cr
    import make_revgcn112_golden as mk
    aggrs = sys.argv[1:] or ["max", "power"]
    inp = cr.revgcn_inputs(1.0)
    perm = torch.randperm(inp["edge_index"].size(1), generator=torch.Generator().manual_seed(99))
    pin = dict(inp, edge_index=inp["edge_index"][:, perm].contiguous(), edge_attr=inp["edge_attr"][perm].contiguous())
    rows = cr.sample_rows(inp["n"], cr.REVGCN_OUT_ROWS, 303)
    for aggr in aggrs:
        fix = torch.load(cr.revgcn_fixture_path(aggr), map_location="cpu", weights_only=False)
        r = mk.run(aggr, 112, 1.0, torch.float32, pin)
        print(aggr, "permuted float32 run done", r["seconds"], flush=True)
        relmax = lambda a, b: float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-300))
        out = dict(aggr=aggr, perm_seed=99,
                   grad_err_perm_vs_64={k: relmax(r["grads"][k], fix["grads64"][k]) for k in fix["grads64"]},
                   grad_err_perm_vs_32={k: relmax(r["grads"][k], fix["grads32"][k]) for k in fix["grads32"]},
                   hn_max_abs_perm_vs_64=float((r["hn"][rows].double() - fix["hn_rows64"].double()).abs().max()),
                   hn_max_abs_perm_vs_32=float((r["hn"][rows].double() - fix["hn_rows32"].double()).abs().max()),
                   drift=r["drift"], seconds=r["seconds"], threads=torch.get_num_threads())
        path = cr.revgcn_fixture_path(aggr).replace(".pt", "_perm.pt")
        torch.save(out, path)
        w64 = max(out["grad_err_perm_vs_64"].items(), key=lambda kv: kv[1])
        w32 = max(out["grad_err_perm_vs_32"].items(), key=lambda kv: kv[1])
        print(aggr, "->", path, "worst vs float64:", w64, "; worst vs the first float32 run:", w32, "; hn:",
              out["hn_max_abs_perm_vs_64"], out["hn_max_abs_perm_vs_32"], flush=True)


if __name__ == "__main__":
    main()
