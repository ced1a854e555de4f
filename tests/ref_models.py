"""Test helper: instantiate the REFERENCE's example architectures (files under /root/reference/examples,
loaded by path, never copied) on top of whichever `gcn_lib` is registered in sys.modules.
Only usable where the reference tree exists (the build container)."""
import argparse
import importlib.util
import os
import sys
import types

import torch

sys.dont_write_bytecode = True   # the reference tree stays free of __pycache__ (it is read-only input)

REF = "/root/reference"


def have_reference():
    return os.path.isdir(os.path.join(REF, "examples"))


def _load(path, name):
    d = os.path.dirname(path)
    sys.modules.pop("__init__", None)            # the examples do `import __init__` (their own path hack)
    sys.path.insert(0, d)
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(d)
        sys.modules.pop("__init__", None)
    return mod


def dense_deepgcn(n_blocks=28, **over):
    arch = _load(os.path.join(REF, "examples/sem_seg_dense/architecture.py"), "ref_sem_seg_dense_architecture")
    opt = argparse.Namespace(n_filters=64, k=16, act="relu", norm="batch", bias=True, epsilon=0.2, stochastic=True,
                             conv="edge", n_blocks=n_blocks, block="res", in_channels=9, dropout=0.3, n_classes=13)
    for k, v in over.items():
        setattr(opt, k, v)
    return arch.DenseDeepGCN(opt)


def arxiv_deepergcn(num_layers=28, **over):
    m = _load(os.path.join(REF, "examples/ogb/ogbn_arxiv/model.py"), "ref_ogbn_arxiv_model")
    args = argparse.Namespace(num_layers=num_layers, dropout=0.5, block="res+", in_channels=128, hidden_channels=128,
                              num_tasks=40, conv="gen", gcn_aggr="softmax_sg", t=0.1, learn_t=False, p=1.0,
                              learn_p=False, y=0.0, learn_y=False, msg_norm=False, learn_msg_scale=False,
                              norm="batch", mlp_layers=1)
    for k, v in over.items():
        setattr(args, k, v)
    return m.DeeperGCN(args)


def ppi_deepgcn(conv="mr", n_blocks=3, **over):
    arch = _load(os.path.join(REF, "examples/ppi/architecture.py"), "ref_ppi_architecture")
    opt = argparse.Namespace(n_filters=64, act="relu", norm="batch", bias=True, conv=conv, n_heads=1, n_blocks=n_blocks,
                             block="res", in_channels=50, dropout=0.2, n_classes=121)
    for k, v in over.items():
        setattr(opt, k, v)
    return arch.DeepGCN(opt)


def proteins_revgcn(tmpdir, num_layers=4, hidden=64, aggr="power", n_table=50, **over):
    # rev_layer.py wraps its imports in one try-block that starts with torch_geometric: give it a stub
    if "torch_geometric" not in sys.modules:
        tgnn = types.ModuleType("torch_geometric.nn")
        for n in ("GCNConv", "SAGEConv", "GATConv"):
            setattr(tgnn, n, type(n, (torch.nn.Module,), {}))
        tg = types.ModuleType("torch_geometric")
        tg.nn = tgnn
        sys.modules["torch_geometric"], sys.modules["torch_geometric.nn"] = tg, tgnn
    if REF not in sys.path:
        sys.path.append(REF)
    nf = os.path.join(tmpdir, "nf.pt")
    torch.save(torch.rand(n_table, 8), nf)
    m = _load(os.path.join(REF, "examples/ogb_eff/ogbn_proteins/model_rev.py"), "ref_proteins_model_rev")
    args = argparse.Namespace(num_layers=num_layers, dropout=0.2, group=2, hidden_channels=hidden, num_tasks=112,
                              gcn_aggr=aggr, t=1.0, learn_t=False, p=1.0, learn_p=True, y=0.0, learn_y=False,
                              msg_norm=False, learn_msg_scale=False, conv_encode_edge=True, norm="layer", mlp_layers=2,
                              nf_path=nf, use_one_hot_encoding=True, device="cpu")
    for k, v in over.items():
        setattr(args, k, v)
    return m.RevGCN(args)
