"""CPU drop-in checks: the reference's OWN example architectures (loaded by path from /root/reference,
build container only) are instantiated on top of this package's gcn_lib/utils and must expose exactly
the reference's state_dict keys, shapes and parameter counts (checkpoint compatibility).
Forward passes are exercised on the GPU (tests/test_models_gpu.py); here nothing is computed."""
import io
import sys
from contextlib import redirect_stdout

import pytest
import torch

pytestmark = pytest.mark.usefixtures("identity_dropout_mask")    # dropout-0 models: see conftest.py

import ref_models
from conftest import load_golden

KEYS = load_golden("model_keys.pt")

needs_ref = pytest.mark.skipif(not ref_models.have_reference(), reason="/root/reference not present")


@pytest.fixture(scope="module")
def dropin():
    import deep_gcns_torch_amd
    for name in [n for n in sys.modules if n == "gcn_lib" or n.startswith("gcn_lib.") or n == "utils" or n.startswith("utils.")]:
        del sys.modules[name]
    lib = deep_gcns_torch_amd.install(reference_root=ref_models.REF if ref_models.have_reference() else None)
    assert "deep_gcns_torch_amd" in sys.modules["gcn_lib"].__name__
    yield lib


def _check(model, name):
    want = KEYS[name]
    got = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert list(got.keys()) == list(want["keys"].keys())
    assert got == want["keys"]
    assert sum(p.numel() for p in model.parameters()) == want["n_params"]


@needs_ref
def test_sem_seg_dense_resgcn28(dropin):
    with redirect_stdout(io.StringIO()):
        m = ref_models.dense_deepgcn(28)
    _check(m, "sem_seg_dense_resgcn28")
    assert type(m.head.gconv).__module__.startswith("deep_gcns_torch_amd")
    assert m.backbone[26].body.k == 16 and m.backbone[26].body.d == 27       # attributes callers read


@needs_ref
def test_ogbn_arxiv_deepergcn(dropin):
    with redirect_stdout(io.StringIO()):
        m = ref_models.arxiv_deepergcn(28)
        m2 = ref_models.arxiv_deepergcn(3, gcn_aggr="softmax", learn_t=True, msg_norm=True, learn_msg_scale=True,
                                        mlp_layers=2, norm="layer")
    _check(m, "ogbn_arxiv_deepergcn28")
    _check(m2, "ogbn_proteins_style_learn")
    g = m2.gcns[0]
    assert isinstance(g.t, torch.nn.Parameter) and g.t.shape == (1,) and g.learn_t is True
    assert m.gcns[0].t == 0.1 and m.gcns[0].learn_t is False
    assert g.msg_norm.msg_scale.shape == (1,)
    with redirect_stdout(io.StringIO()):
        m2.print_params(final=True)          # reads gcn.t / gcn.msg_norm.msg_scale (model.py:142-178)


@needs_ref
@pytest.mark.parametrize("conv", ["mr", "edge"])
def test_ppi_deepgcn(dropin, conv):
    _check(ref_models.ppi_deepgcn(conv), f"ppi_deepgcn_{conv}")


@needs_ref
def test_proteins_revgcn(dropin, tmp_path):
    with redirect_stdout(io.StringIO()):
        m = ref_models.proteins_revgcn(str(tmp_path))
    _check(m, "proteins_revgcn")


def test_install_registers_every_reference_module_path(dropin):
    from gcn_lib.dense import (BasicConv, DenseDilatedKnnGraph, DenseDynBlock2d, GraphConv2d, PlainDynBlock2d,  # noqa: F401
                               ResDynBlock2d)
    from gcn_lib.sparse import DenseGraphBlock, GraphConv, MLP, MultiSeq, ResGraphBlock  # noqa: F401
    from gcn_lib.sparse.torch_nn import norm_layer  # noqa: F401
    from gcn_lib.sparse.torch_vertex import GENConv  # noqa: F401
    from gcn_lib.sparse.torch_message import GenMessagePassing, MsgNorm  # noqa: F401
    from utils.pyg_util import scatter_  # noqa: F401
    from utils.data_util import get_atom_feature_dims, get_bond_feature_dims
    assert get_atom_feature_dims() == [119, 4, 12, 12, 10, 6, 6, 2, 2] and get_bond_feature_dims() == [5, 6, 2]


def test_unsupported_options_fail_loudly(dropin):
    from gcn_lib.dense import DynConv2d, EdgeConv2d
    from gcn_lib.sparse import GraphConv
    assert EdgeConv2d(8, 8, "relu", "instance")._per_edge and EdgeConv2d(8, 8, "prelu", "batch")._per_edge
    assert not EdgeConv2d(8, 8, "leakyrelu", "batch")._per_edge
    assert DynConv2d(8, 8, knn="tree").dilated_knn_graph.__class__.__name__ == "DilatedKnnGraph"
    with pytest.raises(NotImplementedError):
        GraphConv(8, 8, "gat")
    with pytest.raises(NotImplementedError):
        GraphConv(8, 8, "bogus")


@needs_ref
def test_install_fuse_models_patches_the_real_model_files_on_import(dropin, tmp_path):
    """What an example script does -- ``sys.path[0]`` is its own directory, ``from model import DeeperGCN`` /
    ``from model_rev import RevGCN`` -- after ``install(fuse_models=True)``: the classes of the UNCHANGED files get the
    fused forward (and keep their own as the fallback), the state_dict is the file's, and an instance that does not
    qualify (CPU tensors here) runs the file's own loop."""
    import os
    import types
    import deep_gcns_torch_amd
    from deep_gcns_torch_amd import fuse
    deep_gcns_torch_amd.install(reference_root=ref_models.REF)          # fuse_models defaults to True (round 5)
    assert fuse._FINDER in sys.meta_path
    if "torch_geometric" not in sys.modules:               # rev_layer.py's import block starts with torch_geometric
        tgnn = types.ModuleType("torch_geometric.nn")
        for n in ("GCNConv", "SAGEConv", "GATConv"):
            setattr(tgnn, n, type(n, (torch.nn.Module,), {}))
        tg = types.ModuleType("torch_geometric")
        tg.nn = tgnn
        sys.modules["torch_geometric"], sys.modules["torch_geometric.nn"] = tg, tgnn
    try:
        for sub, modname, clsname, repl in (("examples/ogb/ogbn_arxiv", "model", "DeeperGCN", fuse._deepergcn_forward),
                                            ("examples/ogb/ogbn_products", "model", "DeeperGCN", fuse._deepergcn_forward),
                                            ("examples/ogb_eff/ogbn_proteins", "model_rev", "RevGCN", fuse._revgcn_forward)):
            d = os.path.join(ref_models.REF, sub)
            for name in (modname, "__init__"):
                sys.modules.pop(name, None)
            sys.path.insert(0, d)
            try:
                with redirect_stdout(io.StringIO()):
                    mod = __import__(modname)
            finally:
                sys.path.remove(d)
            cls = getattr(mod, clsname)
            assert mod.__file__.startswith(d)
            assert cls.forward is repl and callable(cls.__dict__[fuse._ORIG])
            # a subclass without a forward of its own inherits the replacement; fusing it again must not record the
            # replacement as its "original" (the fall-back would recurse for ever: ADVICE r4)
            sub = type("Sub" + clsname, (cls,), {})
            assert fuse.fuse_model_class(sub) and fuse._ORIG not in sub.__dict__
            assert getattr(sub, fuse._ORIG) is cls.__dict__[fuse._ORIG]
            for name in (modname, "__init__"):
                sys.modules.pop(name, None)
        # other model files (ogbn_proteins/model.py: forward(x, node_index, edge_index, edge_attr) of a DeeperGCN with
        # edge encoders per layer) keep their forward
        d = os.path.join(ref_models.REF, "examples/ogb/ogbn_proteins")
        sys.path.insert(0, d)
        try:
            with redirect_stdout(io.StringIO()):
                mod = __import__("model")
        finally:
            sys.path.remove(d)
            for name in ("model", "__init__"):
                sys.modules.pop(name, None)
        assert fuse._ORIG not in mod.DeeperGCN.__dict__
    finally:
        deep_gcns_torch_amd.install(reference_root=ref_models.REF, fuse_models=False)      # the opt-out removes the hook
        assert fuse._FINDER not in sys.meta_path
    # the classes loaded by path (not through the hook) are fused explicitly; state_dict and CPU fallback
    with redirect_stdout(io.StringIO()):
        m = ref_models.arxiv_deepergcn(3)
    before = list(m.state_dict())
    fuse.fuse_model(m)
    assert list(m.state_dict()) == before and type(m).__name__ == "DeeperGCN"
    x, ei = torch.randn(20, 128), torch.randint(0, 20, (2, 60))
    assert not fuse._deepergcn_qualifies(m, x, ei)          # CPU tensors: the file's own loop runs
    from gcn_lib.sparse import torch_message
    import config_replays
    saved = torch_message.GenMessagePassing.propagate
    torch_message.GenMessagePassing.propagate = config_replays.oracle_propagate
    try:
        m.eval()
        with torch.no_grad():
            out = m(x, ei)
            ref = type(m).__dict__[fuse._ORIG](m, x, ei)
    finally:
        torch_message.GenMessagePassing.propagate = saved
    assert torch.equal(out, ref)
