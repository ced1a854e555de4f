"""ORACLE / TEST INFRASTRUCTURE ONLY.

Imports the REAL reference modules from /root/reference in this container so that golden
vectors come from the reference's own code.  The Python packages the reference needs but
that are not installed (torch_scatter, torch_geometric, torch_cluster, h5py) are replaced by
modules built from oracle/thirdparty.py (restated published algorithms).  Nothing here runs
on the GPU box: /root/reference does not exist there, only the committed fixtures travel.
"""
from __future__ import annotations

import importlib
import sys
import types

REFERENCE_ROOT = "/root/reference"


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    from oracle import thirdparty as tp

    class _Dummy:  # placeholder base classes for wrappers that are out of scope
        def __init__(self, *a, **k):
            raise NotImplementedError("out-of-scope PyG class (SURVEY.md §2a #5)")

    _module("torch_scatter", scatter=tp.scatter, scatter_add=tp.scatter_add, scatter_sum=tp.scatter_sum,
            scatter_mean=tp.scatter_mean, scatter_max=tp.scatter_max, scatter_min=tp.scatter_min,
            scatter_softmax=tp.scatter_softmax)
    _module("torch_cluster", knn_graph=tp.knn_graph)
    tg_nn = _module("torch_geometric.nn", MessagePassing=tp.MessagePassing, EdgeConv=tp.EdgeConv,
                    GATConv=_Dummy, SAGEConv=_Dummy, GCNConv=_Dummy, GINConv=_Dummy)
    tg_utils = _module("torch_geometric.utils", degree=tp.degree, remove_self_loops=tp.remove_self_loops,
                       add_self_loops=tp.add_self_loops)
    tg_data = _module("torch_geometric.data", InMemoryDataset=object, Data=object, extract_zip=None)
    _module("torch_geometric", nn=tg_nn, utils=tg_utils, data=tg_data)
    _module("h5py")


def import_reference():
    """Returns the reference's (gcn_lib.dense, gcn_lib.sparse) packages."""
    sys.dont_write_bytecode = True  # keep /root/reference free of __pycache__
    install_stubs()
    for name in [n for n in sys.modules if n == "gcn_lib" or n.startswith("gcn_lib.")
                 or n == "utils" or n.startswith("utils.")]:
        del sys.modules[name]
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    dense = importlib.import_module("gcn_lib.dense")
    sparse = importlib.import_module("gcn_lib.sparse")
    assert dense.__file__.startswith(REFERENCE_ROOT) and sparse.__file__.startswith(REFERENCE_ROOT)
    return dense, sparse
