"""deep_gcns_torch_amd -- MI355X (gfx950) native message-passing hot path of
lightaime/deep_gcns_torch behind the reference's own `gcn_lib` module API.

    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()            # `import gcn_lib` / `import utils` now resolve here
    from gcn_lib.sparse.torch_vertex import GENConv

The hot path (kNN graph build, neighbour gather, edge MLP, scatter-{max,softmax,powermean})
runs only as hand-written HIP kernels in csrc/libdgcn.so; there is no CPU or eager fallback.
"""
import importlib
import os
import sys

__version__ = "0.1.0"

_SUBMODULES = (
    "gcn_lib", "gcn_lib.sparse", "gcn_lib.sparse.torch_nn", "gcn_lib.sparse.torch_edge",
    "gcn_lib.sparse.torch_message", "gcn_lib.sparse.torch_vertex",
    "gcn_lib.dense", "gcn_lib.dense.torch_nn", "gcn_lib.dense.torch_edge", "gcn_lib.dense.torch_vertex",
    "utils", "utils.pyg_util", "utils.data_util",
    "eff_gcn_modules", "eff_gcn_modules.rev", "eff_gcn_modules.rev.gcn_revop", "eff_gcn_modules.rev.memgcn",
    "eff_gcn_modules.rev.rev_layer",
)


def install(reference_root=None, fuse_models=True):
    """Register this package's `gcn_lib` and `utils` under their top-level names so the reference's
    example scripts (`from gcn_lib.sparse.torch_vertex import GENConv`, `from utils.pyg_util import
    scatter_`, ...) import them unchanged.  With ``reference_root`` the reference's own `utils/`
    directory is appended to `utils.__path__`, so `utils.ckpt_util`, `utils.metrics`, `utils.optim`,
    ... (pure-torch helpers outside the hot path) keep resolving to the reference's files, and
    `eff_gcn_modules` / `examples` become importable from there.

    ``fuse_models=True`` (default since round 5; ``False`` opts out and removes the hook): the layer loops the reference writes in its MODEL files (examples/ogb/ogbn_arxiv/model.py:88-106,
    ogbn_products/model.py, ogb_eff/ogbn_proteins/model_rev.py:98-99) are routed through ``blocks.res_plus_layer`` /
    ``blocks.ComposedEdgeEmbedding`` when those files are imported -- unchanged files, same ``state_dict``, same values
    (deep_gcns_torch_amd/fuse.py; ``fuse.fuse_model(instance)`` does it for one object)."""
    for name in _SUBMODULES:
        mod = importlib.import_module(f"{__name__}.{name}")
        sys.modules[name] = mod
    if reference_root is not None:
        ref_utils = os.path.join(reference_root, "utils")
        utils_mod = sys.modules["utils"]
        if os.path.isdir(ref_utils) and ref_utils not in utils_mod.__path__:
            utils_mod.__path__.append(ref_utils)
        if reference_root not in sys.path:
            sys.path.append(reference_root)
    if os.environ.get("DGCN_NO_WARMUP") != "1":
        try:
            from . import graph
            graph.warm_up()                    # graph-build module load off the first real build (GPU hosts only)
        except Exception:                      # noqa: BLE001 -- a warm-up must never keep install() from registering modules
            pass
    from . import fuse
    if fuse_models:
        fuse.enable_import_hook()
    else:
        fuse.disable_import_hook()
    return sys.modules["gcn_lib"]
