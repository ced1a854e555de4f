"""CPU: libdgcn.so loads and exports every symbol include/dgcn.h declares (no compute calls)."""
import os
import re

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dgcn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dgcn_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_loads_and_exports_header_symbols():
    import __graft_entry__
    __graft_entry__.build()
    from deep_gcns_torch_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert declared, "no prototypes parsed from include/dgcn.h"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in dgcn.h but not exported by libdgcn.so"
    # the ctypes table covers exactly the header (a prototype without a binding is a bug)
    assert sorted(_lib.exported_symbols()) == declared
    assert lib.dgcn_version() == 100
    assert lib.dgcn_strerror(0) == b"ok"
    assert b"NULL" in lib.dgcn_strerror(-1)


def test_argument_errors_do_not_launch():
    """Negative return codes are produced on the host before any launch: safe without a GPU."""
    from deep_gcns_torch_amd import _lib
    lib = _lib.load()
    g = _lib.DgcnGraph()
    g.n_dst, g.n_src, g.n_edges = 4, 4, 0
    assert lib.dgcn_gen_aggr_fwd_f32(None, None, 0, None, 4, 3, 1, 0, 1.0, 1.0, 1e-7, None, None, None,
                                     None, None, None, None, 0, None) == -1
    import ctypes as C
    buf = (C.c_float * 64)()
    ptr = C.addressof(buf)
    # bad mode
    assert lib.dgcn_gen_aggr_fwd_f32(C.byref(g), ptr, 4, None, 4, 99, 1, 0, 1.0, 1.0, 1e-7, None, None, ptr,
                                     None, None, None, None, 0, None) == -4
    # bad stride
    assert lib.dgcn_gen_aggr_fwd_f32(C.byref(g), ptr, 2, None, 4, 3, 1, 0, 1.0, 1.0, 1e-7, None, None, ptr,
                                     None, None, None, None, 0, None) == -2
    assert lib.dgcn_selftest_axpy_f32(1.0, None, None, 4, None) == -1
    # the max-winner backward of the fused edge GEMM: null inputs, channel / feature limits, row strides, nothing to do
    big = (C.c_float * 4096)()
    bp = (C.addressof(big) + 15) & ~15
    assert lib.dgcn_egemm_max_bwd_num_partials(0) == 0 and lib.dgcn_egemm_max_bwd_num_partials(13253) == 255
    assert lib.dgcn_egemm_max_bwd_f32(None, bp, 4, 8, bp, 16, bp, 16, 8, bp, 16, bp, None) == -1
    assert lib.dgcn_egemm_max_bwd_f32(bp, bp, 4, 8, bp, 16, bp, 16, 132, bp, 16, bp, None) == -2      # channels > 128
    assert lib.dgcn_egemm_max_bwd_f32(bp, bp, 4, 8, bp, 16, bp, 18, 8, bp, 16, bp, None) == -2       # n_feat % 4
    assert lib.dgcn_egemm_max_bwd_f32(bp, bp, 4, 8, bp, 16, bp, 16, 8, bp, 12, bp, None) == -2       # gradient stride < n_feat
    assert lib.dgcn_egemm_max_bwd_f32(bp, bp, 4, 8, bp + 4, 16, bp, 16, 8, bp, 16, bp, None) == -3    # misaligned feature rows
    assert lib.dgcn_egemm_max_bwd_f32(bp, bp, 4, 8, None, 16, bp, 16, 8, bp, 16, bp, None) == -1      # dW needs the features
    assert lib.dgcn_egemm_max_bwd_f32(bp, bp, 0, 8, bp, 16, bp, 16, 8, bp, 16, bp, None) == 0         # no rows: no launch
    assert lib.dgcn_egemm_max_bwd_f32(bp, bp, 4, 8, bp, 16, bp, 16, 8, None, 0, None, None) == 0      # nothing requested


def test_hot_path_refuses_cpu_tensors():
    import pytest
    import torch
    from deep_gcns_torch_amd import ops
    x = torch.randn(4, 8)
    ei = torch.tensor([[0, 1], [1, 2]])
    with pytest.raises(RuntimeError, match="no CPU fallback|There is no CPU fallback"):
        ops.gen_aggregate(x, ei, aggr="softmax")
