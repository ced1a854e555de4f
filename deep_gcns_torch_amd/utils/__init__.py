"""Dependency-light stand-in for the reference's `utils` package: only what
`gcn_lib` itself imports (pyg_util.scatter_, data_util feature tables / partition helpers).
`deep_gcns_torch_amd.install(reference_root=...)` can append the reference's own utils directory
to this package's __path__ so that `utils.ckpt_util`, `utils.metrics`, ... keep resolving."""
