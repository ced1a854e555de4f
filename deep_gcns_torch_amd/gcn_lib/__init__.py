"""Drop-in `gcn_lib` package: same module paths and class names as the reference's gcn_lib,
hot path backed by libdgcn (HIP, gfx950)."""
