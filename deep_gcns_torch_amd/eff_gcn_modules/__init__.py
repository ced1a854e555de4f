"""Drop-in for the reference's `eff_gcn_modules` package (reversible GNN blocks, eff_gcn_modules/rev/*) with the
reversible step fused around the HIP message-passing kernels."""
