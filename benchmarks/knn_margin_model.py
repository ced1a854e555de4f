#!/usr/bin/env python
"""How many rows of a layer end with a candidate list outside [K, CAP] under a given sample-threshold margin (CPU only).

The kNN filter kernel (csrc/knn_dense.hip) takes a row's threshold from the r-th smallest of n_s uniformly placed sample
distances.  The number of the N candidates at or below that threshold is  r + BetaBinomial(N - n_s, r, n_s - r + 1)  whatever
the data (order statistics of a continuous distribution), so the expected number of rows that must be redone by the exact
kernel follows from r alone.  knn_sample_rank() picks  r = ceil(min(K + z sqrt(unit K) + 2 unit, (K + CAP) / 2) / unit),
unit = N / n_s.  This script prints, for config 2 (N = 4096, n_s = 512, 32,768 rows per layer, K = 16 d), the expected rows
below K / above CAP for z = 3.2 (rounds 3 - 4) and for the rule of round 5 (4.8 below eight sample ranks, 4.0 above), next
to the best any rank could do.  Measured counterpart: profiles/r05_knn_by_dilation.md.

    python benchmarks/knn_margin_model.py
"""
import math

from scipy.stats import betabinom

N, NS, ROWS = 4096, 512, 8 * 4096
UNIT = N / NS


def expected_failures(r, K, cap):
    n = N - NS
    low = betabinom.cdf(K - r - 1, n, r, NS - r + 1)         # count < K
    high = betabinom.sf(cap - r, n, r, NS - r + 1)           # count > cap
    return ROWS * low, ROWS * high


def rank(K, cap, z):
    r0 = K / UNIT
    target = min(K + z * UNIT * math.sqrt(max(r0, 1.0)) + 2.0 * UNIT, 0.5 * (K + cap))
    return math.ceil(target / UNIT)


def main():
    print("| d | K | list | z = 3.2: rank, rows below K / above CAP | round 5: z, rank, rows below / above | best rank: rows outside |")
    print("|---|---|---|---|---|---|")
    for d in range(1, 29):
        K = 16 * d
        r0 = K / UNIT
        cap = 512 if K + 3.2 * UNIT * math.sqrt(max(r0, 1.0)) + 2.0 * UNIT + 96 <= 512 else 1024
        z5 = 4.8 if r0 < 8.0 else 4.0
        ra, rb = rank(K, cap, 3.2), rank(K, cap, z5)
        la, ha = expected_failures(ra, K, cap)
        lb, hb = expected_failures(rb, K, cap)
        best = min(range(max(2, K // 8), cap // 8), key=lambda r: sum(expected_failures(r, K, cap)))
        print(f"| {d} | {K} | {cap} | {ra}: {la:.2g} / {ha:.2g} | {z5}, {rb}: {lb:.2g} / {hb:.2g} | "
              f"{best}: {sum(expected_failures(best, K, cap)):.2g} |")


if __name__ == "__main__":
    main()
