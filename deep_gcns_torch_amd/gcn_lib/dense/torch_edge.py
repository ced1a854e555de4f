"""Dense dilated-kNN graph construction on (B,C,N,1) features: public surface of the reference's
gcn_lib/dense/torch_edge.py (DenseDilated :6-29, pairwise_distance :32-42, dense_knn_matrix :45-58,
DenseDilatedKnnGraph :61-76, DilatedKnnGraph :79-101).  Distances, the sorted top-(k*d) selection and
the dilation are ONE HIP kernel (dgcn_knn_dense_f32): no (B,N,N) matrix, no (B,N,k*d) index list."""
import torch
from torch import nn

from ... import dense_ops

__all__ = ["DenseDilated", "pairwise_distance", "dense_knn_matrix", "DenseDilatedKnnGraph", "DilatedKnnGraph"]


def _take_random_k(edge_index, k, dilation):
    """Stochastic dilation: a random k of the k*d neighbours (reference :22-24)."""
    pick = torch.randperm(k * dilation)[:k]
    return edge_index[:, :, :, pick]


class DenseDilated(nn.Module):
    """Select the dilated neighbours from a full (2,B,N,k*d) neighbour list.  In stochastic mode the
    CPU RNG is consumed exactly like the reference: `torch.rand(1)` is drawn first (even in eval),
    `torch.randperm` only when the random branch is taken."""

    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.dilation = dilation
        self.stochastic = stochastic
        self.epsilon = epsilon
        self.k = k

    def _random_branch(self):
        return bool(self.stochastic and torch.rand(1) < self.epsilon and self.training)

    def forward(self, edge_index):
        if self._random_branch():
            return _take_random_k(edge_index, self.k, self.dilation)
        return edge_index[:, :, :, ::self.dilation]


def pairwise_distance(x):
    """(B,N,C) -> (B,N,N) squared distances with the reference's fp32 association order.
    Kept as a public helper; the kNN kernel computes the same values tile by tile in LDS instead."""
    inner = -2 * torch.matmul(x, x.transpose(2, 1))
    sq = torch.sum(torch.mul(x, x), dim=-1, keepdim=True)
    return sq + inner + sq.transpose(2, 1)


def dense_knn_matrix(x, k=16):
    """x (B,C,N,1) -> (2,B,N,k) int64: k nearest neighbours (self included, ascending distance) and
    the centre ids."""
    return dense_ops.knn_edge_index(x, k, 1)


class DenseDilatedKnnGraph(nn.Module):
    """edge_index of the dilated kNN graph of x (B,C,N,1): (2,B,N,k)."""

    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.dilation = dilation
        self.stochastic = stochastic
        self.epsilon = epsilon
        self.k = k
        self._dilated = DenseDilated(k, dilation, stochastic, epsilon)
        self.knn = dense_knn_matrix

    def forward(self, x):
        if self._dilated._random_branch():          # same RNG stream as DenseDilated.forward
            full = dense_ops.knn_edge_index(x, self.k * self.dilation, 1)
            return _take_random_k(full, self.k, self.dilation)
        return dense_ops.knn_edge_index(x, self.k, self.dilation)   # dilation fused into the kernel


class DilatedKnnGraph(nn.Module):
    """The reference's torch_cluster variant (:79-101): per sample, the k*d nearest neighbours of every point
    EXCLUDING the point itself (knn_graph(..., loop=False)), grouped by centre, then dilated.  Served by the
    same brute-force HIP kernel with the query point masked out (exact, not a tree)."""

    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.dilation = dilation
        self.stochastic = stochastic
        self.epsilon = epsilon
        self.k = k
        self._dilated = DenseDilated(k, dilation, stochastic, epsilon)
        self.knn = dense_ops.knn_edge_index

    def forward(self, x):
        if self._dilated._random_branch():
            return _take_random_k(self.knn(x, self.k * self.dilation, 1, exclude_self=True), self.k, self.dilation)
        return self.knn(x, self.k, self.dilation, exclude_self=True)
