"""Sparse-layout (flattened (2, N*k) edge lists) dilated kNN graphs -- public surface of the
reference's gcn_lib/sparse/torch_edge.py (Dilated :6-29, DilatedKnnGraph :32-50, knn_matrix :66-91,
knn_graph_matrix :94-104).  The neighbour search itself is libdgcn's fused distance+select kernel
(the same one the dense library uses); the tree-based torch_cluster variant is not provided.
"""
import torch
from torch import nn

__all__ = ["Dilated", "DilatedKnnGraph", "knn_matrix", "knn_graph_matrix", "knn_graph"]


class Dilated(nn.Module):
    """Keep every d-th of the k*d sorted neighbours of each point; in stochastic mode, with
    probability epsilon (training only), keep a random k of them instead.  The CPU RNG calls
    (`torch.rand(1)` first, then `torch.randperm`) mirror the reference's stream exactly."""

    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.dilation = dilation
        self.stochastic = stochastic
        self.epsilon = epsilon
        self.k = k

    def forward(self, edge_index, batch=None):
        if self.stochastic and torch.rand(1) < self.epsilon and self.training:
            num = self.k * self.dilation
            pick = torch.randperm(num)[:self.k]
            return edge_index.view(2, -1, num)[:, :, pick].reshape(2, -1)
        return edge_index[:, ::self.dilation]


def _knn_flat(x, k, batch, exclude_self, allow_ragged=False):
    from ... import dense_ops
    with torch.no_grad():
        n_clouds = 1 if batch is None else int(batch[-1]) + 1
        equal = x.shape[0] % n_clouds == 0
        if equal and batch is not None and n_clouds > 1 and allow_ragged:
            n_each = x.shape[0] // n_clouds                       # equal count is not enough: check the boundaries
            equal = bool((batch[::n_each] == torch.arange(n_clouds, device=batch.device)).all()) and \
                bool((batch[n_each - 1::n_each] == torch.arange(n_clouds, device=batch.device)).all())
        if not equal:
            if not allow_ragged:
                raise NotImplementedError("kNN kernel needs equally sized clouds (N_total divisible by batch size)")
            return _knn_ragged(x, k, batch, exclude_self)
        pts = x.detach().reshape(n_clouds, -1, x.shape[-1])
        n_points = pts.shape[1]
        nn_idx = dense_ops.knn_indices(pts, k, 1, exclude_self)                  # (B, N, k) int64
        nn_idx = nn_idx + torch.arange(0, n_points * n_clouds, n_points, device=x.device).view(n_clouds, 1, 1)
        center = torch.arange(0, n_points * n_clouds, device=x.device).repeat_interleave(k)
    return nn_idx.reshape(1, -1), center.view(1, -1)


def _knn_ragged(x, k, batch, exclude_self):
    """Clouds of different sizes (torch_cluster.knn_graph accepts any sorted ``batch`` vector,
    gcn_lib/dense/torch_edge.py:97, gcn_lib/sparse/torch_edge.py:46): one kernel launch per cloud on its slice.
    ``batch`` must be sorted (as PyG batches are); every cloud needs at least k (+1 without self) points."""
    from ... import dense_ops
    sizes = torch.bincount(batch).tolist()
    nn_parts, ctr_parts, start = [], [], 0
    for n_b in sizes:
        if n_b == 0:
            continue
        pts = x.detach()[start:start + n_b].unsqueeze(0)
        idx = dense_ops.knn_indices(pts, k, 1, exclude_self)[0] + start         # (n_b, k)
        nn_parts.append(idx.reshape(-1))
        ctr_parts.append(torch.arange(start, start + n_b, device=x.device).repeat_interleave(k))
        start += n_b
    return torch.cat(nn_parts).view(1, -1), torch.cat(ctr_parts).view(1, -1)


def knn_matrix(x, k=16, batch=None):
    """(N_total, C) features of equally sized clouds -> (nn_idx, center_idx), each (1, N_total*k),
    neighbours sorted by ascending distance, self included, ids offset per cloud."""
    return _knn_flat(x, k, batch, False)


def knn_graph(x, k, batch=None, loop=False, flow="source_to_target"):
    """Stand-in for torch_cluster.knn_graph: (2, N_total*k), row 0 = neighbour, row 1 = centre, grouped by centre,
    self excluded unless ``loop`` (exact brute force on the GPU).  Equally sized clouds go through one batched launch,
    ragged batches through one launch per cloud."""
    nn_idx, center_idx = _knn_flat(x, k, batch, not loop, allow_ragged=True)
    return torch.cat((nn_idx, center_idx), dim=0)


def knn_graph_matrix(x, k=16, batch=None):
    nn_idx, center_idx = knn_matrix(x, k, batch)
    return torch.cat((nn_idx, center_idx), dim=0)


class DilatedKnnGraph(nn.Module):
    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0, knn='matrix'):
        super().__init__()
        self.dilation = dilation
        self.stochastic = stochastic
        self.epsilon = epsilon
        self.k = k
        self._dilated = Dilated(k, dilation, stochastic, epsilon)
        self.knn = knn_graph_matrix if knn == 'matrix' else knn_graph

    def forward(self, x, batch):
        return self._dilated(self.knn(x, self.k * self.dilation, batch), batch)
