"""Dense graph convolutions on (B,C,N,1) tensors: same classes, signatures and ``state_dict`` keys as
the reference's gcn_lib/dense/torch_vertex.py (MRConv2d :8-20, EdgeConv2d :23-35, GraphConv2d :38-52,
DynConv2d :55-72, Plain/Res/DenseDynBlock2d :75-116).  ``self.nn`` is still the BasicConv container
(checkpoints load unchanged); its parameters feed libdgcn's fused kernels instead of a per-edge conv."""
import torch
from torch import nn

from ... import dense_ops
from .torch_edge import DenseDilatedKnnGraph, DilatedKnnGraph
from .torch_nn import BasicConv, batched_index_select

__all__ = ["MRConv2d", "EdgeConv2d", "GraphConv2d", "DynConv2d", "PlainDynBlock2d", "ResDynBlock2d",
           "DenseDynBlock2d"]


def _act_code(stack):
    """(act code, slope) of the activation inside a BasicConv stack."""
    for m in stack:
        if isinstance(m, nn.ReLU):
            return dense_ops.ACT_RELU, 0.0
        if isinstance(m, nn.LeakyReLU):
            return dense_ops.ACT_LEAKY, float(m.negative_slope)
        if isinstance(m, nn.PReLU):
            raise NotImplementedError("EdgeConv2d: act='prelu' is not supported by the fused kernel")
    return dense_ops.ACT_NONE, 0.0


def _to_bcn1(rows):
    """(B,N,C) point-major rows -> contiguous (B,C,N,1)."""
    return rows.transpose(1, 2).unsqueeze(-1).contiguous()


class MRConv2d(nn.Module):
    """Max-Relative graph convolution: nn(cat[x, max_j (x_j - x_i)])."""

    def __init__(self, in_channels, out_channels, act='relu', norm=None, bias=True):
        super().__init__()
        self.nn = BasicConv([in_channels * 2, out_channels], act, norm, bias)

    def forward(self, x, edge_index):
        # x_i is constant over the neighbourhood and fl(a - b) is monotone in a:
        # max_j (x_j - x_i) == (max_j x_j) - x_i bit for bit -> one gather-max over point-major rows.
        dense_ops.check_centres(edge_index)
        rows = x.squeeze(-1).transpose(1, 2).contiguous()                       # (B,N,C)
        vmax, _, _, _ = dense_ops.edge_reduce(rows, edge_index[0], has_p=False)
        rel = _to_bcn1(vmax - rows)
        return self.nn(torch.cat([x, rel], dim=1))


def _with_skip(y, x, res_scale):
    """``y + x * res_scale`` as the reference writes it (gcn_lib/dense/torch_vertex.py:101); the multiplication by the
    default scale 1 is skipped (x * 1 == x bit for bit)."""
    if res_scale is None:
        return y
    return y + x if res_scale == 1 else y + x * res_scale


class EdgeConv2d(nn.Module):
    """Edge convolution max_j norm(act(W [x_i ; x_j - x_i] + b)) for dense data.

    W [x_i ; x_j - x_i] + b = ((W1 - W2) x_i + b) + W2 x_j = P_i + Q_j: the 1x1 conv runs once per
    vertex (fp32 MFMA), the edge kernel gathers Q rows, applies the activation and keeps only
    max_j, min_j and the BatchNorm sums; the affine norm is applied to the selected extreme."""

    def __init__(self, in_channels, out_channels, act='relu', norm=None, bias=True):
        super().__init__()
        self.nn = BasicConv([in_channels * 2, out_channels], act, norm, bias)
        # instance norm (statistics per sample) and PReLU (a learnable slope inside the batch statistics) do not fit
        # the vertex split: those two options run the reference's per-edge formulation (gcn_lib/dense/torch_vertex.py:
        # 45-53) on library ops -- same results and the reference's (B, 2C, N, k) memory, no fused kernel
        self._per_edge = any(isinstance(m, (nn.InstanceNorm2d, nn.PReLU)) for m in self.nn)
        if not self._per_edge:
            _act_code(self.nn)

    def _forward_per_edge(self, x, edge_index):
        x_i = batched_index_select(x, edge_index[1])
        x_j = batched_index_select(x, edge_index[0])
        return torch.max(self.nn(torch.cat([x_i, x_j - x_i], dim=1)), -1, keepdim=True)[0]

    def forward(self, x, edge_index, res_scale=None):
        """res_scale (not in the reference's signature): the caller's ``+ x * res_scale`` done in the last kernel."""
        if self._per_edge:
            return _with_skip(self._forward_per_edge(x, edge_index), x, res_scale)
        dense_ops.check_centres(edge_index)
        conv = self.nn[0]
        act, slope = _act_code(self.nn)
        bn = next((m for m in self.nn if isinstance(m, nn.BatchNorm2d)), None)
        if conv.out_channels % 4 == 0:
            # fused path: P/Q GEMM straight from the conv weight, edge kernel, BN finalize, transposing apply
            if res_scale is not None and (x.dim() != 4 or conv.out_channels * 2 != conv.in_channels or not x.is_floating_point()
                                          or x.dtype != torch.float32):
                return _with_skip(dense_ops.edgeconv2d_fused(x, conv.weight, conv.bias, edge_index[0], act, slope, bn),
                                  x, res_scale)
            return dense_ops.edgeconv2d_fused(x, conv.weight, conv.bias, edge_index[0], act, slope, bn, res_scale)
        return _with_skip(self._forward_composed(x, edge_index, conv, act, slope, bn), x, res_scale)

    def _forward_composed(self, x, edge_index, conv, act, slope, bn):
        """Same math from separately differentiable pieces (channel counts that are not a multiple of 4)."""
        cin = conv.in_channels // 2
        cout = conv.out_channels
        W = conv.weight.view(cout, 2 * cin)
        w1, w2 = W[:, :cin], W[:, cin:]
        wcat = torch.cat([(w1 - w2).t(), w2.t()], dim=1)                        # (C, 2C')
        bcat = None
        if conv.bias is not None:
            bcat = torch.cat([conv.bias, torch.zeros_like(conv.bias)])
        pq = dense_ops.vertex_gemm(x, wcat, bcat)                               # (B,N,2C') = [P | Q]
        if bn is None:
            vmax, _, _, _ = dense_ops.edge_reduce(pq, edge_index[0], True, act, slope)
            return _to_bcn1(vmax)
        use_batch = bn.training or not bn.track_running_stats
        vmax, vmin, s1, s2 = dense_ops.edge_reduce(pq, edge_index[0], True, act, slope,
                                                   need_min=True, need_stats=use_batch)
        if use_batch:
            count = float(x.size(0) * x.size(2) * edge_index.size(-1))
            mean = s1 / count                                                   # float64, (C',)
            var = (s2 / count - mean * mean).clamp_min(0.0)                      # biased batch variance
            if bn.training and bn.track_running_stats:
                with torch.no_grad():
                    bn.num_batches_tracked += 1
                    mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                    bn.running_mean.mul_(1 - mom).add_((mom * mean).to(bn.running_mean.dtype))
                    unbiased = var * (count / max(count - 1.0, 1.0))
                    bn.running_var.mul_(1 - mom).add_((mom * unbiased).to(bn.running_var.dtype))
        else:
            mean, var = bn.running_mean.double(), bn.running_var.double()
        scale = bn.weight.double() * torch.rsqrt(var + bn.eps)
        shift = bn.bias.double() - mean * scale
        scale, shift = scale.float(), shift.float()
        sel = torch.where(scale >= 0, vmax, vmin)                               # BN is affine per channel
        return _to_bcn1(sel * scale + shift)


class GraphConv2d(nn.Module):
    """Static graph convolution layer (conv in edge|mr)."""

    def __init__(self, in_channels, out_channels, conv='edge', act='relu', norm=None, bias=True):
        super().__init__()
        if conv == 'edge':
            self.gconv = EdgeConv2d(in_channels, out_channels, act, norm, bias)
        elif conv == 'mr':
            self.gconv = MRConv2d(in_channels, out_channels, act, norm, bias)
        else:
            raise NotImplementedError('conv:{} is not supported'.format(conv))

    def forward(self, x, edge_index, res_scale=None):
        if res_scale is None:
            return self.gconv(x, edge_index)
        if isinstance(self.gconv, EdgeConv2d):
            return self.gconv(x, edge_index, res_scale)
        return _with_skip(self.gconv(x, edge_index), x, res_scale)


class DynConv2d(GraphConv2d):
    """Dynamic graph convolution: dilated kNN graph of the current features, then GraphConv2d."""

    def __init__(self, in_channels, out_channels, kernel_size=9, dilation=1, conv='edge', act='relu',
                 norm=None, bias=True, stochastic=False, epsilon=0.0, knn='matrix'):
        super().__init__(in_channels, out_channels, conv, act, norm, bias)
        self.k = kernel_size
        self.d = dilation
        graph_cls = DenseDilatedKnnGraph if knn == 'matrix' else DilatedKnnGraph
        self.dilated_knn_graph = graph_cls(kernel_size, dilation, stochastic, epsilon)

    def forward(self, x, edge_index=None, res_scale=None):
        if edge_index is None:
            edge_index = self.dilated_knn_graph(x)
        return super().forward(x, edge_index, res_scale)


class PlainDynBlock2d(nn.Module):
    def __init__(self, in_channels, kernel_size=9, dilation=1, conv='edge', act='relu', norm=None,
                 bias=True, stochastic=False, epsilon=0.0, knn='matrix'):
        super().__init__()
        self.body = DynConv2d(in_channels, in_channels, kernel_size, dilation, conv, act, norm, bias,
                              stochastic, epsilon, knn)

    def forward(self, x, edge_index=None):
        return self.body(x, edge_index)


class ResDynBlock2d(nn.Module):
    def __init__(self, in_channels, kernel_size=9, dilation=1, conv='edge', act='relu', norm=None,
                 bias=True, stochastic=False, epsilon=0.0, knn='matrix', res_scale=1):
        super().__init__()
        self.body = DynConv2d(in_channels, in_channels, kernel_size, dilation, conv, act, norm, bias,
                              stochastic, epsilon, knn)
        self.res_scale = res_scale

    def forward(self, x, edge_index=None):
        # self.body(x, edge_index) + x * self.res_scale, the skip connection added by the convolution's last kernel
        return self.body(x, edge_index, self.res_scale)


class DenseDynBlock2d(nn.Module):
    def __init__(self, in_channels, out_channels=64, kernel_size=9, dilation=1, conv='edge', act='relu',
                 norm=None, bias=True, stochastic=False, epsilon=0.0, knn='matrix'):
        super().__init__()
        self.body = DynConv2d(in_channels, out_channels, kernel_size, dilation, conv, act, norm, bias,
                              stochastic, epsilon, knn)

    def forward(self, x, edge_index=None):
        return torch.cat((x, self.body(x, edge_index)), 1)
