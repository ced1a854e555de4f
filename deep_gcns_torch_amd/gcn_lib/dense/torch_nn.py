"""Building blocks of the dense (B,C,N,1) library: same names, signatures and ``state_dict`` keys
as the reference's gcn_lib/dense/torch_nn.py (act_layer :9-21, norm_layer :24-33, MLP :36-45,
BasicConv :48-72, batched_index_select :75-96).  BasicConv/MLP are stock PyTorch containers; the
graph convolutions in torch_vertex.py read their parameters and run the fused HIP path instead of
calling them per edge."""
import torch
from torch import nn

__all__ = ["act_layer", "norm_layer", "MLP", "BasicConv", "batched_index_select"]


def act_layer(act, inplace=False, neg_slope=0.2, n_prelu=1):
    kind = act.lower()
    if kind == 'relu':
        return nn.ReLU(inplace)
    if kind == 'leakyrelu':
        return nn.LeakyReLU(neg_slope, inplace)
    if kind == 'prelu':
        return nn.PReLU(num_parameters=n_prelu, init=neg_slope)
    raise NotImplementedError('activation layer [%s] is not found' % kind)


def norm_layer(norm, nc):
    kind = norm.lower()
    if kind == 'batch':
        return nn.BatchNorm2d(nc, affine=True)
    if kind == 'instance':
        return nn.InstanceNorm2d(nc, affine=False)
    raise NotImplementedError('normalization layer [%s] is not found' % kind)


def _on(opt):
    return opt is not None and opt.lower() != 'none'


class MLP(nn.Sequential):
    """Linear -> act -> norm stack (note: every norm is sized by channels[-1], as in the reference)."""

    def __init__(self, channels, act='relu', norm=None, bias=True):
        stages = []
        for cin, cout in zip(channels[:-1], channels[1:]):
            stages.append(nn.Linear(cin, cout, bias))
            if _on(act):
                stages.append(act_layer(act))
            if _on(norm):
                stages.append(norm_layer(norm, channels[-1]))
        super().__init__(*stages)


class BasicConv(nn.Sequential):
    """1x1 Conv2d -> act -> norm -> Dropout2d per layer (activation BEFORE normalisation);
    kaiming-normal conv weights, zero bias, unit/zero norm affine."""

    def __init__(self, channels, act='relu', norm=None, bias=True, drop=0.):
        stages = []
        for cin, cout in zip(channels[:-1], channels[1:]):
            stages.append(nn.Conv2d(cin, cout, 1, bias=bias))
            if _on(act):
                stages.append(act_layer(act))
            if _on(norm):
                stages.append(norm_layer(norm, channels[-1]))
            if drop > 0:
                stages.append(nn.Dropout2d(drop))
        super().__init__(*stages)
        self.reset_parameters()

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d)):
                if m.weight is not None:
                    m.weight.data.fill_(1)
                    m.bias.data.zero_()


def batched_index_select(x, idx):
    """out[b,c,n,l] = x[b,c,idx[b,n,l]]  -- x (B,C,N,1), idx (B,N,l) -> (B,C,N,l).

    Public helper kept for API parity.  The graph convolutions of this package never call it: their
    kernels gather neighbour rows on the fly and the (B,C,N,l) tensor is not materialised."""
    B, C, N = x.shape[:3]
    l = idx.shape[-1]
    src = x.reshape(B, C, N)
    flat = idx.reshape(B, 1, N * l).expand(B, C, N * l)
    return torch.gather(src, 2, flat).view(B, C, N, l)
