"""Test helper: the reference's REVERSIBLE model path restated on top of whichever `gcn_lib` is registered in
sys.modules (the reference's own files cannot travel to the GPU box).  Attribute names follow the reference so
its state_dicts load unchanged; the golden fixture (tests/golden/revgcn.pt) comes from the reference's REAL
files (oracle/make_golden.py: examples/ogb_eff/ogbn_proteins/model_rev.py on eff_gcn_modules/rev/*).

  ReversibleFn / InvertibleModuleWrapper   eff_gcn_modules/rev/gcn_revop.py:17-157,160-268
      forward under no_grad on detached inputs, the node-feature INPUT storage is freed, backward re-creates the
      input with fn.inverse (freeing the output storage), re-runs fn with grad enabled and calls autograd.grad
  GroupAdditiveCoupling                      eff_gcn_modules/rev/memgcn.py:8-52
  SharedDropout / GENBlock                   eff_gcn_modules/rev/rev_layer.py:12-75
  RevGCN                                     examples/ogb_eff/ogbn_proteins/model_rev.py:12-112
"""
import copy

import torch
import torch.nn.functional as F
from torch import nn


class ReversibleFn(torch.autograd.Function):
    """y = fn(x, *rest) without keeping x: x is rebuilt from y in the backward."""

    @staticmethod
    def forward(ctx, fn, fn_inverse, n_inputs, *inputs_and_weights):
        inputs = inputs_and_weights[:n_inputs]
        ctx.fn, ctx.fn_inverse = fn, fn_inverse
        ctx.weights = inputs_and_weights[n_inputs:]
        ctx.requires = [t.requires_grad for t in inputs]
        with torch.no_grad():
            y = fn(*[t.detach() if isinstance(t, torch.Tensor) else t for t in inputs])
        y = y.detach_()
        inputs[0].untyped_storage().resize_(0)          # gcn_revop.py:62-67: only the node features are dropped
        ctx.inputs, ctx.output = inputs, y
        return y

    @staticmethod
    def backward(ctx, grad_y):
        inputs, y = ctx.inputs, ctx.output
        with torch.no_grad():                            # gcn_revop.py:98-119
            x = ctx.fn_inverse(y, *inputs[1:])
            y.untyped_storage().resize_(0)
            inputs[0].untyped_storage().resize_(x.numel() * x.element_size())
            inputs[0].set_(x)
        with torch.enable_grad():                        # gcn_revop.py:121-133
            det = []
            for t, req in zip(inputs, ctx.requires):
                d = t.detach()
                d.requires_grad = req
                det.append(d)
            out = ctx.fn(*det)
        wrt = [d for d in det if d.requires_grad]
        grads = torch.autograd.grad(out, tuple(wrt) + tuple(ctx.weights), grad_y)
        it = iter(grads[:len(wrt)])
        gin = [next(it) if req else None for req in ctx.requires]
        return (None, None, None) + tuple(gin) + tuple(grads[len(wrt):])


class InvertibleModuleWrapper(nn.Module):
    def __init__(self, fn, keep_input=False):
        super().__init__()
        assert not keep_input
        self._fn = fn

    def forward(self, *xin):
        ws = tuple(p for p in self._fn.parameters() if p.requires_grad)
        return ReversibleFn.apply(self._fn.forward, self._fn.inverse, len(xin), *(xin + ws))


class GroupAdditiveCoupling(nn.Module):
    def __init__(self, Fms, split_dim=-1, group=2):
        super().__init__()
        self.Fms, self.split_dim, self.group = Fms, split_dim, group

    def _chunks(self, args):
        per_arg = [torch.chunk(a, self.group, dim=self.split_dim) for a in args]
        return list(zip(*per_arg))

    def forward(self, x, edge_index, *args):
        xs = torch.chunk(x, self.group, dim=self.split_dim)
        extra = self._chunks(args)
        y_in = sum(xs[1:])
        ys = []
        for i in range(self.group):
            y_in = xs[i] + self.Fms[i](y_in, edge_index, *extra[i])
            ys.append(y_in)
        return torch.cat(ys, dim=self.split_dim)

    def inverse(self, y, edge_index, *args):
        ys = torch.chunk(y, self.group, dim=self.split_dim)
        extra = self._chunks(args)
        xs = []
        for i in range(self.group - 1, -1, -1):
            y_in = ys[i - 1] if i != 0 else sum(xs)
            xs.append(ys[i] - self.Fms[i](y_in, edge_index, *extra[i]))
        return torch.cat(xs[::-1], dim=self.split_dim)


class SharedDropout(nn.Module):
    def __init__(self):
        super().__init__()
        self.mask = None

    def set_mask(self, mask):
        self.mask = mask

    def forward(self, x):
        return x * self.mask if self.training else x


class GENBlock(nn.Module):
    """norm -> ReLU -> shared dropout -> GENConv  (rev_layer.py:27-75)."""

    def __init__(self, in_channels, out_channels, norm="layer", **gen_kw):
        super().__init__()
        from gcn_lib.sparse.torch_nn import norm_layer
        from gcn_lib.sparse.torch_vertex import GENConv
        self.norm = norm_layer(norm, in_channels)
        self.dropout = SharedDropout()
        self.gcn = GENConv(in_channels, out_channels, norm=norm, **gen_kw)

    def forward(self, x, edge_index, dropout_mask=None, edge_emb=None):
        out = F.relu(self.norm(x))
        if dropout_mask is not None:
            self.dropout.set_mask(dropout_mask)
        out = self.dropout(out)
        return self.gcn(out, edge_index, edge_emb) if edge_emb is not None else self.gcn(out, edge_index)


class RevGCN(nn.Module):
    def __init__(self, num_layers=3, hidden=64, group=2, num_tasks=112, aggr="max", dropout=0.2, t=1.0, learn_t=False,
                 p=1.0, learn_p=False, norm="layer", mlp_layers=2, conv_encode_edge=True, node_table=None,
                 use_one_hot_encoding=True, impl="restated", composed_edges=False):
        """impl='restated': the wrapper / coupling / block classes of this file (the reference's algorithm);
        impl='product': the package's own eff_gcn_modules.rev drop-ins (fused inverse + recompute, in-place
        accumulation of the shared edge-embedding gradient), as model_rev.py would import them after install()."""
        super().__init__()
        from gcn_lib.sparse.torch_nn import norm_layer
        if impl == "product":
            from eff_gcn_modules.rev import memgcn as _mem, rev_layer as _rl
            _Block, _Coupling, _Wrapper = _rl.GENBlock, _mem.GroupAdditiveCoupling, _mem.InvertibleModuleWrapper
        else:
            _Block, _Coupling, _Wrapper = GENBlock, GroupAdditiveCoupling, InvertibleModuleWrapper
        self.num_layers, self.dropout, self.group = num_layers, dropout, group
        self.composed_edges = composed_edges
        self.use_one_hot_encoding = use_one_hot_encoding
        self.gcns = nn.ModuleList()
        self.last_norm = norm_layer(norm, hidden)
        for _ in range(num_layers):
            fm = _Block(hidden // group, hidden // group, norm=norm, aggr=aggr, t=t, learn_t=learn_t, p=p,
                        learn_p=learn_p, y=0.0, learn_y=False, msg_norm=False, learn_msg_scale=False,
                        encode_edge=conv_encode_edge, edge_feat_dim=hidden, mlp_layers=mlp_layers)
            Fms = nn.ModuleList([fm] + [copy.deepcopy(fm) for _ in range(group - 1)])
            self.gcns.append(_Wrapper(_Coupling(Fms, group=group), keep_input=False))
        self.node_features = node_table                      # a plain attribute in the reference (not a buffer)
        if use_one_hot_encoding:
            self.node_one_hot_encoder = nn.Linear(8, 8)
            self.node_features_encoder = nn.Linear(16, hidden)
        else:
            self.node_features_encoder = nn.Linear(8, hidden)
        self.edge_encoder = nn.Linear(8, hidden)
        self.node_pred_linear = nn.Linear(hidden, num_tasks)

    def forward(self, x, node_index, edge_index, edge_attr, mask=None):
        """Returns (prediction, last_norm output).  ``mask``: the shared dropout mask (already divided by the keep
        probability) -- drawn here like the reference (model_rev.py:101-102) unless the test supplies it."""
        feats = self.node_features[node_index]
        if self.use_one_hot_encoding:
            feats = torch.cat((feats, self.node_one_hot_encoder(x)), dim=1)
        h = self.node_features_encoder(feats)
        if self.composed_edges:
            # the model file's two lines replaced by one: the embedding is never built
            from deep_gcns_torch_amd.blocks import ComposedEdgeEmbedding
            edge_emb = ComposedEdgeEmbedding(self.edge_encoder, edge_attr, repeat=self.group)
        else:
            edge_emb = torch.cat([self.edge_encoder(edge_attr)] * self.group, dim=-1)
        if mask is None:
            mask = torch.zeros_like(h).bernoulli_(1 - self.dropout).requires_grad_(False) / (1 - self.dropout)
        for layer in range(self.num_layers):
            h = self.gcns[layer](h, edge_index, mask, edge_emb)
        hn = self.last_norm(h)
        out = F.dropout(F.relu(hn), p=self.dropout, training=self.training)
        return self.node_pred_linear(out), hn


class RevGCNModelFile(RevGCN):
    """``RevGCN`` with the forward of the reference's model FILE (examples/ogb_eff/ogbn_proteins/model_rev.py:85-112):
    signature ``(x, node_index, edge_index, edge_attr, epoch=-1)``, the dropout mask drawn inside, the prediction
    returned alone -- what ``deep_gcns_torch_amd.fuse`` sees when the example script imports ``model_rev``."""

    def forward(self, x, node_index, edge_index, edge_attr, epoch=-1):
        node_features_1st = self.node_features[node_index]
        if self.use_one_hot_encoding:
            node_features = torch.cat((node_features_1st, self.node_one_hot_encoder(x)), dim=1)
        else:
            node_features = node_features_1st
        h = self.node_features_encoder(node_features)
        edge_emb = self.edge_encoder(edge_attr)
        edge_emb = torch.cat([edge_emb] * self.group, dim=-1)
        m = torch.zeros_like(h).bernoulli_(1 - self.dropout)
        mask = m.requires_grad_(False) / (1 - self.dropout)
        h = self.gcns[0](h, edge_index, mask, edge_emb)
        for layer in range(1, self.num_layers):
            h = self.gcns[layer](h, edge_index, mask, edge_emb)
        h = F.relu(self.last_norm(h))
        h = F.dropout(h, p=self.dropout, training=self.training)
        return self.node_pred_linear(h)
