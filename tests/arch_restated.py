"""Test helper: the reference's example ARCHITECTURES restated on top of the `gcn_lib` registered in
sys.modules, with the reference's attribute names so its state_dicts load unchanged.  Needed because the
reference's own example files cannot travel to the GPU box; tests/test_dropin.py separately proves (in
the build container) that the real files build the same state_dict on this package's gcn_lib.
  DenseDeepGCN  examples/sem_seg_dense/architecture.py:7-56
  DeeperGCN     examples/ogb/ogbn_arxiv/model.py:10-140 ('res+' path incl. gradient checkpointing)
  DeepGCN       examples/ppi/architecture.py:6-55
"""
import torch
import torch.nn.functional as F
from torch.nn import Sequential as Seq
from torch.utils.checkpoint import checkpoint


class DenseDeepGCN(torch.nn.Module):
    def __init__(self, n_blocks=4, channels=32, k=8, in_channels=9, n_classes=13, act="relu", norm="batch"):
        super().__init__()
        from gcn_lib.dense import BasicConv, DenseDilatedKnnGraph, GraphConv2d, ResDynBlock2d
        self.n_blocks = n_blocks
        self.knn = DenseDilatedKnnGraph(k, 1, False, 0.0)
        self.head = GraphConv2d(in_channels, channels, "edge", act, norm, True)
        self.backbone = Seq(*[ResDynBlock2d(channels, k, 1 + i, "edge", act, norm, True, False, 0.0)
                              for i in range(n_blocks - 1)])
        fusion_dims = channels * n_blocks
        self.fusion_block = BasicConv([fusion_dims, 1024], act, norm, True)
        self.prediction = Seq(BasicConv([fusion_dims + 1024, 512], act, norm, True),
                              BasicConv([512, 256], act, norm, True), torch.nn.Dropout(p=0.0),
                              BasicConv([256, n_classes], None, None, True))

    def forward(self, inputs):
        feats = [self.head(inputs, self.knn(inputs[:, 0:3]))]
        for i in range(self.n_blocks - 1):
            feats.append(self.backbone[i](feats[-1]))
        feats = torch.cat(feats, dim=1)
        fusion = torch.max_pool2d(self.fusion_block(feats), kernel_size=[feats.shape[2], feats.shape[3]])
        fusion = torch.repeat_interleave(fusion, repeats=feats.shape[2], dim=2)
        return self.prediction(torch.cat((fusion, feats), dim=1)).squeeze(-1)


class DeeperGCN(torch.nn.Module):
    """'res+' DeeperGCN (ogbn_arxiv/model.py:10-140).  ``dropout``: the reference's ``F.dropout`` after every norm + ReLU
    (args default 0.5; 0.0 here keeps the parity tests deterministic).  ``fused_layers``: the layer loop through
    ``deep_gcns_torch_amd.blocks.res_plus_layer`` (same arithmetic, same parameters and state_dict) -- what the example's
    model.py looks like after the change INTEGRATION.md shows."""

    def __init__(self, num_layers=8, in_channels=32, hidden=64, num_tasks=10, aggr="softmax_sg", t=0.1,
                 norm="batch", mlp_layers=1, dropout=0.0, fused_layers=False, checkpoint="reference", **gen_kw):
        super().__init__()
        from gcn_lib.sparse.torch_nn import norm_layer
        from gcn_lib.sparse.torch_vertex import GENConv
        self.num_layers = num_layers
        self.dropout = dropout
        self.block = "res+"                      # ogbn_arxiv/model.py:16 (args.block; the README's command uses res+)
        self.fused_layers = fused_layers
        # the reference checkpoints the convolutions of deep softmax / power stacks because torch_scatter keeps several
        # (E, C) temporaries per layer alive; "never" = what a user of this package can do instead: nothing of size
        # (E, C) exists here, a layer keeps two (N, C) arrays (arxiv, 28 layers: 4.9 GB), and the recomputation of
        # aggregation + GEMM disappears
        # "reference_full": res_plus_layer recomputes the aggregation too (what torch.utils.checkpoint around the
        # reference's GENConv does); "reference": it keeps the aggregation's (N, C) outputs and recomputes the rest
        if checkpoint not in ("reference", "reference_full", "never"):
            raise ValueError(checkpoint)
        self.checkpoint_grad = checkpoint != "never" and aggr in ("softmax_sg", "softmax", "power") and num_layers > 7
        self.checkpoint_mode = "full" if checkpoint == "reference_full" else True
        self.ckp_k = num_layers // 2
        self.gcns = torch.nn.ModuleList()
        self.norms = torch.nn.ModuleList()
        self.node_features_encoder = torch.nn.Linear(in_channels, hidden)
        self.node_pred_linear = torch.nn.Linear(hidden, num_tasks)
        for _ in range(num_layers):
            self.gcns.append(GENConv(hidden, hidden, aggr=aggr, t=t, norm=norm, mlp_layers=mlp_layers, **gen_kw))
            self.norms.append(norm_layer(norm, hidden))

    def forward(self, x, edge_index):
        h = self.node_features_encoder(x)
        if self.fused_layers:
            from deep_gcns_torch_amd import blocks, node_ops
            h, stats = self.gcns[0](h, edge_index, want_stats=True)
            for layer in range(1, self.num_layers):
                h, stats = blocks.res_plus_layer(self.norms[layer - 1], self.gcns[layer], h, edge_index, p=self.dropout,
                                                 training=self.training, stats=stats,
                                                 use_checkpoint=(self.checkpoint_grad and layer % self.ckp_k != 0)
                                                 and self.checkpoint_mode)
            h = node_ops.pre_activation(self.norms[self.num_layers - 1], h, p=self.dropout, training=self.training,
                                        stats=stats)
            return torch.log_softmax(self.node_pred_linear(h), dim=-1)
        h = self.gcns[0](h, edge_index)
        for layer in range(1, self.num_layers):
            h2 = F.dropout(F.relu(self.norms[layer - 1](h)), p=self.dropout, training=self.training)
            if self.checkpoint_grad and layer % self.ckp_k != 0:
                h = checkpoint(self.gcns[layer], h2, edge_index, use_reentrant=True) + h
            else:
                h = self.gcns[layer](h2, edge_index) + h
        h = F.dropout(F.relu(self.norms[self.num_layers - 1](h)), p=self.dropout, training=self.training)
        return torch.log_softmax(self.node_pred_linear(h), dim=-1)


class DeepGCN(torch.nn.Module):
    def __init__(self, conv="mr", n_blocks=3, channels=64, in_channels=50, n_classes=121, act="relu", norm="batch"):
        super().__init__()
        from gcn_lib.sparse import MLP, GraphConv, MultiSeq, ResGraphBlock
        self.n_blocks = n_blocks
        self.head = GraphConv(in_channels, channels, conv, act, norm, True, 1)
        self.backbone = MultiSeq(*[ResGraphBlock(channels, conv, act, norm, True, 1, 1) for _ in range(n_blocks - 1)])
        fusion_dims = channels * n_blocks
        self.fusion_block = MLP([fusion_dims, 1024], act, None, True)
        self.prediction = Seq(MLP([1 + fusion_dims, 512], act, norm, True), torch.nn.Dropout(p=0.0),
                              MLP([512, 256], act, norm, True), torch.nn.Dropout(p=0.0),
                              MLP([256, n_classes], None, None, True))

    def forward(self, x, edge_index):
        feats = [self.head(x, edge_index)]
        for i in range(self.n_blocks - 1):
            feats.append(self.backbone[i](feats[-1], edge_index)[0])
        feats = torch.cat(feats, 1)
        fusion, _ = torch.max(self.fusion_block(feats), 1, keepdim=True)
        return self.prediction(torch.cat((feats, fusion), 1))
