"""``GATConv`` -- drop-in for ``stgraph.nn.pytorch.static.gat_conv.GATConv``
(reference nn/pytorch/static/gat_conv.py:7-61).

Same constructor, parameters (``fc.weight``, ``attn_l``, ``attn_r``; Xavier-normal
with ReLU gain) and the same vertex function, including its quirks: ``max(embs)``
is Python's builtin over a one-element list, so the "softmax shift" is ``emb - emb``
and attention degenerates to a uniform mean, and
``attn_drop`` is constructed but never applied (D11).
"""
from __future__ import annotations

import torch
from torch import nn

from ....compiler import STGraph
from ....compiler.backend.pytorch.torch_callback import STGraphBackendTorch
from ... import functional as SF


class GATConv(nn.Module):
    def __init__(self, in_feats, out_feats, num_heads, feat_drop=0., attn_drop=0.,
                 negative_slope=0.2, activation=None):
        super().__init__()
        self._num_heads = num_heads
        self._in_feats = in_feats
        self._out_feats = out_feats
        self.fc = nn.Linear(self._in_feats, out_feats * num_heads, bias=False)
        self.attn_l = nn.Parameter(torch.FloatTensor(size=(num_heads, out_feats)))
        self.attn_r = nn.Parameter(torch.FloatTensor(size=(num_heads, out_feats)))
        self.feat_drop = nn.Dropout(feat_drop)
        self.attn_drop = nn.Dropout(attn_drop)
        self.leaky_relu = nn.LeakyReLU(negative_slope)
        self.negative_slope = negative_slope

        self.activation = activation
        self.stgraph = STGraph(STGraphBackendTorch())
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_normal_(self.fc.weight, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        nn.init.xavier_normal_(self.attn_r, gain=gain)

    def forward(self, graph, feat):
        h_dst = h_src = self.feat_drop(feat)  # noqa: F841
        if SF.gat_fc_layer_usable(graph, h_src, self.fc, self._num_heads, self._out_feats):
            # the fc GEMM with the attention projections in its epilogue, the GAT units, and all of it backward, as
            # one autograd node (stg_gat_fc_fwd)
            elu = SF.is_elu(self.activation)        # ... and F.elu in the same node (stg_gat_fc_out / the backward prepass)
            rst = SF.gat_fc_layer(graph, h_src, self.fc, self.attn_l, self.attn_r, self.negative_slope,
                                  self._num_heads, self._out_feats, elu)
            return self.activation(rst) if self.activation and not elu else rst
        # self.fc through functional.linear: same Linear, wide outputs in 128-column slices of the row GEMM (rocBLAS'
        # tile choice for [N, in] x [in, H*D] costs 3x its M = 128 launches), weight gradient on the split-K MFMA kernel
        feat_src = feat_dst = SF.linear(h_src, self.fc.weight, self.fc.bias).view(-1, self._num_heads, self._out_feats)
        if SF.gat_layer_usable(graph, feat_src):
            # static graphs: projections, the three GAT units and the projection gradients as one autograd node
            # (the same kernels the compiled vertex function below dispatches to, minus ~15 torch passes over [N,H,D])
            rst = SF.gat_layer(graph, feat_src, self.attn_l, self.attn_r, self.negative_slope)
            return self.activation(rst) if self.activation else rst
        el = (feat_src * self.attn_l).sum(dim=-1).unsqueeze(-1)
        er = (feat_dst * self.attn_r).sum(dim=-1).unsqueeze(-1)

        @self.stgraph.compile(gnn_module=self)
        def nb_forward(v):
            embs = [nb.el + v.er for nb in v.innbs]
            coeff = [torch.exp(self.leaky_relu(emb - max(embs))) for emb in embs]
            s = sum(coeff)
            alpha = [c / s for c in coeff]
            feat_src = [nb.feat_src for nb in v.innbs]
            return sum([alpha[i] * feat_src[i] for i in range(len(feat_src))])

        rst = nb_forward(g=graph, n_feats={"el": el, "er": er, "feat_src": feat_src})

        if self.activation:
            rst = self.activation(rst)
        return rst
