"""``GCNConv`` -- drop-in for ``stgraph.nn.pytorch.static.gcn_conv.GCNConv``
(reference nn/pytorch/static/gcn_conv.py:78-189).

Same constructor, parameter names (``weight [in, out]`` Xavier-uniform, ``bias
[out]`` zeros), checks and error messages, and the same two vertex functions; the
dense ``h @ W`` stays in torch (rocBLAS), the neighbour aggregation runs in the
fused gfx950 kernel selected by ``stgraph_amd.compiler``.
"""
from __future__ import annotations

import torch
from torch import nn

from ....compiler import STGraph
from ....compiler.backend.pytorch.torch_callback import STGraphBackendTorch
from ....utils.constants import SizeConstants
from ... import functional as SF


class GCNConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, activation=None, bias: bool = True) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.Tensor(in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.bias = None
        self.activation = activation
        self.stgraph = STGraph(STGraphBackendTorch())
        self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    @staticmethod
    def check_norm(graph) -> None:
        """reference gcn_conv.py:151-156"""
        if graph.get_ndata("norm") is None:
            raise KeyError("StaticGraph passed to GCNConv forward pass does not contain 'norm' node data")
        if (len(graph.get_ndata("norm").shape) != SizeConstants.NODE_NORM_SIZE.value or
                graph.get_ndata("norm").shape[1] != 1 or
                graph.get_ndata("norm").shape[0] != graph.get_num_nodes()):
            raise ValueError("Node data 'norm' passed to GCNConv should be of shape (num_nodes, 1)")

    def aggregate(self, graph, h, edge_weight=None):
        """The vertex-centric part of the layer (reference gcn_conv.py:160-182) on an already
        transformed feature matrix ``h``."""
        if edge_weight is None:

            @self.stgraph.compile(gnn_module=self)
            def nb_compute(v):
                return sum([nb.h * nb.norm for nb in v.innbs]) * v.norm

            return nb_compute(g=graph, n_feats={"norm": graph.get_ndata("norm"), "h": h})

        @self.stgraph.compile(gnn_module=self)
        def nb_compute(v):  # noqa: F811
            return sum(
                [
                    nb_edge.src.norm * nb_edge.src.h * nb_edge.edge_weight
                    for nb_edge in v.inedges
                ],
            ) * v.norm

        return nb_compute(
            g=graph,
            n_feats={"norm": graph.get_ndata("norm"), "h": h},
            e_feats={"edge_weight": edge_weight},
        )

    def forward(self, graph, h, edge_weight=None):
        self.check_norm(graph)
        if SF.input_layer_usable(graph, h, self.weight, self.activation):
            # an input that carries no gradient (the dataset's features): aggregate first, so that the backward pass
            # of this layer needs no aggregation at all (functional._InputLayer)
            return SF.input_layer(graph, h, self.weight, self.bias, self.activation, edge_weight)
        h = SF.mm(h, self.weight)            # torch.mm forward; weight gradient on the fp32 matrix cores
        if (self.bias is not None or self.activation is not None) and SF.gcn_layer_tail_usable(graph, h, self.activation):
            # same aggregation kernel as the compiled vertex function below, with `+ bias` and the
            # activation applied at its store instead of in two more passes over [N, F]
            return SF.gcn_layer_tail(graph, h, self.bias, self.activation, edge_weight)
        h = self.aggregate(graph, h, edge_weight)
        if self.bias is not None:
            h = h + self.bias
        if self.activation:
            h = self.activation(h)
        return h
