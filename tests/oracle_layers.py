"""CPU stand-ins for the aggregation kernels, backed by the oracle (TEST CODE ONLY).

Lets the layer stack (GCNConv/TGCN dense parts, the training loops, the data-parallel
path under gloo) run in the GPU-less container, and serves as the CPU reference model
in the GPU parity tests.  Pinned against the reference-generated fixtures in
tests/test_oracle_models.py.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from oracle import stg_oracle as orc


class OracleGraphView:
    """Minimal graph object for the oracle layers: fwd/bwd OracleCSR + norm node data."""

    def __init__(self, src, dst, num_nodes, graph_type="csr_unsorted"):
        self.g = orc.build_graph(src, dst, num_nodes)
        self._ndata = {}
        self._type = graph_type

    def get_num_nodes(self):
        return self.g.num_nodes

    def get_num_edges(self):
        return self.g.num_edges

    def get_ndata(self, k):
        return self._ndata.get(k)

    def set_ndata(self, k, v):
        self._ndata[k] = v

    def graph_type(self):
        return self._type

    def in_degrees(self):
        return self.g.in_degrees()


class _Agg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, norm, ew, graph, f_active):
        ctx.graph, ctx.f_active = graph, f_active
        ctx.norm = norm.detach().numpy()
        ctx.ew = None if ew is None else ew.detach().numpy()
        nid = graph.graph_type() == "csr"
        out = orc.gcn_agg(x.detach().numpy(), ctx.norm, ctx.norm, graph.g.fwd, ew=ctx.ew, use_node_ids=nid,
                          f_active=f_active)
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, g):
        nid = ctx.graph.graph_type() == "csr"
        gx = orc.gcn_agg(g.contiguous().numpy(), ctx.norm, ctx.norm, ctx.graph.g.bwd, ew=ctx.ew, use_node_ids=nid,
                         f_active=ctx.f_active)
        return torch.from_numpy(gx), None, None, None, None


class OracleGCNConv(nn.Module):
    """Same parameters / forward as GCNConv (gcn_conv.py:78-189), aggregation by the oracle."""

    ref_compat = False

    def __init__(self, in_channels, out_channels, activation=None, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.Tensor(in_channels, out_channels))
        self.bias = nn.Parameter(torch.Tensor(out_channels)) if bias else None
        self.activation = activation
        nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, graph, h, edge_weight=None):
        h = torch.mm(h, self.weight)
        F = h.shape[1]
        fa = orc.ref_active_columns(F) if OracleGCNConv.ref_compat else F
        h = _Agg.apply(h, graph.get_ndata("norm"), edge_weight, graph, fa)
        if self.bias is not None:
            h = h + self.bias
        if self.activation:
            h = self.activation(h)
        return h


def make_oracle_tgcn():
    from stgraph_amd.nn.pytorch.temporal.tgcn import TGCN

    class OracleTGCN(TGCN):
        def __init__(self, in_channels, out_channels):
            super().__init__(in_channels, out_channels)
            self.conv_z = OracleGCNConv(in_channels, out_channels)
            self.conv_r = OracleGCNConv(in_channels, out_channels)
            self.conv_h = OracleGCNConv(in_channels, out_channels)

    return OracleTGCN


def gcn_norm_tensor(in_degrees) -> torch.Tensor:
    deg = torch.from_numpy(np.asarray(in_degrees)).float()
    norm = torch.pow(deg, -0.5)
    norm[torch.isinf(norm)] = 0
    return norm.unsqueeze(1)
