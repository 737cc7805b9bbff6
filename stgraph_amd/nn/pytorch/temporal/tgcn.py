"""``TGCN`` -- drop-in for ``stgraph.nn.pytorch.temporal.tgcn.TGCN``
(reference nn/pytorch/temporal/tgcn.py:4-55): three GCNConv gates + GRU update.
Module names (``conv_z/r/h``, ``linear_z/r/h``) match for ``state_dict`` exchange.

``fuse_gates`` (default on): the three gates aggregate the SAME input
over the SAME graph, ``A_hat (X W_g) + b_g`` for g in {z, r, h}.  Aggregation is column
independent, so one launch at width 3*out over ``X [W_z | W_r | W_h]`` produces exactly the
columns the three separate launches produce (same per-column summation order), with one
third of the launches and 768-byte instead of 256-byte gathered rows.  The only numerical
difference is rocBLAS choosing its tiling for a [N, in] x [in, 3*out] product instead of three
[in, out] ones (fp32 rounding of the K = in dot products).  ``fuse_gates = False`` runs the
reference's three separate layers.

``fuse_cell`` (default on, needs ``fuse_gates``): bias + clamp, the two ``cat``s, sigmoid / tanh and
the GRU blend run as six fused HIP kernels around the three rocBLAS gate GEMMs inside one autograd
node (``cell.TGCNCellFn``) instead of ~57 torch launches per snapshot.

The gate ``Linear`` layers keep their modules (and parameter names) but are applied through
``stgraph_amd.nn.functional.linear``: same forward, weight gradient by the split-K MFMA kernel.
"""
from __future__ import annotations

import torch

from .... import kernels
from ... import functional as SF
from ..static.gcn_conv import GCNConv
from . import cell


class TGCN(torch.nn.Module):
    fuse_gates = True
    fuse_cell = True     # everything after the aggregation as ONE autograd node (temporal/cell.py)
    fuse_transform = True  # aggregate at the narrow input width, transform on the matrix cores (one kernel)

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.conv_z = GCNConv(self.in_channels, self.out_channels, activation=None)
        self.linear_z = torch.nn.Linear(2 * self.out_channels, self.out_channels)
        self.conv_r = GCNConv(self.in_channels, self.out_channels, activation=None)
        self.linear_r = torch.nn.Linear(2 * self.out_channels, self.out_channels)
        self.conv_h = GCNConv(self.in_channels, self.out_channels, activation=None)
        self.linear_h = torch.nn.Linear(2 * self.out_channels, self.out_channels)

    def _set_hidden_state(self, X, H):
        if H is None:
            H = torch.zeros(X.shape[0], self.out_channels, device=X.device)   # no host copy: capturable
        return H

    def _gate_convs(self, g, X, edge_weight):
        """(conv_z(X), conv_r(X), conv_h(X)), clamped as in tgcn.py:22,30,38."""
        convs = (self.conv_z, self.conv_r, self.conv_h)
        # (reference-compat mode keeps the three launches: defect D1 depends on the launch width)
        if self.fuse_gates and not kernels.reference_compat() and \
                all(type(c) is GCNConv and c.bias is not None for c in convs):
            GCNConv.check_norm(g)
            W = torch.cat([c.weight for c in convs], dim=1)
            b = torch.cat([c.bias for c in convs], dim=0)
            h = self.conv_z.aggregate(g, SF.mm(X, W), edge_weight) + b
            h = torch.clamp(h, min=-1e6, max=1e6)
            return torch.split(h, self.out_channels, dim=1)
        return tuple(torch.clamp(c(g, X, edge_weight=edge_weight), min=-1e6, max=1e6) for c in convs)

    @staticmethod
    def _apply_linear(lin: torch.nn.Linear, x):
        return SF.linear(x, lin.weight, lin.bias)

    def _calculate_update_gate(self, h, H):
        Z = torch.cat((h, H), dim=1)
        Z = self._apply_linear(self.linear_z, Z)
        Z = torch.sigmoid(Z)
        return Z

    def _calculate_reset_gate(self, h, H):
        R = torch.cat((h, H), dim=1)
        R = self._apply_linear(self.linear_r, R)
        R = torch.sigmoid(R)
        return R

    def _calculate_candidate_state(self, h, H, R):
        H_tilde = torch.cat((h, H * R), dim=1)
        H_tilde = self._apply_linear(self.linear_h, H_tilde)
        H_tilde = torch.tanh(H_tilde)
        return H_tilde

    def _calculate_hidden_state(self, Z, H, H_tilde):
        H = Z * H + (1 - Z) * H_tilde
        return H

    def _fused_cell(self, g, X, edge_weight, H):
        """One aggregation launch + ``cell.TGCNCellFn`` (fused row-local stages); same math as below."""
        convs = (self.conv_z, self.conv_r, self.conv_h)
        GCNConv.check_norm(g)
        if self.fuse_transform and SF.agg_transform_usable(g, X, (self.in_channels, 3 * self.out_channels)):
            # whole step as one autograd node: (A_hat X) W on the matrix cores (gather at width in_channels),
            # fused row-local stages, weight gradients deferred to the end of the backward pass
            return cell.TGCNStepFn.apply(
                X, H, g.get_ndata("norm"), edge_weight, g.csr("fwd"), g.csr("bwd"), kernels.rows_by_node_ids(g.graph_type()),
                convs[0].weight, convs[1].weight, convs[2].weight, convs[0].bias, convs[1].bias, convs[2].bias,
                self.linear_z.weight, self.linear_z.bias, self.linear_r.weight, self.linear_r.bias,
                self.linear_h.weight, self.linear_h.bias)
        W = torch.cat([c.weight for c in convs], dim=1)
        b3 = torch.cat([c.bias for c in convs], dim=0)
        a3 = self.conv_z.aggregate(g, SF.mm(X, W), edge_weight)
        return cell.TGCNCellFn.apply(a3, b3, H, self.linear_z.weight, self.linear_z.bias,
                                     self.linear_r.weight, self.linear_r.bias,
                                     self.linear_h.weight, self.linear_h.bias)

    def forward(self, g, X, edge_weight=None, H=None):
        H = self._set_hidden_state(X, H)
        if self.fuse_cell and self.fuse_gates and not kernels.reference_compat() and cell.usable(X, H) and \
                all(type(c) is GCNConv and c.bias is not None for c in (self.conv_z, self.conv_r, self.conv_h)):
            return self._fused_cell(g, X, edge_weight, H)
        hz, hr, hh = self._gate_convs(g, X, edge_weight)
        Z = self._calculate_update_gate(hz, H)
        R = self._calculate_reset_gate(hr, H)
        H_tilde = self._calculate_candidate_state(hh, H, R)
        H = self._calculate_hidden_state(Z, H, H_tilde)
        return H
