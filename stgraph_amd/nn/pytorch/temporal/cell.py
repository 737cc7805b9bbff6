"""Fused row-local part of one TGCN step as a single autograd node.

Input: the aggregated gate pre-activations ``a3 = A_hat (X [Wz|Wr|Wh])`` ([N, 3C], output of the
fused gcn_agg launch), the concatenated GCN biases, the previous hidden state and the three gate
``Linear`` layers.  Output: the new hidden state.  Mathematically identical to
reference nn/pytorch/temporal/tgcn.py:21-55 (bias add, clamp to +-1e6, ``cat``/``Linear``/sigmoid
for Z and R, ``cat``/``Linear``/tanh for the candidate, GRU blend).

Per snapshot the eager formulation costs ~57 elementwise/cat/fill launches (forward + backward) plus
autograd bookkeeping; here: 3 fused forward kernels + 3 rocBLAS GEMMs, and in backward 3 fused kernels
+ 3 rocBLAS GEMMs (input gradients) + 3 split-K MFMA launches that produce each gate's weight AND
bias gradient together (stg_gemm_tn_colsum_f32).  The concatenated GEMM operands ([hz|H], [hr|H],
[hh|H*R]) are written in place by the fused kernels, so no ``cat`` exists.
"""
from __future__ import annotations

import torch

from .... import kernels

CLAMP = 1e6      # tgcn.py:22,30,38


class TGCNCellFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a3, b3, H, Wz, bz, Wr, br, Wh, bh):
        a3, H = a3.contiguous(), H.contiguous()
        b3 = b3.contiguous()
        N, C = H.shape
        dev = H.device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        CZ, CR, CH = new(N, 2 * C), new(N, 2 * C), new(N, 2 * C)
        kernels.tgcn_cell_call("prep_fwd", (a3, b3, H, CZ, CR, CH), N, C, -CLAMP, CLAMP)
        zl = torch.addmm(bz, CZ, Wz.t())
        rl = torch.addmm(br, CR, Wr.t())
        Z, R = new(N, C), new(N, C)
        kernels.tgcn_cell_call("gates_fwd", (zl, rl, H, Z, R, CH), N, C)
        hl = torch.addmm(bh, CH, Wh.t())
        Ht, Hn = new(N, C), new(N, C)
        kernels.tgcn_cell_call("update_fwd", (hl, Z, H, Ht, Hn), N, C)
        ctx.save_for_backward(a3, b3, H, Wz, Wr, Wh, CZ, CR, CH, Z, R, Ht)
        return Hn

    @staticmethod
    def backward(ctx, dHn):
        a3, b3, H, Wz, Wr, Wh, CZ, CR, CH, Z, R, Ht = ctx.saved_tensors
        dHn = dHn.contiguous()
        N, C = H.shape
        dev = H.device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        dhl, dzl, dH = new(N, C), new(N, C), new(N, C)
        kernels.tgcn_cell_call("update_bwd", (dHn, Z, H, Ht, dhl, dzl, dH), N, C)
        dCH = torch.mm(dhl, Wh)                               # [N, 2C] = grad of [hh | H*R]
        drl = new(N, C)
        kernels.tgcn_cell_call("gates_bwd", (dCH, R, H, drl, dH), N, C)
        dCZ = torch.mm(dzl, Wz)
        dCR = torch.mm(drl, Wr)
        da3 = new(N, 3 * C)
        kernels.tgcn_cell_call("prep_bwd", (dCZ, dCR, dCH, a3, b3, da3, dH), N, C, -CLAMP, CLAMP)
        # weight + bias gradients of the three gate Linears: dW = dpre^T [h|H], db = colsum(dpre)
        dWz, dbz = kernels.gemm_tn(dzl, CZ, colsum=True)
        dWr, dbr = kernels.gemm_tn(drl, CR, colsum=True)
        dWh, dbh = kernels.gemm_tn(dhl, CH, colsum=True)
        db3 = da3.sum(0) if ctx.needs_input_grad[1] else None
        return da3, db3, dH, dWz, dbz, dWr, dbr, dWh, dbh


def usable(a3: torch.Tensor, H: torch.Tensor) -> bool:
    return a3.is_cuda and a3.dtype == torch.float32 and H.shape[1] % 4 == 0 and H.shape[0] > 0
