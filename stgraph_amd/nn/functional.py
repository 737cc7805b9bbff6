"""Dense neighbours of the Seastar kernels with an MI355X-native weight gradient.

Forward and input gradient stay on rocBLAS (ordinary GEMM shapes).  The WEIGHT gradient is a
tall-skinny contraction over the vertex dimension (K = |V|, output at most a few hundred wide) for
which rocBLAS/hipBLASLt launch 4-16 workgroups; ``kernels.gemm_tn`` splits K over the whole chip on
the fp32 matrix cores.  Selected per call: only when K is large and the output small, otherwise the
stock torch op runs.  ``set_native_weight_grad(False)`` restores torch everywhere.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .. import kernels

_NATIVE_WGRAD = True
MIN_K = 4096          # below this the stock GEMM is launch-bound either way
MAX_MN = 1 << 18      # output elements; larger outputs are ordinary GEMMs


def set_native_weight_grad(enabled: bool) -> None:
    global _NATIVE_WGRAD
    _NATIVE_WGRAD = bool(enabled)


def _use_native(x: torch.Tensor, k: int, m: int, n: int) -> bool:
    return (_NATIVE_WGRAD and x.is_cuda and x.dtype == torch.float32 and k >= MIN_K and m * n <= MAX_MN)


class _MM(torch.autograd.Function):
    """``x @ w`` (reference: ``torch.mm(h, self.weight)``, nn/pytorch/static/gcn_conv.py:158)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return torch.mm(x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.mm(g, w.t())
        if ctx.needs_input_grad[1]:
            gw = kernels.gemm_tn(x, g) if _use_native(x, x.shape[0], x.shape[1], g.shape[1]) else torch.mm(x.t(), g)
        return gx, gw


class _Linear(torch.autograd.Function):
    """``F.linear(x, w, b)`` for 2-D ``x`` (TGCN's gate Linears, nn/pytorch/temporal/tgcn.py:21-47)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.mm(g, w)
        if ctx.needs_input_grad[1]:
            gw = kernels.gemm_tn(g, x) if _use_native(x, x.shape[0], g.shape[1], x.shape[1]) else torch.mm(g.t(), x)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return gx, gw, gb


def mm(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    if x.dim() == 2 and _use_native(x, x.shape[0], x.shape[1], w.shape[1]):
        return _MM.apply(x, w)
    return torch.mm(x, w)


def linear(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None = None) -> torch.Tensor:
    if x.dim() == 2 and _use_native(x, x.shape[0], w.shape[0], w.shape[1]):
        return _Linear.apply(x, w, b)
    return F.linear(x, w, b)
