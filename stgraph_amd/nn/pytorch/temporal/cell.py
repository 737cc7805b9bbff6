"""One TGCN step as a single autograd node.

Mathematically identical to reference nn/pytorch/temporal/tgcn.py:21-55 (three GCNConv gates: bias
add and clamp to +-1e6; ``cat``/``Linear``/sigmoid for Z and R; ``cat``/``Linear``/tanh for the
candidate; GRU blend); what changes is how many kernels it takes.

``TGCNCellFn``  : everything AFTER the aggregation (input: ``a3 = A_hat (X [Wz|Wr|Wh])``).
``TGCNStepFn``  : the whole step including the aggregation, run as the fused aggregate-then-transform
                  kernel (``kernels.gcn_agg_transform``: gather at width ``in_channels``, the
                  [in, 3*out] weight applied on the matrix cores from LDS).

Per snapshot the eager torch formulation costs ~100 launches (forward + backward); ``TGCNStepFn`` takes
1 fused aggregation + 3 fused row-local kernels + 3 rocBLAS GEMMs forward, and 3 fused kernels +
4 rocBLAS GEMMs + 1 aggregation backward.  The concatenated GEMM operands ([hz|H], [hr|H], [hh|H*R])
are written in place by the fused kernels (no ``cat``).  Weight and bias gradients are not produced per
step at all: they are registered with ``nn.deferred`` and computed once per backward pass, one split-K
MFMA launch per parameter over all timesteps (``functional.set_deferred_weight_grads(False)`` computes
them per step instead and returns them through autograd as usual).
"""
from __future__ import annotations

import torch

from .... import kernels
from ... import deferred
from ... import functional as SF

CLAMP = 1e6      # tgcn.py:22,30,38
_FUSED_FWD = True
_FUSED_BWD = True


def set_fused_backward(enabled: bool) -> None:
    """True (default): the backward chain (GRU / gate / clamp backward and the three input-gradient GEMMs) is ONE
    launch (kernels.tgcn_cell_fused_bwd) when the hidden width is 32 or 64."""
    global _FUSED_BWD
    _FUSED_BWD = bool(enabled)


def set_fused_forward(enabled: bool) -> None:
    """True (default): the forward chain (bias + clamp, three gate GEMMs, sigmoid / tanh, GRU blend) is ONE launch
    (kernels.tgcn_cell_fused_fwd) when the hidden width is 32 or 64.  False: three fused elementwise kernels
    around three rocBLAS GEMMs."""
    global _FUSED_FWD
    _FUSED_FWD = bool(enabled)


def _cell_forward(a3, b3, H, Wz, bz, Wr, br, Wh, bh):
    N, C = H.shape
    dev = H.device
    if _FUSED_FWD and kernels.tgcn_cell_fused_supported(C) and all(
            t.is_contiguous() for t in (Wz, bz, Wr, br, Wh, bh)):
        return kernels.tgcn_cell_fused_fwd(a3, b3, H, Wz, bz, Wr, br, Wh, bh, -CLAMP, CLAMP)
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    CZ, CR, CH = new(N, 2 * C), new(N, 2 * C), new(N, 2 * C)
    kernels.tgcn_cell_call("prep_fwd", (a3, b3, H, CZ, CR, CH), N, C, -CLAMP, CLAMP)
    zl = kernels.linear_fwd(CZ, Wz, bz)
    rl = kernels.linear_fwd(CR, Wr, br)
    Z, R = new(N, C), new(N, C)
    kernels.tgcn_cell_call("gates_fwd", (zl, rl, H, Z, R, CH), N, C)
    hl = kernels.linear_fwd(CH, Wh, bh)
    Ht, Hn = new(N, C), new(N, C)
    kernels.tgcn_cell_call("update_fwd", (hl, Z, H, Ht, Hn), N, C)
    return Hn, (CZ, CR, CH, Z, R, Ht)


_FUSED_DX = True


def set_fused_dx(enabled: bool) -> None:
    """True (default): where supported (hidden 32 / 64, 32 input features) the backward launch also forms
    ``da3 @ Wcat.T`` (kernels.tgcn_cell_fused_bwd with ``Wcat``) instead of a separate GEMM."""
    global _FUSED_DX
    _FUSED_DX = bool(enabled)


def _cell_backward(dHn, a3, b3, H, Wz, Wr, Wh, CZ, CR, CH, Z, R, Ht, Wcat=None):
    """Returns da3, dH and the three (d_preactivation, operand) pairs of the gate Linears; with ``Wcat`` (and the fused
    path able to take it) a fourth element ``dx = da3 @ Wcat.T``, else None there."""
    N, C = H.shape
    dev = H.device
    if _FUSED_BWD and kernels.tgcn_cell_fused_supported(C) and all(t.is_contiguous() for t in (Wz, Wr, Wh)):
        if Wcat is not None and _FUSED_DX and kernels.tgcn_cell_fused_bwd_dx_supported(C, Wcat.shape[0]):
            da3, dH, dzl, drl, dhl, dx = kernels.tgcn_cell_fused_bwd(dHn, Z, H, Ht, R, a3, b3, Wz, Wr, Wh, -CLAMP, CLAMP,
                                                                     Wcat=Wcat)
            return da3, dH, ((dzl, CZ), (drl, CR), (dhl, CH)), dx
        da3, dH, dzl, drl, dhl = kernels.tgcn_cell_fused_bwd(dHn, Z, H, Ht, R, a3, b3, Wz, Wr, Wh, -CLAMP, CLAMP)
        return (da3, dH, ((dzl, CZ), (drl, CR), (dhl, CH))) if Wcat is None else \
            (da3, dH, ((dzl, CZ), (drl, CR), (dhl, CH)), None)
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    dhl, dzl, dH = new(N, C), new(N, C), new(N, C)
    kernels.tgcn_cell_call("update_bwd", (dHn, Z, H, Ht, dhl, dzl, dH), N, C)
    dCH = kernels.matmul(dhl, Wh)                         # [N, 2C] = grad of [hh | H*R]
    drl = new(N, C)
    kernels.tgcn_cell_call("gates_bwd", (dCH, R, H, drl, dH), N, C)
    dCZ = kernels.matmul(dzl, Wz)
    dCR = kernels.matmul(drl, Wr)
    da3 = new(N, 3 * C)
    kernels.tgcn_cell_call("prep_bwd", (dCZ, dCR, dCH, a3, b3, da3, dH), N, C, -CLAMP, CLAMP)
    if Wcat is not None:
        return da3, dH, ((dzl, CZ), (drl, CR), (dhl, CH)), None
    return da3, dH, ((dzl, CZ), (drl, CR), (dhl, CH))


def _linear_grads(pairs, params, defer: bool):
    """Weight/bias gradients of the gate Linears: dW = dpre^T operand, db = colsum(dpre).
    ``params`` = ((W, b), ...) leaf parameters.  Deferred: registered, returns Nones."""
    out = []
    for (dpre, operand), (W, b) in zip(pairs, params):
        if defer and W.is_leaf and b.is_leaf:
            deferred.current().add(("lin", id(W)), dpre, operand,
                                   sink=lambda dW, W=W: deferred.add_to_grad(W, dW),
                                   colsum_sink=lambda db, b=b: deferred.add_to_grad(b, db))
            out += [None, None]
        else:
            dW, db = kernels.gemm_tn(dpre, operand, colsum=True)
            out += [dW, db]
    return out


class TGCNCellFn(torch.autograd.Function):
    """Hn = cell(a3, b3, H; gate Linears) -- the row-local part only."""

    @staticmethod
    def forward(ctx, a3, b3, H, Wz, bz, Wr, br, Wh, bh):
        a3, b3, H = a3.contiguous(), b3.contiguous(), H.contiguous()
        Hn, extra = _cell_forward(a3, b3, H, Wz, bz, Wr, br, Wh, bh)
        ctx.save_for_backward(a3, b3, H, Wz, Wr, Wh, *extra)
        ctx.params = ((Wz, bz), (Wr, br), (Wh, bh))
        return Hn

    @staticmethod
    def backward(ctx, dHn):
        a3, b3, H, Wz, Wr, Wh, CZ, CR, CH, Z, R, Ht = ctx.saved_tensors
        da3, dH, pairs = _cell_backward(dHn.contiguous(), a3, b3, H, Wz, Wr, Wh, CZ, CR, CH, Z, R, Ht)
        g = _linear_grads(pairs, ctx.params, SF.deferred_weight_grads())
        db3 = da3.sum(0) if ctx.needs_input_grad[1] else None
        return (da3, db3, dH, g[0], g[1], g[2], g[3], g[4], g[5])


class TGCNStepFn(torch.autograd.Function):
    """Hn = TGCN(graph, x, H): aggregation (fused aggregate-then-transform) + cell, one node."""

    @staticmethod
    def forward(ctx, x, H, norm, ew, fwd_csr, bwd_csr, use_nid,
                Wcz, Wcr, Wch, bcz, bcr, bch, Wz, bz, Wr, br, Wh, bh):
        x, H = x.contiguous(), H.contiguous()
        Wcat, b3 = _concatenated((Wcz, Wcr, Wch), (bcz, bcr, bch))
        a3, P = kernels.gcn_agg_transform(x, Wcat, norm, norm, fwd_csr, ew=ew, use_node_ids=use_nid)
        Hn, extra = _cell_forward(a3, b3, H, Wz, bz, Wr, br, Wh, bh)
        ctx.save_for_backward(a3, b3, H, Wz, Wr, Wh, *extra, P, Wcat, norm,
                              ew if ew is not None else norm.new_empty(0))
        ctx.has_ew = ew is not None
        ctx.bwd_csr, ctx.use_nid = bwd_csr, use_nid
        ctx.params = ((Wz, bz), (Wr, br), (Wh, bh))
        ctx.conv_params = ((Wcz, Wcr, Wch), (bcz, bcr, bch))
        return Hn

    @staticmethod
    def backward(ctx, dHn):
        a3, b3, H, Wz, Wr, Wh, CZ, CR, CH, Z, R, Ht, P, Wcat, norm, ew = ctx.saved_tensors
        ew = ew if ctx.has_ew else None
        defer = SF.deferred_weight_grads()
        want_dx = ctx.needs_input_grad[0]
        if want_dx:
            da3, dH, pairs, z = _cell_backward(dHn.contiguous(), a3, b3, H, Wz, Wr, Wh, CZ, CR, CH, Z, R, Ht, Wcat=Wcat)
        else:
            (da3, dH, pairs), z = _cell_backward(dHn.contiguous(), a3, b3, H, Wz, Wr, Wh, CZ, CR, CH, Z, R, Ht), None
        g = _linear_grads(pairs, ctx.params, defer)
        dx = None
        if want_dx:
            if z is None:
                z = kernels.matmul_t(da3, Wcat)                       # d(A_hat x) = da3 Wcat^T      [N, in]
            dx = kernels.gcn_agg(z, norm, norm, ctx.bwd_csr, ew=ew, use_node_ids=ctx.use_nid)
        # GCN weights/biases: dWcat^T = da3^T P  ([3C, in]),  db3 = colsum(da3)
        Ws, bs = ctx.conv_params
        C = H.shape[1]

        def sink_w(dWt, Ws=Ws):
            for i, W in enumerate(Ws):
                deferred.add_to_grad(W, dWt[i * C:(i + 1) * C].t())

        def sink_b(db, bs=bs):
            for i, b in enumerate(bs):
                deferred.add_to_grad(b, db[i * C:(i + 1) * C])

        if defer and all(t.is_leaf for t in (*Ws, *bs)):
            deferred.current().add(("conv", id(Ws[0])), da3, P, sink=sink_w, colsum_sink=sink_b)
            gw = [None] * 6
        else:
            dWt, db = kernels.gemm_tn(da3, P, colsum=True)
            gw = [dWt[i * C:(i + 1) * C].t() for i in range(3)] + [db[i * C:(i + 1) * C] for i in range(3)]
        return (dx, dH, None, None, None, None, None, *gw, *g)


_CAT_CACHE = {}
_CAPTURE = [False, 0]          # was the previous call made under stream capture; number of captures seen


def _concatenated(ws, bs):
    """``cat(ws, dim=1)``, ``cat(bs)`` of the three GCN gates, kept until one of them changes: a BPTT window applies
    the same parameters at every step, so the two concatenations run once per optimizer step (or once per captured
    graph replay -- the cache is filled inside the capture, by the first step of the window) instead of per step.
    Keyed on storage address and in-place version counter of every part, and on the stream capture the call is made
    under: a captured graph must contain its own concatenation (replays see new parameter values), so nothing
    computed outside a capture -- or in an earlier one -- is reused inside it."""
    capturing = ws[0].is_cuda and torch.cuda.is_current_stream_capturing()
    if capturing and not _CAPTURE[0]:
        _CAPTURE[1] += 1
    _CAPTURE[0] = capturing
    key = tuple(id(t) for t in (*ws, *bs))
    stamp = (_CAPTURE[1] if capturing else 0,) + tuple((t.data_ptr(), t._version) for t in (*ws, *bs))
    hit = _CAT_CACHE.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1], hit[2]
    if len(_CAT_CACHE) > 64:
        _CAT_CACHE.clear()
    with torch.no_grad():
        wcat, b3 = torch.cat(list(ws), dim=1), torch.cat(list(bs), dim=0)
    _CAT_CACHE[key] = (stamp, wcat, b3)
    return wcat, b3


def usable(a3: torch.Tensor, H: torch.Tensor) -> bool:
    return a3.is_cuda and a3.dtype == torch.float32 and H.shape[1] % 4 == 0 and H.shape[0] > 0
