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
from . import deferred

_NATIVE_WGRAD = True
_DEFER = True
MIN_K = 4096          # below this the stock GEMM is launch-bound either way
MAX_MN = 1 << 18      # output elements; larger outputs are ordinary GEMMs


def set_native_weight_grad(enabled: bool) -> None:
    global _NATIVE_WGRAD
    _NATIVE_WGRAD = bool(enabled)


def set_deferred_weight_grads(enabled: bool) -> None:
    """True (default): weight/bias gradients of this package's dense nodes are accumulated once per
    backward pass with one launch per parameter (see ``nn/deferred.py``).  False: computed per call and
    returned through autograd."""
    global _DEFER
    _DEFER = bool(enabled)


def deferred_weight_grads() -> bool:
    return _DEFER and _NATIVE_WGRAD


def _use_native(x: torch.Tensor, k: int, m: int, n: int) -> bool:
    return (_NATIVE_WGRAD and x.is_cuda and x.dtype == torch.float32 and k >= MIN_K and m * n <= MAX_MN)


LT_MIN_ROWS = 500_000
_ZERO_BIAS = {}


def _mm(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """``x @ w`` from the BLAS library.  Tall fp32 products go through ``addmm`` with a (cached, zero) 1-D bias: that
    form is dispatched to hipBLASLt, whose kernel for [1M, 128] x [128, 128] takes 0.37 ms against rocBLAS' 0.42
    (``torch.mm``); below ~ 500 K rows the two are equal.  Same values (tested)."""
    if w.dim() == 2 and w.is_contiguous() and kernels.rowgemm16_usable(x, w.shape[0], w.shape[1]) and w.data_ptr() % 16 == 0:
        return kernels.rowgemm(x, w, None, trans_w=False)
    if x.is_cuda and x.dtype == torch.float32 and x.shape[0] >= LT_MIN_ROWS and w.is_contiguous():
        key = (w.shape[1], x.device)
        z = _ZERO_BIAS.get(key)
        if z is None:
            if torch.cuda.is_current_stream_capturing():      # not from a graph's private pool
                return torch.mm(x, w)
            z = _ZERO_BIAS[key] = torch.zeros(w.shape[1], dtype=torch.float32, device=x.device)
        return torch.addmm(z, x, w)
    return torch.mm(x, w)


class _MM(torch.autograd.Function):
    """``x @ w`` (reference: ``torch.mm(h, self.weight)``, nn/pytorch/static/gcn_conv.py:158)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.w = w
        # x = relu(...) of _InputLayer with its sign pattern as bits: the backward multiplies g w^T by it in the same launch
        ctx.relu_bits = _tagged(x, "_stg_relu_bits")
        # x = the ReLU output of a GCNConv tail (_GcnLayerTail): a small layer's backward masks its input gradient and sums its
        # columns in the launch that forms it
        ctx.relu_out = _tagged(x, "_stg_relu_out") is not None
        return _mm(x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and ctx.relu_bits is None and kernels.mm_bwd_small_usable(g, x, w):
            # a layer one workgroup holds (Cora's 16 -> 7): both gradients from ONE launch -- launch count is what a small graph's
            # epoch costs.  Below a ReLU layer (and while nobody can see x's own gradient: see the masked product further down) the
            # same launch applies that ReLU's mask and leaves its bias gradient on the tensor.
            observed = x.retains_grad or bool(getattr(x, "_backward_hooks", None))
            if ctx.relu_out and not observed:
                gx, gw, cs = kernels.mm_bwd_small(g, x, w, relu_input=True)
                _tag(gx, "_stg_relu_masked_out", x)
                gx._stg_colsum = (cs, gx._version, gx.data_ptr())
                return gx, gw
            return kernels.mm_bwd_small(g, x, w)
        if ctx.needs_input_grad[0]:
            if w.is_contiguous() and kernels.rowgemm16_usable(g, w.shape[1], w.shape[0]) and w.data_ptr() % 16 == 0:
                g = g.contiguous()
                bits = ctx.relu_bits
                # the masked product is the gradient of the ReLU's PRE-activation, not of x: taken only while nobody can see x's
                # own gradient (x.retain_grad() / x.register_hook(), looked up NOW -- the saved input is the caller's tensor).
                # torch.autograd.grad(loss, x) cannot be seen from here: kernels.set_relu_bits(False) for that
                observed = x.retains_grad or bool(getattr(x, "_backward_hooks", None))
                if bits is not None and not observed and kernels.rowgemm_bits_usable(g, w.shape[1], w.shape[0]):
                    # the ReLU below masks this gradient anyway (and again whatever autograd adds to it: masking twice is
                    # masking once); done here it costs 16 bytes per row instead of a pass over the ReLU's output
                    gx = kernels.rowgemm_masked_t(g, w, bits)
                    _tag(gx, "_stg_relu_masked", bits)
                else:
                    gx = kernels.rowgemm(g, w, None, trans_w=True)           # g @ w.T with w read in place ([in][out] = [M][K])
            else:
                gx = _mm(g, w.t().contiguous()) if g.shape[0] >= LT_MIN_ROWS else torch.mm(g, w.t())
        if ctx.needs_input_grad[1] and kernels.gemm_tn_small_usable(x, g) and not _use_native(x, x.shape[0], x.shape[1], g.shape[1]):
            gw = kernels.gemm_tn_small(x, g)                      # a small graph: one launch (the library GEMM reduces K on a few workgroups)
        elif ctx.needs_input_grad[1]:
            native = _use_native(x, x.shape[0], x.shape[1], g.shape[1])
            if native and deferred_weight_grads() and ctx.w.is_leaf:
                W = ctx.w
                deferred.current().add(("mm", id(W)), x, g.contiguous(), sink=lambda d, W=W: deferred.add_to_grad(W, d))
            else:
                gw = kernels.gemm_tn(x, g) if native else torch.mm(x.t(), g)
        return gx, gw


class _Linear(torch.autograd.Function):
    """``F.linear(x, w, b)`` for 2-D ``x`` (TGCN's gate Linears, nn/pytorch/temporal/tgcn.py:21-47)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.w, ctx.b = w, b
        return kernels.linear_fwd(x, w, b) if x.is_cuda else F.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = kernels.matmul(g, w) if g.is_cuda else torch.mm(g, w)
        want_w = ctx.needs_input_grad[1]
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        native = _use_native(x, x.shape[0], g.shape[1], x.shape[1])
        W, b = ctx.w, ctx.b
        if native and want_w and deferred_weight_grads() and W.is_leaf and (not want_b or b.is_leaf):
            deferred.current().add(("linear", id(W)), g.contiguous(), x,
                                   sink=lambda d, W=W: deferred.add_to_grad(W, d),
                                   colsum_sink=(lambda d, b=b: deferred.add_to_grad(b, d)) if want_b else None)
            return gx, None, None
        if want_w:
            if native and want_b:
                gw, gb = kernels.gemm_tn(g, x, colsum=True)
                return gx, gw, gb
            gw = kernels.gemm_tn(g, x) if native else torch.mm(g.t(), x)
        if want_b:
            gb = g.sum(0)
        return gx, gw, gb


class _TgcnHead(torch.autograd.Function):
    """(y, y_out, loss) = head(h): ``y = relu(h) W1^T + b1``, ``y_out = y W2^T + b2``, ``loss = mean((y_out - t)^2)``
    -- the model head and per-timestep loss of the static-temporal TGCN harness
    (benchmarking/static-temporal-tgcn/seastar/model.py:6-18 and its train loop) as one launch forward, one backward.
    Weight and bias gradients go through ``nn.deferred`` like every other Linear of the step."""

    @staticmethod
    def forward(ctx, h, W1, b1, W2, b2, target, cost=None):
        h = h.contiguous()
        target = target.contiguous()
        r, y, y_out, loss = kernels.tgcn_head_fwd(h, W1, b1, W2, b2, target,
                                                  loss_in=None if cost is None else cost.contiguous())
        ctx.save_for_backward(h, r, y, y_out, target, W1, W2)
        ctx.params = (W1, b1, W2, b2)
        ctx.has_cost = cost is not None
        ctx.set_materialize_grads(False)
        return y, y_out, loss.reshape(())

    @staticmethod
    def backward(ctx, g_y, g_yout, g_loss):
        h, r, y, y_out, target, W1, W2 = ctx.saved_tensors
        W1p, b1p, W2p, b2p = ctx.params
        cont = lambda t: None if t is None else t.contiguous()  # noqa: E731
        dh, dyt, dyo = kernels.tgcn_head_bwd(cont(g_loss), cont(g_y), cont(g_yout), h, y_out, target, W1, W2)
        grads = [None, None, None, None]
        if deferred_weight_grads() and all(p.is_leaf for p in ctx.params):
            acc = deferred.current()
            acc.add(("head1", id(W1p)), dyt, r, sink=lambda d, W=W1p: deferred.add_to_grad(W, d),
                    colsum_sink=lambda d, b=b1p: deferred.add_to_grad(b, d))
            acc.add(("head2", id(W2p)), dyo, y, sink=lambda d, W=W2p: deferred.add_to_grad(W, d),
                    colsum_sink=lambda d, b=b2p: deferred.add_to_grad(b, d))
        else:
            gW1, gb1 = kernels.gemm_tn(dyt, r, colsum=True)
            gW2, gb2 = kernels.gemm_tn(dyo, y, colsum=True)
            grads = [gW1, gb1, gW2, gb2]
        # the running cost passes its gradient through: d(cost + loss) / d cost = 1
        return (dh, *grads, None, g_loss if ctx.has_cost else None)


def tgcn_head_usable(h: torch.Tensor, W1: torch.Tensor, b1, W2: torch.Tensor, b2, target: torch.Tensor) -> bool:
    return (h.is_cuda and h.dim() == 2 and h.dtype == torch.float32 and b1 is not None and b2 is not None
            and W1.dim() == 2 and W2.dim() == 2 and W1.shape[1] == h.shape[1] and W2.shape[1] == W1.shape[0]
            and target.dtype == torch.float32 and target.numel() == h.shape[0] * W2.shape[0]
            and all(t.is_contiguous() for t in (W1, b1, W2, b2))
            and kernels.tgcn_head_supported(h.shape[1], W1.shape[0], W2.shape[0]))


def tgcn_head(h, W1, b1, W2, b2, target, cost=None):
    """Returns ``(y, y_out, loss)`` as ``relu -> F.linear -> F.linear -> torch.mean((y_out - target) ** 2)`` would
    (fp32 rounding apart); the fused launch when ``tgcn_head_usable``, that composition otherwise.  ``cost`` (a 0-dim
    or one-element tensor): the training loop's running cost -- the third result is then ``cost + loss``."""
    if tgcn_head_usable(h, W1, b1, W2, b2, target):
        if cost is not None and not (torch.is_tensor(cost) and cost.is_cuda and cost.dtype == torch.float32
                                     and cost.numel() == 1):
            y, y_out, loss = _TgcnHead.apply(h, W1, b1, W2, b2, target)
            return y, y_out, cost + loss
        return _TgcnHead.apply(h, W1, b1, W2, b2, target, cost)
    y = linear(F.relu(h), W1, b1)
    y_out = linear(y, W2, b2)
    loss = torch.mean((y_out - target) ** 2)
    return y, y_out, loss if cost is None else cost + loss


def _known_colsum(g: torch.Tensor):
    """The column sums `_CrossEntropy.backward` left on the gradient tensor ``g``, or None when ``g`` is not that tensor any
    more: another tensor object (autograd summed two gradients out of place), or the same object modified since (summed in
    place: the version counter moved)."""
    tag = getattr(g, "_stg_colsum", None)
    if tag is None:
        return None
    colsum, version, ptr = tag
    if g._version != version or g.data_ptr() != ptr:
        return None
    return colsum


def _tagged(t: torch.Tensor, name: str):
    """The object a producer left on tensor ``t`` under ``name`` (with the tensor's version and address at that time), or None
    when ``t`` has been written since or is another tensor."""
    tag = getattr(t, name, None)
    if tag is None:
        return None
    what, version, ptr = tag
    if t._version != version or t.data_ptr() != ptr:
        return None
    return what


def _tag(t: torch.Tensor, name: str, what) -> None:
    setattr(t, name, (what, t._version, t.data_ptr()))


class _CrossEntropy(torch.autograd.Function):
    """``F.cross_entropy(logits, labels)`` (mean) as one launch each way (csrc/xent.hip)."""

    @staticmethod
    def forward(ctx, logits, labels, rows=None):
        logits = logits.contiguous()
        labels = labels.contiguous()
        ctx.d = ctx.colsum = None
        ctx.small = kernels.xent_small_usable(logits)
        if ctx.small:
            # a matrix one workgroup covers (Cora): one launch each way instead of five -- launch count is what a small graph's epoch costs
            loss, lse, n_counted, _ = kernels.xent_small_fwd(logits, labels, rows)
        elif ctx.needs_input_grad[0] and kernels.xent_fwd_grad_usable(logits):
            # the loss is going to be differentiated: the gradient for an upstream gradient of 1 from the same pass over the logits
            loss, lse, n_counted, _, ctx.d, ctx.colsum = kernels.xent_fwd_grad(logits, labels, rows)
        else:
            loss, lse, n_counted, _ = kernels.xent_fwd(logits, labels, rows)
        ctx.save_for_backward(logits, labels, lse, n_counted)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, labels, lse, n_counted = ctx.saved_tensors
        if ctx.small:
            d, colsum = kernels.xent_small_bwd(g.contiguous().float(), logits, labels, lse, n_counted)
        elif ctx.d is not None:
            d, colsum = ctx.d, ctx.colsum
            ctx.d = ctx.colsum = None                    # scaled in place below: a second backward takes the two-pass form
            kernels.xent_scale_grad(d, colsum, g.contiguous().float())
        else:
            d, colsum = kernels.xent_bwd(g.contiguous(), logits, labels, lse, n_counted, want_colsum=True)
        if colsum is not None:
            # the gradient's column sums ride along on the tensor object: a bias layer right below the loss (GCNConv's
            # `h + self.bias`) takes them as its bias gradient instead of re-reading the matrix (_GcnLayerTail.backward).
            # They describe the tensor AS WRITTEN HERE: the version counter and the storage address go with them, so a consumer
            # can tell when autograd has since accumulated another consumer's gradient into the same tensor in place (logits
            # with a second loss term) and must re-read the matrix.
            d._stg_colsum = (colsum, d._version, d.data_ptr())
        return d, None, None


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor, rows: int | None = None) -> torch.Tensor:
    """``nn.CrossEntropyLoss()(logits, labels)``: the fused launches for 2-D fp32 device logits and int64 class
    labels on the same device, ``F.cross_entropy`` otherwise (same value and gradient up to fp32 rounding).  Rows
    labelled ``ignore_index`` = -100 are left out of the sum, the mean's denominator and the gradient, as in torch;
    any other label outside [0, K) -- a device assert in torch -- is left out the same way and recorded in a sticky
    per-device word that ``kernels.check_xent_status()`` reads (one sync; call it where that is affordable).
    ``rows``: take the loss on ``logits[:rows]`` / ``labels[:rows]`` (the train-mask prefix of the GCN scripts)
    without slicing -- the backward then writes the whole gradient matrix (zero beyond ``rows``) in its one launch
    instead of autograd's fill + copy."""
    n = logits.shape[0] if rows is None else int(rows)
    if (logits.is_cuda and logits.dim() == 2 and logits.dtype == torch.float32 and labels.dtype == torch.int64
            and labels.is_cuda and labels.device == logits.device
            and labels.dim() == 1 and labels.shape[0] >= n and 0 < n <= logits.shape[0]):
        return _CrossEntropy.apply(logits, labels, None if rows is None else n)
    return F.cross_entropy(logits[:n], labels[:n])


class _LinkHead(torch.autograd.Function):
    """(y, loss) = link_head(h): ``y = relu(h) W1^T + b1``, ``logit_e = <y[src_e], y[dst_e]>``,
    ``loss = BCEWithLogits(logits, target)`` (mean) -- the head, decoder and per-timestep loss of the
    dynamic-temporal harness (benchmarking/dynamic-temporal-tgcn/seastar/model.py:5-21 and its train loop) as
    three launches forward, two backward; the backward sums per node over a sorted incidence list (no atomics)."""

    @staticmethod
    def forward(ctx, h, W1, b1, edge_index, target, incidence, cost=None):
        h = h.contiguous()
        r, y, logits, loss = kernels.link_head_fwd(h, W1, b1, edge_index, target,
                                                   loss_in=None if cost is None else cost.contiguous())
        ctx.save_for_backward(h, r, y, logits, target, W1, *incidence)
        ctx.params = (W1, b1)
        ctx.has_cost = cost is not None
        ctx.set_materialize_grads(False)
        return y, loss.reshape(())

    @staticmethod
    def backward(ctx, g_y, g_loss):
        h, r, y, logits, target, W1, row_ptr, other, eid = ctx.saved_tensors
        W1p, b1p = ctx.params
        cont = lambda t: None if t is None else t.contiguous()  # noqa: E731
        dh, dyt = kernels.link_head_bwd(cont(g_loss), cont(g_y), h, y, logits, target, (row_ptr, other, eid), W1)
        if deferred_weight_grads() and W1p.is_leaf and b1p.is_leaf:
            deferred.current().add(("head1", id(W1p)), dyt, r, sink=lambda d, W=W1p: deferred.add_to_grad(W, d),
                                   colsum_sink=lambda d, b=b1p: deferred.add_to_grad(b, d))
            gW1 = gb1 = None
        else:
            gW1, gb1 = kernels.gemm_tn(dyt, r, colsum=True)
        return dh, gW1, gb1, None, None, None, (g_loss if ctx.has_cost else None)


_INCIDENCE = {}


def _incidence_of(edge_index: torch.Tensor, N: int):
    """kernels.link_incidence(edge_index, N), kept per index tensor (a training loop passes the same tensor for
    a timestamp every epoch); keyed on identity, storage address and version counter."""
    key = id(edge_index)
    stamp = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), N)
    hit = _INCIDENCE.get(key)
    if hit is not None and hit[0] == stamp and hit[1]() is edge_index:
        return hit[2]
    if len(_INCIDENCE) > 4096:
        _INCIDENCE.clear()
    import weakref
    inc = kernels.link_incidence(edge_index, N)
    _INCIDENCE[key] = (stamp, weakref.ref(edge_index), inc)
    return inc


def link_head_usable(h, W1, b1, edge_index, target) -> bool:
    return (h.is_cuda and h.dim() == 2 and h.dtype == torch.float32 and b1 is not None and W1.dim() == 2
            and W1.shape[1] == h.shape[1] and edge_index.dtype == torch.int64 and edge_index.dim() == 2
            and edge_index.shape[0] == 2 and edge_index.shape[1] > 0 and edge_index.is_contiguous()
            and target.dtype == torch.float32 and target.is_contiguous() and target.numel() == edge_index.shape[1]
            and h.shape[0] > 0 and W1.is_contiguous() and b1.is_contiguous()
            and kernels.link_head_supported(h.shape[1], W1.shape[0]))


def link_head(h, W1, b1, edge_index, target, cost=None):
    """Returns ``(y, loss)`` as ``y = linear(relu(h))``, ``BCEWithLogitsLoss()((y[ei[0]] * y[ei[1]]).sum(-1), target)``
    would (fp32 rounding apart); the fused launches when ``link_head_usable``, that composition otherwise.  ``cost``
    (a one-element device tensor): the loop's running cost -- the second result is then ``cost + loss``."""
    if link_head_usable(h, W1, b1, edge_index, target):
        inc = _incidence_of(edge_index, h.shape[0])
        if cost is not None and not (torch.is_tensor(cost) and cost.is_cuda and cost.dtype == torch.float32
                                     and cost.numel() == 1):
            y, loss = _LinkHead.apply(h, W1, b1, edge_index, target, inc)
            return y, cost + loss
        return _LinkHead.apply(h, W1, b1, edge_index, target, inc, cost)
    y = linear(F.relu(h), W1, b1)
    out = (y[edge_index[0]] * y[edge_index[1]]).sum(dim=-1).view(-1)
    loss = F.binary_cross_entropy_with_logits(out, target)
    return y, loss if cost is None else cost + loss


def _small_layer(x: torch.Tensor, w: torch.Tensor) -> bool:
    """A dense layer one workgroup holds (Cora's 2708 x 16 -> 7): its backward is one launch (kernels.mm_bwd_small)."""
    return (x.is_cuda and x.dtype == torch.float32 and w.dim() == 2 and x.requires_grad and w.requires_grad
            and bool(kernels._MM_BWD_SMALL) and bool(kernels._C.lib.stg_mm_bwd_small_supported(int(x.shape[0]), int(x.shape[1]), int(w.shape[1]))))


def _small_graph_layer(x: torch.Tensor, w: torch.Tensor) -> bool:
    """A layer on a small graph whose weight gradient x^T g is one launch of kernels.gemm_tn_small (any input width, <= 16 outputs)."""
    return (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and w.dim() == 2 and w.requires_grad
            and not x.requires_grad and x.is_contiguous() and bool(kernels._MM_BWD_SMALL)
            and bool(kernels._C.lib.stg_gemm_tn_small_supported(int(x.shape[0]), int(x.shape[1]), int(w.shape[1]))))


def mm(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    if x.dim() == 2 and (_use_native(x, x.shape[0], x.shape[1], w.shape[1]) or _small_layer(x, w) or _small_graph_layer(x, w)):
        return _MM.apply(x, w)
    return torch.mm(x, w)


def linear(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None = None) -> torch.Tensor:
    if x.dim() == 2 and _use_native(x, x.shape[0], w.shape[0], w.shape[1]):
        return _Linear.apply(x, w, b)
    return F.linear(x, w, b)


class _AggTransform(torch.autograd.Function):
    """``A_hat(x W)`` computed as ``(A_hat x) W`` by the fused kernel (kernels.gcn_agg_transform)."""

    @staticmethod
    def forward(ctx, x, W, norm, ew, fwd_csr, bwd_csr, use_nid):
        out, P = kernels.gcn_agg_transform(x, W, norm, norm, fwd_csr, ew=ew, use_node_ids=use_nid)
        ctx.save_for_backward(P, W, norm, ew if ew is not None else norm.new_empty(0))
        ctx.has_ew = ew is not None
        ctx.bwd_csr, ctx.use_nid = bwd_csr, use_nid
        return out

    @staticmethod
    def backward(ctx, g):
        P, W, norm, ew = ctx.saved_tensors
        ew = ew if ctx.has_ew else None
        g = g.contiguous()
        dW = dx = None
        if ctx.needs_input_grad[1]:
            dW = kernels.gemm_tn(P, g) if _use_native(P, P.shape[0], P.shape[1], g.shape[1]) else torch.mm(P.t(), g)
        if ctx.needs_input_grad[0]:
            z = torch.mm(g, W.t())                                    # d(A_hat x) = g W^T          [N, Fin]
            dx = kernels.gcn_agg(z, norm, norm, ctx.bwd_csr, ew=ew, use_node_ids=ctx.use_nid)   # A_hat^T z
        return dx, dW, None, None, None, None, None


class _GcnLayerTail(torch.autograd.Function):
    """``act(A_hat h + bias)`` as ONE launch (kernels.gcn_agg with the layer epilogue), and its backward as
    two: ReLU mask + bias gradient in one pass (kernels.bias_act_bwd), then the transposed aggregation.
    Reference: the emitted GCN unit followed by ``h + self.bias`` and ``self.activation(h)``
    (nn/pytorch/static/gcn_conv.py:160-188)."""

    @staticmethod
    def forward(ctx, h, bias, norm, ew, fwd_csr, bwd_csr, use_nid, act):
        out = kernels.gcn_agg(h, norm, norm, fwd_csr, ew=ew, use_node_ids=use_nid, bias=bias, act=act)
        ctx.save_for_backward(out if act != kernels.ACT_NONE else norm.new_empty(0), norm,
                              ew if ew is not None else norm.new_empty(0))
        ctx.has_ew, ctx.act, ctx.has_bias = ew is not None, act, bias is not None
        ctx.bwd_csr, ctx.use_nid, ctx.bias = bwd_csr, use_nid, bias
        if act == kernels.ACT_RELU:
            _tag(out, "_stg_relu_out", True)
        return out

    @staticmethod
    def backward(ctx, g):
        out, norm, ew = ctx.saved_tensors
        ew = ew if ctx.has_ew else None
        known = _known_colsum(g)                         # column sums of THIS tensor object, left by the loss's / the consumer's backward
        masked_by = _tagged(g, "_stg_relu_masked_out")   # the consumer's backward already multiplied g by [out > 0] (_MM.backward, small layers)
        g = g.contiguous()
        want_b = ctx.has_bias and ctx.needs_input_grad[1]
        gb = None
        same = (masked_by is not None and masked_by.data_ptr() == out.data_ptr() and masked_by.shape == out.shape
                and masked_by._version == out._version)
        if (ctx.act == kernels.ACT_RELU and same and known is not None and known.numel() == g.shape[-1] and g.dim() == 2):
            gb = known if want_b else None
        elif ctx.act == kernels.ACT_NONE and want_b and known is not None and known.numel() == g.shape[-1] and g.dim() == 2:
            gb = known
        elif ctx.act != kernels.ACT_NONE or want_b:
            g, gb = kernels.bias_act_bwd(g, out if ctx.act != kernels.ACT_NONE else None, want_colsum=want_b)
        gh = None
        if ctx.needs_input_grad[0]:
            gh = kernels.gcn_agg(g, norm, norm, ctx.bwd_csr, ew=ew, use_node_ids=ctx.use_nid)
        return gh, gb, None, None, None, None, None, None


class _InputLayer(torch.autograd.Function):
    """A GCN layer whose INPUT needs no gradient (the first layer on the dataset's features), as
    ``act((A_hat x) W + bias)`` instead of ``act(A_hat (x W) + bias)`` (reference nn/pytorch/static/gcn_conv.py:158-188;
    equal by linearity, fp32 rounding apart).  The aggregate ``P = A_hat x`` is what the weight gradient needs
    (``dW = P^T g`` = ``x^T (A_hat^T g)``), so the layer's backward has NO aggregation: the reference order spends
    one there only to reach a gradient with respect to ``x W`` that then feeds ``dW`` alone.  Used when the
    aggregation is not wider this way round (in <= out)."""

    @staticmethod
    def forward(ctx, x, w, bias, norm, ew, fwd_csr, use_nid, act):
        P = kernels.gcn_agg(x, norm, norm, fwd_csr, ew=ew, use_node_ids=use_nid)
        bits = None
        fused_relu = getattr(torch, "_addmm_activation", None)
        if (w.is_contiguous() and act in (kernels.ACT_NONE, kernels.ACT_RELU) and kernels.rowgemm16_usable(P, w.shape[0], w.shape[1])
                and w.data_ptr() % 16 == 0):
            if act == kernels.ACT_RELU and kernels.rowgemm_bits_usable(P, w.shape[0], w.shape[1]):
                out, bits = kernels.rowgemm_relu_bits(P, w, bias)  # ... and [out > 0] as bits for the backward (16 bytes per row)
            else:
                out = kernels.rowgemm_act(P, w, bias, False, act)  # product, bias and activation in one launch (csrc/rowgemm.hip)
        elif bias is not None and w.is_contiguous() and (act == kernels.ACT_NONE or fused_relu is not None):
            # bias (+ ReLU) in the library GEMM's epilogue (hipBLASLt): 0.36 ms at [1M, 128] x [128, 128] against
            # 0.42 + 0.17 for rocBLAS + one more pass
            out = fused_relu(bias, P, w) if act == kernels.ACT_RELU else torch.addmm(bias, P, w)
        else:
            out = _mm(P, w)
            kernels.bias_act_fwd_(out, bias, act)
        ctx.save_for_backward(P, out if act != kernels.ACT_NONE else norm.new_empty(0))
        ctx.act, ctx.has_bias, ctx.w = act, bias is not None, w
        ctx.relu_bits = bits
        if bits is not None:
            _tag(out, "_stg_relu_bits", bits)
        return out

    @staticmethod
    def backward(ctx, g):
        P, out = ctx.saved_tensors
        g = g.contiguous()
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        want_w = ctx.needs_input_grad[1]
        gb = gw = None
        native = _use_native(P, P.shape[0], P.shape[1], g.shape[1])
        masked = _tagged(g, "_stg_relu_masked")
        if ctx.act == kernels.ACT_RELU and want_w and native and masked is not None and masked is ctx.relu_bits:
            # the consumer's backward already multiplied g by [out > 0] (_MM.backward): the plain contraction and its column sums
            gwt, gb = kernels.gemm_tn(g, P, colsum=True)
            return None, gwt.t().contiguous(), (gb if want_b else None), None, None, None, None, None
        if ctx.act == kernels.ACT_RELU and want_w and native and (int(P.shape[0]) * max(P.shape[1], g.shape[1]) < (1 << 29)):
            # nothing but dW and db needs the masked gradient (the input carries none): one launch forms
            # (g * [out > 0])^T P and its column sums, g * [out > 0] is never written
            gwt, gb = kernels.gemm_tn_relu_mask(g, out, P, colsum=True)
            return None, gwt.t().contiguous(), (gb if want_b else None), None, None, None, None, None
        if ctx.act != kernels.ACT_NONE or want_b:
            g, gb = kernels.bias_act_bwd(g, out if ctx.act != kernels.ACT_NONE else None, want_colsum=want_b)
        if want_w:
            W = ctx.w
            if native and deferred_weight_grads() and W.is_leaf:
                deferred.current().add(("mm", id(W)), P, g, sink=lambda d, W=W: deferred.add_to_grad(W, d))
            else:
                gw = kernels.gemm_tn(P, g) if native else torch.mm(P.t(), g)
        return None, gw, gb, None, None, None, None, None


_INPUT_LAYER = True


def set_input_layer_reorder(on: bool) -> None:
    """False: every GCNConv runs in the reference's order (x W first), whatever its input."""
    global _INPUT_LAYER
    _INPUT_LAYER = bool(on)


def input_layer_usable(graph, x: torch.Tensor, weight: torch.Tensor, activation) -> bool:
    return (_INPUT_LAYER and not x.requires_grad and weight.shape[0] <= weight.shape[1]
            and gcn_layer_tail_usable(graph, x, activation) and kernels._EDGE_CACHE and not kernels.reference_compat())


def input_layer(graph, x: torch.Tensor, weight, bias, activation, edge_weight=None) -> torch.Tensor:
    norm = graph.get_ndata("norm")
    return _InputLayer.apply(x, weight, bias, norm, edge_weight, graph.csr("fwd"),
                             kernels.rows_by_node_ids(graph.graph_type()), activation_code(activation))


def activation_code(activation):
    """STG_ACT_* of a GCNConv ``activation`` argument, or None if it is not one the epilogue implements."""
    if activation is None:
        return kernels.ACT_NONE
    if activation in (torch.relu, F.relu, torch.nn.functional.relu) or (
            isinstance(activation, torch.nn.ReLU)):
        return kernels.ACT_RELU
    return None


def gcn_layer_tail_usable(graph, h: torch.Tensor, activation) -> bool:
    """Static graphs only: a dynamic graph's backward CSR is reached through the executor's timestamp
    stack (compiler/executor.py), which this shortcut does not enter."""
    from ..compiler import dispatch
    from ..graph.dynamic.dynamic_graph import DynamicGraph
    return (h.is_cuda and h.dtype == torch.float32 and h.dim() == 2 and hasattr(graph, "csr")
            and not isinstance(graph, DynamicGraph) and kernels.layer_epilogue_usable()
            and not dispatch._FORCE_GENERATED
            and activation_code(activation) is not None)


def gcn_layer_tail(graph, h: torch.Tensor, bias, activation, edge_weight=None) -> torch.Tensor:
    norm = graph.get_ndata("norm")
    return _GcnLayerTail.apply(h, bias, norm, edge_weight, graph.csr("fwd"), graph.csr("bwd"),
                               kernels.rows_by_node_ids(graph.graph_type()), activation_code(activation))


class _GatLayer(torch.autograd.Function):
    """GATConv from the projected features on: attention projections (one pass), the emitted GAT units K0/K1
    forward; K2 backward followed by ONE pass that adds the projection terms to the feature gradient and forms
    the attn_l / attn_r gradients (reference nn/pytorch/static/gat_conv.py:43-56)."""

    @staticmethod
    def forward(ctx, feat, attn_l, attn_r, fwd_csr, bwd_csr, use_nid, slope):
        feat = feat.contiguous()
        el, er = kernels.gat_proj_fwd(feat, attn_l, attn_r)
        out, A, S = kernels.gat_fwd(el, er, feat, fwd_csr, slope, use_nid, ones_shortcut=True)
        ctx.save_for_backward(feat, attn_l, attn_r, el, er, A, S, out)
        ctx.csrs, ctx.use_nid, ctx.slope = (fwd_csr, bwd_csr), use_nid, slope
        return out

    @staticmethod
    def backward(ctx, g):
        feat, attn_l, attn_r, el, er, A, S, out = ctx.saved_tensors
        fwd_csr, bwd_csr = ctx.csrs
        gf, gel, ger = kernels.gat_bwd(A, S, out, g.contiguous(), el, er, feat, fwd_csr, bwd_csr, ctx.slope, ctx.use_nid)
        dfeat, dal, dar = kernels.gat_proj_bwd(feat, attn_l, attn_r, gel, ger, gf, inplace=True)
        return dfeat, dal.view_as(attn_l), dar.view_as(attn_r), None, None, None, None


class _GatFcLayer(torch.autograd.Function):
    """The whole GATConv from its INPUT on: ``feat = x @ fc.weight.T`` with the attention projections in the GEMM's
    epilogue (stg_gat_fc_fwd: feat is written once and not re-read for el / er), then K0 / K1; backward as _GatLayer,
    followed by the fc backward of _Linear (input gradient on rocBLAS, weight gradient on the split-K MFMA kernel or
    deferred to the end of backward)."""

    @staticmethod
    def forward(ctx, x, w, attn_l, attn_r, fwd_csr, bwd_csr, use_nid, slope, H, D, elu):
        uniform = kernels.gat_uniform_usable(x, H, D)
        # the uniform-attention form with its own backward unit never reads feat: left unwritten unless a score is not finite
        lazy = uniform and _GAT_PROJ_FOLD and kernels.gat_bwd_uniform_shape(x, H, D)
        feat, el, er = kernels.gat_fc_fwd(x, w, attn_l, attn_r, H, D, store_feat=not lazy)
        if uniform:
            # K1 at the input width, then the product with W (and the layer's elu in its epilogue)
            out, act, A, S = kernels.gat_fwd_uniform(x, w, el, er, feat, fwd_csr, slope, use_nid, elu, feat_unwritten=lazy)
        else:
            out, A, S = kernels.gat_fwd(el, er, feat, fwd_csr, slope, use_nid, ones_shortcut=True)
            act = F.elu(out) if elu else None
        ctx.save_for_backward(x, w, feat, attn_l, attn_r, el, er, A, S, out)     # `out`: the pre-activation rows
        ctx.csrs, ctx.use_nid, ctx.slope, ctx.w, ctx.elu = (fwd_csr, bwd_csr), use_nid, slope, w, bool(elu)
        ctx.feat_lazy = lazy
        return act if elu else out

    @staticmethod
    def backward(ctx, g):
        x, w, feat, attn_l, attn_r, el, er, A, S, out = ctx.saved_tensors
        fwd_csr, bwd_csr = ctx.csrs
        if _GAT_PROJ_FOLD and kernels.gat_bwd_uniform_usable(A, x, feat.shape[1], feat.shape[2]):
            return _GatFcLayer._backward_uniform(ctx, g)
        if ctx.feat_lazy:                                   # (a switch moved between forward and backward: the unit below gathers feat)
            kernels.gat_fc_feat_if(x, w, feat, None)
            ctx.feat_lazy = False
        gf, gel, ger = kernels.gat_bwd(A, S, out, g.contiguous(), el, er, feat, fwd_csr, bwd_csr, ctx.slope, ctx.use_nid,
                                       elu=ctx.elu)
        small = None
        if _GAT_PROJ_FOLD:
            # feat = x W^T, so everything the attention projections add to the backward pass -- dfeat = gf + gel (x) attn_l + ger (x)
            # attn_r, its products with W and x, and the attn gradients sum_u gel[u, h] feat[u, h, :] -- follows at width H from
            # G = [gel | ger]^T x  [2H, fin] and A = [W_h^T attn_l[h] ; W_h^T attn_r[h]]  [2H, fin]:
            #   gx = gf W + [gel | ger] A,   gw = gf^T x + attn (x) G,   d attn_l[h] = W_h G_l[h]
            # -- dfeat [N, H, D] is never formed (stg_gat_proj_bwd: a read of feat and gf and a write of dfeat, 1.6 GB at cfg3).
            N, fin = x.shape
            H, D = feat.shape[1], feat.shape[2]
            ge = torch.cat([gel.view(N, H), ger.view(N, H)], 1)
            native16 = _use_native(x, N, 2 * H, fin)
            G = kernels.gemm_tn(ge, x) if native16 else torch.mm(ge.t(), x)
            Wh = w.view(H, D, fin)
            al, ar = attn_l.reshape(H, D), attn_r.reshape(H, D)
            dal, dar = torch.einsum("hdf,hf->hd", Wh, G[:H]), torch.einsum("hdf,hf->hd", Wh, G[H:])
            g2 = gf.view(N, H * D)
            gx = None
            if ctx.needs_input_grad[0]:
                Aw = torch.cat([torch.einsum("hdf,hd->hf", Wh, al), torch.einsum("hdf,hd->hf", Wh, ar)], 0)
                gx = torch.addmm(kernels.matmul(g2, w), ge, Aw)
            if ctx.needs_input_grad[1]:
                small = (al.unsqueeze(2) * G[:H].unsqueeze(1) + ar.unsqueeze(2) * G[H:].unsqueeze(1)).reshape(H * D, fin)
        else:
            dfeat, dal, dar = kernels.gat_proj_bwd(feat, attn_l, attn_r, gel, ger, gf, inplace=True)
            g2 = dfeat.view(dfeat.shape[0], -1)
            gx = kernels.matmul(g2, w) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            W = ctx.w
            native = _use_native(x, x.shape[0], g2.shape[1], x.shape[1])
            if native and deferred_weight_grads() and W.is_leaf:
                deferred.current().add(("linear", id(W)), g2, x, sink=lambda d, W=W: deferred.add_to_grad(W, d),
                                       colsum_sink=None)
                if small is not None:
                    deferred.add_to_grad(W, small)
            else:
                gw = kernels.gemm_tn(g2, x) if native else torch.mm(g2.t(), x)
                if small is not None:
                    gw = gw + small
        return gx, gw, dal.view_as(attn_l), dar.view_as(attn_r), None, None, None, None, None, None, None


def _gat_backward_uniform(ctx, g):
    """_GatFcLayer.backward when the forward ran in the uniform-attention form at the shapes of stg_gat_bwd_uniform_edges: the
    backward unit gathers x (width fin) and per-vertex products of g with W instead of g rows of width H * D per edge, and what
    grad_feat was needed for comes from  grad_feat W = A_hat^T (gs W)  and  grad_feat^T x = g^T xm.  The attention projections'
    terms at width H as in backward().  A device flag (a non-finite score) switches every step to the general unit's results."""
    x, w, feat, attn_l, attn_r, el, er, A, S, out = ctx.saved_tensors
    fwd_csr, bwd_csr = ctx.csrs
    N, fin = x.shape
    H, D = feat.shape[1], feat.shape[2]
    gxa, gel, ger, gq, gf, flag, xm = kernels.gat_bwd_uniform(A, S, out, g.contiguous(), x, w, feat, fwd_csr, bwd_csr, ctx.slope,
                                                                ctx.use_nid, elu=ctx.elu)
    ge = torch.cat([gel.view(N, H), ger.view(N, H)], 1)
    G = kernels.gemm_tn(ge, x) if _use_native(x, N, 2 * H, fin) else torch.mm(ge.t(), x)
    Wh = w.view(H, D, fin)
    al, ar = attn_l.reshape(H, D), attn_r.reshape(H, D)
    gx = gw = None
    if ctx.needs_input_grad[1]:
        gw = torch.empty(H * D, fin, dtype=torch.float32, device=x.device)
        kernels.gemm_tn_gated(gq.view(N, H * D), xm, gw, flag, True)         # every score finite: g^T (mean of x over in-edges)
        kernels.gemm_tn_gated(gf.view(N, H * D), x, gw, flag, False)         # otherwise: the general unit's grad_feat^T x
    if kernels.gat_attn_fold_usable(w, H, D, fin):
        # the fold's small products in ONE launch: the attention gradients, A_w for gx and the correction of gw (in place, after the
        # gated contractions above)
        dal, dar, Aw = kernels.gat_attn_fold(w, G, al, ar, H, D, fin, want_aw=ctx.needs_input_grad[0], gw=gw)
    else:
        dal, dar = torch.einsum("hdf,hf->hd", Wh, G[:H]), torch.einsum("hdf,hf->hd", Wh, G[H:])
        Aw = (torch.cat([torch.einsum("hdf,hd->hf", Wh, al), torch.einsum("hdf,hd->hf", Wh, ar)], 0) if ctx.needs_input_grad[0] else None)
        if gw is not None:
            gw += (al.unsqueeze(2) * G[:H].unsqueeze(1) + ar.unsqueeze(2) * G[H:].unsqueeze(1)).reshape(H * D, fin)
    if ctx.needs_input_grad[0]:
        gx = torch.addmm(gxa, ge, Aw)
        kernels.gat_bwd_uniform_gx_fallback(gf, w, gx, flag)
    return gx, gw, dal.view_as(attn_l), dar.view_as(attn_r), None, None, None, None, None, None, None


_GatFcLayer._backward_uniform = staticmethod(_gat_backward_uniform)


def gat_fc_layer_usable(graph, x: torch.Tensor, fc, H: int, D: int) -> bool:
    """``fc``: the layer's nn.Linear ([H*D, fin], no bias: reference nn/pytorch/static/gat_conv.py:27)."""
    from ..compiler import dispatch
    from ..graph.dynamic.dynamic_graph import DynamicGraph
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and fc.bias is None and hasattr(graph, "csr")
            and not isinstance(graph, DynamicGraph) and not kernels.reference_compat()
            and not dispatch._FORCE_GENERATED and _GAT_FC
            and kernels.gat_fc_supported(x.shape[1], H, D) and kernels.gat_proj_supported(H, D))


def gat_fc_layer(graph, x: torch.Tensor, fc, attn_l, attn_r, slope: float, H: int, D: int, elu: bool = False) -> torch.Tensor:
    """``elu``: return ``F.elu`` of the layer's result (the caller's ``activation``, see :func:`is_elu`), formed in the
    epilogue of the producing kernel and differentiated inside the backward unit's per-vertex pass."""
    return _GatFcLayer.apply(x, fc.weight, attn_l, attn_r, graph.csr("fwd"), graph.csr("bwd"),
                             kernels.rows_by_node_ids(graph.graph_type()), float(slope), int(H), int(D), bool(elu))


def is_elu(activation) -> bool:
    """A GATConv ``activation`` that is exactly ``F.elu`` with its defaults (alpha = 1, out of place)."""
    if activation is F.elu or activation is torch.nn.functional.elu:
        return True
    return isinstance(activation, torch.nn.ELU) and activation.alpha == 1.0 and not activation.inplace


_GAT_FC = True
_GAT_PROJ_FOLD = True


def set_gat_proj_fold(on: bool) -> None:
    """False: the fused GATConv's backward forms dfeat and its attention-projection terms at full width (stg_gat_proj_bwd), as the
    un-fused layer does; True (default): at width H from [gel | ger]^T x (see _GatFcLayer.backward)."""
    global _GAT_PROJ_FOLD
    _GAT_PROJ_FOLD = bool(on)



def set_gat_fc(on: bool) -> None:
    """Tests: switch the fused input side of GATConv off (x @ W.T, then stg_gat_proj_fwd)."""
    global _GAT_FC
    _GAT_FC = bool(on)


def gat_layer_usable(graph, feat3: torch.Tensor) -> bool:
    from ..compiler import dispatch
    from ..graph.dynamic.dynamic_graph import DynamicGraph
    return (feat3.is_cuda and feat3.dtype == torch.float32 and feat3.dim() == 3 and hasattr(graph, "csr")
            and not isinstance(graph, DynamicGraph) and not kernels.reference_compat()
            and not dispatch._FORCE_GENERATED
            and kernels.gat_proj_supported(feat3.shape[1], feat3.shape[2]))


def gat_layer(graph, feat3: torch.Tensor, attn_l, attn_r, slope: float) -> torch.Tensor:
    return _GatLayer.apply(feat3, attn_l, attn_r, graph.csr("fwd"), graph.csr("bwd"),
                           kernels.rows_by_node_ids(graph.graph_type()), float(slope))


def agg_transform_usable(graph, x: torch.Tensor, W) -> bool:
    """The fused kernel pays (and is supported) when the gather can run at the narrow input width.
    ``W``: the weight tensor or its (in, out) shape."""
    fin, fout = (int(W.shape[0]), int(W.shape[1])) if isinstance(W, torch.Tensor) else (int(W[0]), int(W[1]))
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and fin < fout
            and kernels.agg_transform_supported(fin, fout)
            and kernels._EDGE_CACHE and not kernels.reference_compat() and hasattr(graph, "csr"))


def agg_transform(graph, x: torch.Tensor, W: torch.Tensor, edge_weight=None) -> torch.Tensor:
    """``GCNConv``'s ``aggregate(graph, x @ W)`` as one fused launch; captures the graph's CURRENT
    forward/backward CSR (dynamic graphs: the snapshot of this timestamp) for the backward pass."""
    norm = graph.get_ndata("norm")
    return _AggTransform.apply(x, W, norm, edge_weight, graph.csr("fwd"), graph.csr("bwd"),
                               kernels.rows_by_node_ids(graph.graph_type()))
