"""Temporal training loops around the hot path, and their multi-GPU form.

Single process: the window loop of the reference's static-temporal and
dynamic-temporal TGCN scripts (benchmarking/static-temporal-tgcn/seastar/
train.py:153-205, benchmarking/dynamic-temporal-tgcn/seastar/train.py:179-231):
every window of ``backprop_every`` consecutive snapshots starts from
``hidden_state = None`` and a fresh ``randn`` input, accumulates the loss, divides
it by ``backprop_every + 1`` (sic), backpropagates through time and
takes one optimizer step.

Multi GPU (new -- the reference has no distributed path): windows
carry no state across each other, so window ``w`` goes to rank ``w mod R``; every
rank holds the full (small) graph and model, and one optimizer step consumes R
windows.  The ONLY collective is one all-reduce (sum, then / R) of the flattened
gradient bucket per optimizer step (RCCL over xGMI when the process group is
"nccl"; ~133 KB for TGCN(32 -> 64), i.e. latency bound, hence a single contiguous
bucket and a single call).  With R = 1 the loop is step-for-step the reference's.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .capture import capture as _graph_capture
from .nn import functional as SF
from .nn.pytorch.temporal.tgcn import TGCN


_FUSED_HEAD = True
_FUSED_WINDOW = True


def set_fused_head(enabled: bool) -> None:
    """True (default): the training loops below run the model head and the per-step loss as one fused launch
    (nn.functional.tgcn_head); False: ``model(...)`` followed by ``torch.mean((y_out - target) ** 2)``, as the
    reference's script spells it."""
    global _FUSED_HEAD
    _FUSED_HEAD = bool(enabled)


def set_fused_window(enabled: bool) -> None:
    """True (default): where the shapes allow (``window_cost_usable``) the static-temporal loops run every snapshot of
    a BPTT window as ONE launch forward and ONE backward (csrc/tgcn_step.hpp) inside a single autograd node, with the
    weight gradients taken once per window; False: one autograd node per snapshot (``STGraphTGCN.step_loss``)."""
    global _FUSED_WINDOW
    _FUSED_WINDOW = bool(enabled)


_ZEROS = {}


def _zeros(n: int, c: int, device) -> torch.Tensor:
    """A zero matrix kept per (shape, device): the first snapshot's hidden state as a weight-gradient operand."""
    key = (n, c, str(device))
    z = _ZEROS.get(key)
    if z is None:
        if len(_ZEROS) > 16:
            _ZEROS.clear()
        z = _ZEROS[key] = torch.zeros(n, c, device=device)
    return z


# Inside the captured window bodies (CapturedStaticWindow / CapturedDynamicWindows) the window's autograd node writes every
# parameter gradient straight into the parameter's ``.grad`` -- a view of the gradient bucket, zeroed at the top of the body -- and
# returns None for it: autograd's AccumulateGrad would otherwise add each of the 16 gradients into its (zero) view with a launch
# of its own, 13-16 launches of ~4.6 us per window.  Opt-in per backward pass; everywhere else the gradients are returned.
_DIRECT_GRADS = [False]


class direct_param_grads:
    def __enter__(self):
        self.prev, _DIRECT_GRADS[0] = _DIRECT_GRADS[0], True

    def __exit__(self, *exc):
        _DIRECT_GRADS[0] = self.prev


def _grad_sinks(params, needs):
    """The ``.grad`` tensors of ``params`` if the window node may write them directly, else None (the gradients are then
    returned to autograd).  Direct writes OVERWRITE ``.grad`` and bypass AccumulateGrad (no accumulation with another node's
    contribution, no post-accumulate hooks), so they are taken only when: the caller opted in (``direct_param_grads``: the
    captured window bodies, which zero the bucket first and use every parameter in this one node); autograd asks for every
    one of these gradients (``needs``: ctx.needs_input_grad of the parameters) and each parameter requires grad -- a frozen
    parameter with a stale ``.grad`` is never written; each ``.grad`` is a contiguous fp32 tensor of the parameter's shape on
    its device; and no two parameters are the same tensor (a parameter used twice would receive two writes)."""
    if not _DIRECT_GRADS[0]:
        return None
    if len(needs) != len(params) or not all(needs) or len({id(p) for p in params}) != len(params):
        return None
    sinks = []
    for p in params:
        g = getattr(p, "grad", None)
        if (not p.requires_grad or g is None or g.shape != p.shape or g.dtype != torch.float32 or g.device != p.device
                or not g.is_contiguous()):
            return None
        sinks.append(g)
    return sinks


# True while an epoch function holds a snapshot of the training state and reads the fold status word itself at the end of the
# epoch (_FoldGuard): the window nodes then skip their own per-window read-back -- a blocking host sync per BPTT window.
_EPOCH_GUARD = [False]


def _fold_switch_off(where: str) -> None:
    from . import kernels
    import warnings
    kernels.set_step_folded(False)
    kernels.set_step_wgrad_from_p(False)
    warnings.warn("stgraph_amd: a TGCN conv output may leave [-1e6, 1e6] on this data (reference nn/pytorch/temporal/tgcn.py:23 clamps there): "
                  "the folded step formulation is switched off for this process (kernels.set_step_folded / set_step_wgrad_from_p) and "
                  f"{where} recomputed in the reference formulation", RuntimeWarning, stacklevel=3)


class _FoldGuard:
    """What makes an epoch on the folded step formulation safe to run without a host read per window: a snapshot of every
    parameter and of the optimizer's state taken at the start of the epoch (a few ``_foreach_copy_`` launches over ~ 400 KB; only
    while kernels.STEP_FOLDED / STEP_WGRAD_FROM_P are on), ONE read of the device's sticky status word at its end (agreed over the
    ranks with a MAX all-reduce: every rank must take the same branch), and ``restore()`` -- the training state exactly as the
    epoch found it.  The epoch function then switches the formulation off, re-captures what it replays and runs the epoch again:
    no optimizer step taken on gradients of the invalid formulation survives, nothing is raised (a drop-in must not fail on
    data the reference handles).  ``owner``: an object that keeps the snapshot buffers between epochs."""

    def __init__(self, optimizer, device, world: int = 1, group=None, owner=None):
        from . import kernels
        self.device, self.world, self.group, self.optimizer = torch.device(device), world, group, optimizer
        # (the folded launches exist on the GPU only: a CPU run -- the oracle-backed layers of tests/test_distributed_cpu.py -- has no word)
        self.active = bool(kernels.STEP_FOLDED or kernels.STEP_WGRAD_FROM_P) and self.device.type == "cuda"
        self.prev = False
        if not self.active:
            return
        params = [p for g in optimizer.param_groups for p in g["params"]]
        self.had_state = len(optimizer.state) > 0
        live = [p.data for p in params]
        for p in params:
            live += [v for v in optimizer.state.get(p, {}).values() if torch.is_tensor(v)]
        saved = getattr(owner, "_fold_guard_buffers", None)
        if saved is None or len(saved) != len(live) or any(a.shape != b.shape or a.dtype != b.dtype or a.device != b.device
                                                          for a, b in zip(saved, live)):
            saved = [torch.empty_like(t) for t in live]
            if owner is not None:
                owner._fold_guard_buffers = saved
        self.live, self.saved = live, saved
        self._copy(saved, live)

    @staticmethod
    def _copy(dst, src):
        with torch.no_grad():
            on_gpu = [(d, s) for d, s in zip(dst, src) if d.is_cuda]
            if on_gpu:
                torch._foreach_copy_([d for d, _ in on_gpu], [s for _, s in on_gpu])
            for d, s in zip(dst, src):
                if not d.is_cuda:                       # a non-capturable optimizer keeps its step counters on the host
                    d.copy_(s)

    def __enter__(self):
        self.prev, _EPOCH_GUARD[0] = _EPOCH_GUARD[0], (self.active or _EPOCH_GUARD[0])
        return self

    def __exit__(self, *exc):
        _EPOCH_GUARD[0] = self.prev
        return False

    def tripped(self) -> bool:
        """Did a step launch of this epoch (on ANY rank) refuse the folded formulation?  One 4-byte read-back; clears the word."""
        if not self.active:
            return False
        from . import kernels
        word = kernels.step_fold_status_word(self.device)
        if self.world > 1:
            dist.all_reduce(word, op=dist.ReduceOp.MAX, group=self.group)
        hit = int(word.item()) != 0
        if hit:
            word.zero_()
        return hit

    def restore(self) -> None:
        self._copy(self.live, self.saved)
        if not self.had_state:                          # the optimizer created its state during the epoch: back to none
            self.optimizer.state.clear()


def _fold_refused(fold_status) -> bool:
    """After the forward step launches of a window node, OUTSIDE a stream capture: did one of them refuse the folded formulation
    (a conv output that may leave the clamp range)?  Then the formulation is switched off for the process, with a warning, and the
    caller recomputes the window in the reference formulation -- a drop-in must not fail on data the reference handles.  Inside a
    capture nothing can be read back, and inside an epoch function nothing NEEDS to be (``_EPOCH_GUARD``): the epoch functions hold a
    snapshot of the training state, read the word once at the end of the epoch and run the epoch again (``_FoldGuard``)."""
    if fold_status is None or _EPOCH_GUARD[0] or torch.cuda.is_current_stream_capturing():
        return False
    if int(fold_status.item()) == 0:
        return False
    fold_status.zero_()
    _fold_switch_off("the window")
    return True


def _unfold_gate_grads(MgT, cs, dWbot, Wc, bc, Wg):
    """Gate + conv parameter gradients of one gate from the contractions over the window's rows that do not need x3 or da3:
    ``MgT = d_g^T P`` [C, Fin], ``cs`` = column sums of d_g [C], ``dWbot = d_g^T Hx`` [C, C] (d_g: gradient of the gate's
    pre-activation; x3_g = P Wc + bc, unclamped).  Returns (dWg [C, 2C], dbg [C], dWc [Fin, C], dbc [C]):
    ``dWg[:, :C] = d_g^T x3_g = MgT Wc + cs bc^T``; ``dWc = P^T (d_g Wg[:, :C]) = MgT^T Wg[:, :C]``; ``dbc = cs Wg[:, :C]``."""
    C = Wg.shape[0]
    top = Wg[:, :C]
    dWg = torch.cat([torch.addmm(torch.outer(cs, bc), MgT, Wc), dWbot], dim=1)
    return dWg, cs, torch.mm(MgT.t(), top), torch.mv(top.t(), cs)


class _TGCNWindow(torch.autograd.Function):
    """cost = sum_t mean((y_out_t - target_t)^2) over the snapshots of one BPTT window of the static-temporal loop
    (benchmarking/static-temporal-tgcn/seastar/train.py:165-183 with model.py:6-18 and nn/pytorch/temporal/tgcn.py):
    ``hidden = None``, ``y_hat = x0``, then ``y_out, y_hat, hidden = model(g, y_hat, w, hidden)`` per snapshot.
    One launch per snapshot each way (kernels.tgcn_step_fwd / _bwd); the gradient of a snapshot's input is aggregated
    by the previous snapshot's backward launch; six split-K launches per window give every weight gradient."""

    @staticmethod
    def forward(ctx, x0, targets, norm, ew, fwd, bwd, use_nid, lo, hi,
                Wcz, Wcr, Wch, bcz, bcr, bch, Wz, bz, Wr, br, Wh, bh, W1, b1, W2, b2):
        from . import kernels
        dev = x0.device
        B, N = int(targets.shape[0]), int(x0.shape[0])
        C, Fin, Fh = int(Wz.shape[0]), int(x0.shape[1]), int(W1.shape[0])
        x0 = x0.contiguous()
        targets = targets.reshape(B, N).contiguous()
        normv = norm.reshape(-1).contiguous()
        # Wcat [Fin, 3C], its transpose, b3 and the transposed gate / head weights of the backward pass: one launch
        Wcat, WcatT, b3, WzT, WrT, WhT, W1T = kernels.tgcn_pack_weights(Wcz, Wcr, Wch, bcz, bcr, bch, Wz, Wr, Wh, W1)
        ctx.packed_T = (WzT, WrT, WhT, W1T)
        # the gate Linears with the conv folded in: the forward launch's folded form (nobody may read x3: only with from_p)
        from_p = kernels.STEP_WGRAD_FROM_P                      # weight gradients from P: no x3, no da3
        w_fold, b_fold, f_bound, ctx.w_fold_t = (kernels.tgcn_fold_weights(Wcz, Wcr, Wch, bcz, bcr, bch, Wz, bz, Wr, br, Wh, bh, with_bound=True)
                                   if kernels.STEP_FOLDED and from_p else (None, None, None, None))
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
        P, X3 = new(B, N, Fin), (None if from_p else new(B, N, 3 * C))
        fold_status = kernels.step_fold_status_word(dev) if (from_p or w_fold is not None) else None
        Z, R, Ht, Hn, HR = (new(B, N, C) for _ in range(5))
        Y, Yout = new(B, N, Fh), new(B, N)
        tiles = kernels.tgcn_step_loss_partials(N)
        partial = new(B, tiles)
        # clamp mask of x3, one bit per column; the folded forward launch does not form x3 (it bounds it): the all-ones mask then
        mask = (kernels.step_ones_mask(N, dev).unsqueeze(0).expand(B, N, 12) if (from_p and w_fold is not None)
                else torch.empty(B, N, 12, dtype=torch.int32, device=dev))
        # The step kernels visit rows in vertex order whatever the graph type's node_ids say: every row is computed
        # independently of the order (the reference's degree-sorted visiting order is a scheduling detail of its
        # one-thread-per-row kernels), and a 16-row tile of consecutive rows loads and stores contiguous memory where a
        # tile of degree neighbours touches 16 scattered rows -- measured 59.5 / 73.8 us (vertex order) against
        # 60.9 / 82.9 (node_ids) per forward / backward launch at |V| = 50 K.
        nid = None
        with torch.cuda.device(dev):
            nc = kernels._edge_gathered(fwd, "norm", norm, fwd.column_indices)
            ew_e = None if ew is None else kernels._edge_gathered(fwd, "ew", ew, fwd.eids)
        W2v, weights = W2.reshape(-1).contiguous(), [t.contiguous() for t in (Wz, bz, Wr, br, Wh, bh, W1, b1, b2)]
        Wz_, bz_, Wr_, br_, Wh_, bh_, W1_, b1_, b2_ = weights
        for t in range(B):
            kernels.tgcn_step_fwd(N, C, Fin, Fh, 2, lo, hi, dev, row_offsets=fwd.row_offset, column_indices=fwd.column_indices,
                                  node_ids=nid, norm_col_edge=nc, ew_edge=ew_e, norm=normv,
                                  x=x0 if t == 0 else Y[t - 1], H=None if t == 0 else Hn[t - 1], target=targets[t],
                                  WcatT=WcatT, b3=b3, Wz=Wz_, bz=bz_, Wr=Wr_, br=br_, Wh=Wh_, bh=bh_, W1=W1_, b1=b1_,
                                  W2=W2v, b2=b2_, P=P[t], x3=None if from_p else X3[t], Z=Z[t], R=R[t], Ht=Ht[t], Hn=Hn[t],
                                  HR=HR[t], y=Y[t], y_out=Yout[t], loss_partial=partial[t], clamp_mask=mask[t],
                                  w_fold=w_fold, b_fold=b_fold, fold_bound=f_bound, fold_status=fold_status)
        if _fold_refused(fold_status):
            return _TGCNWindow.forward(ctx, x0, targets, norm, ew, fwd, bwd, use_nid, lo, hi,
                                       Wcz, Wcr, Wch, bcz, bcr, bch, Wz, bz, Wr, br, Wh, bh, W1, b1, W2, b2)
        step_loss = new(B)
        cost = kernels.tgcn_window_loss(partial, B, N, step_loss)
        ctx.save_for_backward(x0, targets, norm, normv, ew if ew is not None else norm.new_empty(0), Wcat, Wz_, Wr_, Wh_, W1_, W2v,
                              P, X3, Z, R, Ht, Hn, HR, Y, Yout, mask)
        ctx.has_ew, ctx.csrs, ctx.use_nid, ctx.clamp = ew is not None, (fwd, bwd), use_nid, (float(lo), float(hi))
        ctx.params = (Wcz, Wcr, Wch, bcz, bcr, bch, Wz, bz, Wr, br, Wh, bh, W1, b1, W2, b2)
        ctx.step_loss = step_loss
        return cost.reshape(())

    @staticmethod
    def backward(ctx, g_cost):
        from . import kernels
        (x0, targets, norm, normv, ew, Wcat, Wz, Wr, Wh, W1, W2v, P, X3, Z, R, Ht, Hn, HR, Y, Yout, mask) = ctx.saved_tensors
        ew = ew if ctx.has_ew else None
        fwd, bwd = ctx.csrs
        lo, hi = ctx.clamp
        dev = x0.device
        B, N, C = Z.shape
        Fin, Fh = int(x0.shape[1]), int(Y.shape[2])
        g = g_cost.reshape(1).contiguous().float()
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
        from_p = X3 is None                                     # weight gradients from P (kernels.STEP_WGRAD_FROM_P): no x3, no da3
        wide_d = from_p and kernels.STEP_WGRAD_ZR_TOGETHER
        if wide_d:
            # the gate gradients as the column blocks of ONE matrix: [d_z | d_r] is then one operand against [H | P]
            D3 = new(B, N, 3 * C)
            dzl, drl, dhl, da3 = D3[:, :, :C], D3[:, :, C:2 * C], D3[:, :, 2 * C:], None
        else:
            dzl, drl, dhl, da3 = new(B, N, C), new(B, N, C), new(B, N, C), (None if from_p else new(B, N, 3 * C))
        dyt, dyo = new(B, N, Fh), new(B, N)
        dH, zbuf = new(2, N, C), new(2, N, Fin)
        WzT, WrT, WhT, W1T = ctx.packed_T
        nid = None                                  # vertex order: see forward
        with torch.cuda.device(dev):
            nc = kernels._edge_gathered(bwd, "norm", norm, bwd.column_indices)
            ew_e = None if ew is None else kernels._edge_gathered(bwd, "ew", ew, bwd.eids)
        want_dx0 = ctx.needs_input_grad[0]
        for t in range(B - 1, -1, -1):
            last = t == B - 1
            kernels.tgcn_step_bwd(N, C, Fin, Fh, 2, lo, hi, dev, row_offsets=bwd.row_offset, column_indices=bwd.column_indices,
                                  node_ids=nid, norm_col_edge=nc, ew_edge=ew_e, norm=normv,
                                  zn=None if last else zbuf[(t + 1) & 1], g_y=None, dHn=None if last else dH[(t + 1) & 1],
                                  g_cost=g, Z=Z[t], R=R[t], Ht=Ht[t], H=None if t == 0 else Hn[t - 1], Hn=Hn[t], x3=None,
                                  clamp_mask=mask[t],
                                  y_out=Yout[t], target=targets[t], WzT=WzT, WrT=WrT, WhT=WhT, Wcat=Wcat, W1T=W1T, W2=W2v,
                                  dzl=dzl[t], drl=drl[t], dhl=dhl[t], da3=None if from_p else da3[t], dH=dH[t & 1],
                                  z=zbuf[t & 1] if (t > 0 or want_dx0 or from_p) else None, dyt=dyt[t], dyo=dyo[t],
                                  w_fold_t=ctx.w_fold_t if from_p else None, ld_d=3 * C if wide_d else 0)
        dx0 = kernels.gcn_agg(zbuf[0], norm, norm, bwd, ew=ew, use_node_ids=ctx.use_nid) if want_dx0 else None
        # weight gradients: one split-K launch per parameter over the window's snapshots
        steps = range(B)
        Hprev = [_zeros(N, C, dev)] + [Hn[t] for t in range(B - 1)]
        # .grad of (Wcz Wcr Wch bcz bcr bch Wz bz Wr br Wh bh W1 b1 W2 b2), or None
        sinks = _grad_sinks(ctx.params, ctx.needs_input_grad[9:25])
        dst = (lambda i, j: dict(out=sinks[i], colsum_out=sinks[j])) if sinks else (lambda i, j: {})  # noqa: E731
        gate = lambda d, k, second, i: dict(  # noqa: E731
            As=[d[t] for t in steps], Bs=[X3[t][:, k * C:(k + 1) * C] for t in steps], M=C, N=2 * C, B2s=second, nsplit=C,
            b_op=kernels.GEMM_B_CLAMP, lo=lo, hi=hi, colsum=True, **dst(i, i + 1))
        # six split-K contractions, ONE reduction launch (kernels.gemm_tn_form_batch); the three conv layers' gradients are one
        # stacked product whose row blocks the reduction transposes straight into the parameters' .grad (sinks 0-2, biases 3-5)
        head_calls = [
            dict(As=[dyt[t] for t in steps], Bs=[Hn[t] for t in steps], M=Fh, N=C, b_op=kernels.GEMM_B_RELU, colsum=True, **dst(12, 13)),
            dict(As=[dyo[t].view(N, 1) for t in steps], Bs=[Y[t] for t in steps], M=1, N=Fh, colsum=True, **dst(14, 15))]
        if from_p:
            # per gate d_g^T Hx [C, C] (+ column sums) and d_g^T P [C, Fin]; the gate and conv parameters' gradients follow from them in
            # a few small products (_unfold_gate_grads)
            Wcz, Wcr, Wch, bcz, bcr, bch, Wz_p, _, Wr_p, _, Wh_p, _ = ctx.params[:12]
            Rg, csg, nres = _gate_contractions(kernels, D3 if wide_d else None, (dzl, drl, dhl), Hprev, [HR[t] for t in steps],
                                               [P[t] for t in steps], steps, C, Fin, head_calls)
            gates = kernels.tgcn_unfold_gate_grads(
                Rg, csg, (Wcz, Wcr, Wch), (bcz, bcr, bch), (Wz_p, Wr_p, Wh_p),
                outs=[(sinks[6 + 2 * k], sinks[7 + 2 * k], sinks[k], sinks[3 + k]) for k in range(3)] if sinks else None)
            if sinks:
                return (dx0,) + (None,) * 24
            (dW1, db1), (dW2, db2) = nres
            return (dx0, None, None, None, None, None, None, None, None, *[g[2] for g in gates], *[g[3] for g in gates],
                    gates[0][0], gates[0][1], gates[1][0], gates[1][1], gates[2][0], gates[2][1], dW1, db1, dW2.view(1, Fh), db2)
        conv = dict(As=[da3[t] for t in steps], Bs=[P[t] for t in steps], M=3 * C, N=Fin, colsum=True)
        if sinks:
            conv.update(out_blocks_t=list(sinks[0:3]), colsum_blocks=list(sinks[3:6]))
        res = kernels.gemm_tn_form_batch([
            gate(dzl, 0, Hprev, 6), gate(drl, 1, Hprev, 8), gate(dhl, 2, [HR[t] for t in steps], 10), conv] + head_calls)
        if sinks:
            return (dx0,) + (None,) * 24
        (dWz, dbz), (dWr, dbr), (dWh, dbh), (dWcT, db3), (dW1, db1), (dW2, db2) = res
        conv_w = [dWcT[k * C:(k + 1) * C].t() for k in range(3)]
        conv_b = [db3[k * C:(k + 1) * C] for k in range(3)]
        return (dx0, None, None, None, None, None, None, None, None, *conv_w, *conv_b, dWz, dbz, dWr, dbr, dWh, dbh,
                dW1, db1, dW2.view(1, Fh), db2)


def _gate_contractions(kernels, D3, ds, Hprev, HRs, Ps, steps, C, Fin, more_calls):
    """The window's gate contractions d_g^T [Hx | P] -> [C, C + Fin] (+ column sums), reduced in one launch together with
    ``more_calls``.  ``D3`` [B, N, 3C] (the gate gradients as column blocks of one matrix, ``ds`` its views): [d_z | d_r] is ONE
    operand against [H | P], which both gates share -- two workgroups per K slice on the same XCD read it once (knob
    gemm_xcd_pair) -- and d_h goes against [H (.) R | P]; without it one contraction per gate.  Returns (R per gate, column sums
    per gate, the results of ``more_calls``)."""
    if D3 is None:
        seconds = (Hprev, Hprev, HRs)
        res = kernels.gemm_tn_form_batch(
            [dict(As=[d[t] for t in steps], Bs=sec, B2s=Ps, M=C, N=C + Fin, nsplit=C, colsum=True) for d, sec in zip(ds, seconds)]
            + list(more_calls))
        return [res[k][0] for k in range(3)], [res[k][1] for k in range(3)], res[3:]
    res = kernels.gemm_tn_form_batch(
        [dict(As=[D3[t][:, :2 * C] for t in steps], Bs=Hprev, B2s=Ps, M=2 * C, N=C + Fin, nsplit=C, colsum=True),
         dict(As=[ds[2][t] for t in steps], Bs=HRs, B2s=Ps, M=C, N=C + Fin, nsplit=C, colsum=True)] + list(more_calls))
    (Rzr, cszr), (Rh, csh) = res[0], res[1]
    return [Rzr[:C], Rzr[C:], Rh], [cszr[:C], cszr[C:], csh], res[2:]


def window_cost_usable(model, graph, x0, edge_weight, targets) -> bool:
    """Shapes / types ``window_cost`` covers: the static-temporal harness model over TGCN(32 -> 64) on a static graph."""
    from . import kernels
    from .graph.dynamic.dynamic_graph import DynamicGraph
    from .nn.pytorch.static.gcn_conv import GCNConv
    if not (_FUSED_WINDOW and _FUSED_HEAD and isinstance(model, STGraphTGCN) and type(model.temporal) is TGCN):
        return False
    tg = model.temporal
    convs = (tg.conv_z, tg.conv_r, tg.conv_h)
    return (x0.is_cuda and x0.dtype == torch.float32 and x0.dim() == 2 and hasattr(graph, "csr")
            and not isinstance(graph, DynamicGraph) and not kernels.reference_compat() and kernels._EDGE_CACHE
            and all(type(c) is GCNConv and c.bias is not None and c.activation is None for c in convs)
            and model.linear2.out_features == 1 and model.linear.bias is not None and model.linear2.bias is not None
            and targets.dtype == torch.float32 and targets.numel() == targets.shape[0] * x0.shape[0]
            and graph.get_ndata("norm") is not None
            and kernels.tgcn_step_supported(tg.out_channels, tg.in_channels, model.linear.out_features)
            and model.linear.out_features == tg.in_channels
            and x0.shape[0] * 3 * tg.out_channels < (1 << 30))


def window_cost(model, graph, x0, edge_weight, targets) -> torch.Tensor:
    """``sum_t mean((y_out_t - targets[t]) ** 2)`` of one BPTT window (``targets`` [B, N] or [B, N, 1]) starting from
    ``hidden = None`` and input ``x0``: the reference loop's ``cost`` before its division by ``backprop_every + 1``."""
    from . import kernels
    from .nn.pytorch.static.gcn_conv import GCNConv
    tg = model.temporal
    GCNConv.check_norm(graph)
    return _TGCNWindow.apply(
        x0, targets, graph.get_ndata("norm"), edge_weight, graph.csr("fwd"), graph.csr("bwd"),
        kernels.rows_by_node_ids(graph.graph_type()), -1e6, 1e6,
        tg.conv_z.weight, tg.conv_r.weight, tg.conv_h.weight, tg.conv_z.bias, tg.conv_r.bias, tg.conv_h.bias,
        tg.linear_z.weight, tg.linear_z.bias, tg.linear_r.weight, tg.linear_r.bias, tg.linear_h.weight, tg.linear_h.bias,
        model.linear.weight, model.linear.bias, model.linear2.weight, model.linear2.bias)


class STGraphTGCN(torch.nn.Module):
    """benchmarking/static-temporal-tgcn/seastar/model.py:6-18."""

    def __init__(self, node_features, num_hidden_units, out_features, tgcn_cls=TGCN):
        super().__init__()
        self.temporal = tgcn_cls(node_features, num_hidden_units)
        self.linear = torch.nn.Linear(num_hidden_units, node_features)
        self.linear2 = torch.nn.Linear(node_features, out_features)

    def forward(self, g, node_feat, edge_weight, hidden_state):
        h = self.temporal(g, node_feat, edge_weight, hidden_state)
        y = F.relu(h)
        y = SF.linear(y, self.linear.weight, self.linear.bias)
        y_out = SF.linear(y, self.linear2.weight, self.linear2.bias)
        return y_out, y, h

    def step_loss(self, g, node_feat, edge_weight, hidden_state, target, cost=None):
        """``forward`` plus the training loop's ``cost = cost + torch.mean((y_out - target) ** 2)``: returns
        (cost, y, h) (``cost`` None: the loss alone).  With the fused head on (``set_fused_head``, default) relu,
        both Linears, the loss and that addition are one launch."""
        if _FUSED_HEAD:
            h = self.temporal(g, node_feat, edge_weight, hidden_state)
            y, _, loss = SF.tgcn_head(h, self.linear.weight, self.linear.bias, self.linear2.weight,
                                      self.linear2.bias, target, cost=cost if torch.is_tensor(cost) else None)
            return loss, y, h
        y_out, y, h = self(g, node_feat, edge_weight, hidden_state)
        loss = torch.mean((y_out - target) ** 2)
        return (loss if not torch.is_tensor(cost) else cost + loss), y, h


class GradBucket:
    """All parameter gradients as views into ONE contiguous fp32 buffer.

    ``zero()`` replaces ``optimizer.zero_grad()`` (which would drop the views);
    ``all_reduce_mean()`` issues the single collective of the data-parallel step.
    """

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        self.comm_seconds = 0.0
        self.comm_calls = 0

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def zero(self) -> None:
        self.flat.zero_()

    def check_views(self) -> None:
        base = self.flat.data_ptr()
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != base + off * self.flat.element_size():
                raise RuntimeError("a parameter's .grad no longer aliases the bucket "
                                   "(use bucket.zero(), not optimizer.zero_grad())")
            off += p.numel()

    def all_reduce_mean(self, world: int, group=None, timed: bool = False, divide: bool = True) -> None:
        """``divide=False``: only the sum -- the caller divides (the captured optimizer step does, in its graph)."""
        if world <= 1:
            return
        if timed and self.flat.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if divide:
            self.flat.div_(world)
        self.comm_calls += 1
        if timed and self.flat.is_cuda:
            e1.record()
            self._pending = getattr(self, "_pending", [])
            self._pending.append((e0, e1))

    def collect_comm_time(self) -> float:
        """Seconds spent in the timed all-reduces so far (synchronises the device)."""
        pend = getattr(self, "_pending", [])
        if pend:
            torch.cuda.synchronize()
            self.comm_seconds += sum(a.elapsed_time(b) for a, b in pend) * 1e-3
            self._pending = []
        return self.comm_seconds


def num_windows(total_timestamps: int, backprop_every: int) -> int:
    """static-temporal-tgcn/seastar/train.py:137-144."""
    if backprop_every == 0:
        backprop_every = total_timestamps
    return (total_timestamps + backprop_every - 1) // backprop_every


def windows_of_rank(total_timestamps: int, backprop_every: int, rank: int, world: int):
    """[(step, window index or None)]: window w runs on rank w mod R during optimizer step w // R.
    ``None`` marks a padding step where this rank only joins the all-reduce with zero gradients."""
    n = num_windows(total_timestamps, backprop_every)
    steps = (n + world - 1) // world
    out = []
    for s in range(steps):
        w = s * world + rank
        out.append((s, w if w < n else None))
    return out


_GENERATORS = {}


def _generator(device) -> torch.Generator:
    key = str(torch.device(device))
    gen = _GENERATORS.get(key)
    if gen is None:
        gen = _GENERATORS[key] = torch.Generator(device=device)
    return gen


def window_input(num_nodes: int, feat: int, epoch: int, window: int, device, seed: int = 0,
                 out: torch.Tensor | None = None) -> torch.Tensor:
    """The fresh ``torch.randn`` input of one BPTT window (static-temporal-tgcn/seastar/train.py:165-168), as a stream that
    is a function of (seed, epoch, window) ONLY: a window's input is the same whichever rank runs it and however many ranks
    there are, and a rank draws just the windows it owns (round 2 drew every window of the epoch on every rank in 64-window
    chunks: one launch, but a serial 256 MB term per rank at cfg4 that did not shrink with the number of ranks).
    ``out``: filled in place (a captured window's resident input slot)."""
    gen = _generator(device)
    gen.manual_seed((seed * 1_000_003 + epoch) * 1_000_003 + window)
    if out is None:
        return torch.randn(num_nodes, feat, device=device, generator=gen)
    return out.normal_(generator=gen)


def train_epoch_static(model, graph, edge_weight, targets, backprop_every: int, optimizer,
                       bucket: GradBucket, feat_size: int, epoch: int = 0, rank: int = 0,
                       world: int = 1, group=None, seed: int = 0, timed_comm: bool = False):
    """One epoch of the static-temporal loop; returns the list of this rank's window losses
    (device tensors, no host sync inside the loop)."""
    total = targets.shape[0]
    if backprop_every == 0:
        backprop_every = total
    n = graph.get_num_nodes()
    losses = []
    with _FoldGuard(optimizer, targets.device, world, group, owner=bucket) as guard:
        for _, w in windows_of_rank(total, backprop_every, rank, world):
            bucket.zero()
            if w is not None:
                y_hat = window_input(n, feat_size, epoch, w, targets.device, seed)
                cost = window_cost_of(model, graph, y_hat, edge_weight, targets[w * backprop_every:(w + 1) * backprop_every])
                cost = cost / (backprop_every + 1)
                cost.backward()
                losses.append(cost.detach())
            bucket.all_reduce_mean(world, group, timed_comm)
            optimizer.step()
    if guard.tripped():            # a window of this epoch met data the folded formulation refuses: the epoch again, without it
        guard.restore()
        _fold_switch_off("the epoch")
        return train_epoch_static(model, graph, edge_weight, targets, backprop_every, optimizer, bucket, feat_size, epoch, rank,
                                  world, group, seed, timed_comm)
    return losses


def window_cost_of(model, graph, x0, edge_weight, targets_window):
    """The window's accumulated cost: one fused autograd node (``window_cost``) where usable, else the reference's
    per-snapshot loop over ``model.step_loss``."""
    if window_cost_usable(model, graph, x0, edge_weight, targets_window):
        return window_cost(model, graph, x0, edge_weight, targets_window)
    cost, hidden, y_hat = 0, None, x0
    for t in range(targets_window.shape[0]):
        cost, y_hat, hidden = model.step_loss(graph, y_hat, edge_weight, hidden, targets_window[t], cost)
    return cost


class DynamicSTGraphTGCN(torch.nn.Module):
    """benchmarking/dynamic-temporal-tgcn/seastar/model.py:5-21 (link-prediction head)."""

    def __init__(self, node_features, num_hidden_units, tgcn_cls=TGCN):
        super().__init__()
        self.temporal = tgcn_cls(node_features, num_hidden_units)
        self.linear = torch.nn.Linear(num_hidden_units, node_features)

    def forward(self, g, node_feat, edge_weight, hidden_state):
        h = self.temporal(g, node_feat, edge_weight, hidden_state)
        y = F.relu(h)
        y = SF.linear(y, self.linear.weight, self.linear.bias)
        return y, h

    def decode(self, z, edge_label_index):
        return (z[edge_label_index[0]] * z[edge_label_index[1]]).sum(dim=-1)

    def step_loss(self, g, node_feat, edge_weight, hidden_state, edge_label_index, target, cost=None):
        """``forward`` + ``decode`` + the training loop's ``cost = cost + BCEWithLogitsLoss()(...)``: returns
        (cost, y, h) (``cost`` None: the loss alone).  With the fused head on (``set_fused_head``, default) everything
        after the TGCN cell is five launches forward + backward."""
        if _FUSED_HEAD:
            h = self.temporal(g, node_feat, edge_weight, hidden_state)
            y, loss = SF.link_head(h, self.linear.weight, self.linear.bias, edge_label_index, target,
                                   cost=cost if torch.is_tensor(cost) else None)
            return loss, y, h
        y, h = self(g, node_feat, edge_weight, hidden_state)
        out = self.decode(y, edge_label_index).view(-1)
        loss = F.binary_cross_entropy_with_logits(out, target)
        return (loss if not torch.is_tensor(cost) else cost + loss), y, h


class _TGCNDynWindow(torch.autograd.Function):
    """cost = sum_t BCEWithLogitsLoss()(decode(y_t, edges_t), targets_t) over the snapshots of one BPTT window of the
    dynamic-temporal loop (benchmarking/dynamic-temporal-tgcn/seastar/train.py:179-231 with model.py:5-21): every
    snapshot has its OWN graph (``steps[t]``: forward / backward CSR, norm), un-weighted GCN gates, link-prediction
    head.  One step launch each way per snapshot plus the decoder; the input gradient of snapshot t + 1 is aggregated
    over ITS backward CSR by snapshot t's backward launch."""

    @staticmethod
    def forward(ctx, x0, steps, use_nid, lo, hi, Wcz, Wcr, Wch, bcz, bcr, bch, Wz, bz, Wr, br, Wh, bh, W1, b1):
        from . import kernels
        dev = x0.device
        ctx.params = (Wcz, Wcr, Wch, bcz, bcr, bch, Wz, bz, Wr, br, Wh, bh, W1, b1)
        B, N = len(steps), int(x0.shape[0])
        C, Fin, Fh = int(Wz.shape[0]), int(x0.shape[1]), int(W1.shape[0])
        x0 = x0.contiguous()
        Wcat, WcatT, b3, WzT, WrT, WhT, W1T = kernels.tgcn_pack_weights(Wcz, Wcr, Wch, bcz, bcr, bch, Wz, Wr, Wh, W1)
        ctx.packed_T = (WzT, WrT, WhT, W1T)
        from_p = kernels.STEP_WGRAD_FROM_P                      # weight gradients from P: no x3, no da3 (_TGCNWindow)
        w_fold, b_fold, f_bound, ctx.w_fold_t = (kernels.tgcn_fold_weights(Wcz, Wcr, Wch, bcz, bcr, bch, Wz, bz, Wr, br, Wh, bh, with_bound=True)
                                   if kernels.STEP_FOLDED and from_p else (None, None, None, None))      # folded form: likewise
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
        P, X3 = new(B, N, Fin), (None if from_p else new(B, N, 3 * C))
        fold_status = kernels.step_fold_status_word(dev) if (from_p or w_fold is not None) else None
        Z, R, Ht, Hn, HR = (new(B, N, C) for _ in range(5))
        Y = new(B, N, Fh)
        mask = (kernels.step_ones_mask(N, dev).unsqueeze(0).expand(B, N, 12) if (from_p and w_fold is not None)
                else torch.empty(B, N, 12, dtype=torch.int32, device=dev))                        # as in _TGCNWindow.forward
        M = int(steps[0]["edges"].shape[1])
        if any(int(st["edges"].shape[1]) != M for st in steps):
            raise ValueError("all snapshots of a window must carry the same number of label edges")
        nparts = (M + 31) // 32
        logits, partial = new(B, M), new(B, nparts)
        Wz_, bz_, Wr_, br_, Wh_, bh_, W1_, b1_ = (t.contiguous() for t in (Wz, bz, Wr, br, Wh, bh, W1, b1))
        with torch.cuda.device(dev):
            for st in steps:
                st["nc_f"] = kernels._edge_gathered(st["fwd"], "norm", st["norm"], st["fwd"].column_indices)
                st["normv"] = st["norm"].reshape(-1)
        for t, st in enumerate(steps):
            f = st["fwd"]
            kernels.tgcn_step_fwd(N, C, Fin, Fh, 1, lo, hi, dev, row_offsets=f.row_offset, column_indices=f.column_indices,
                                  node_ids=None, norm_col_edge=st["nc_f"], ew_edge=None,        # vertex order: see _TGCNWindow
                                  norm=st["normv"], x=x0 if t == 0 else Y[t - 1], H=None if t == 0 else Hn[t - 1],
                                  WcatT=WcatT, b3=b3, Wz=Wz_, bz=bz_, Wr=Wr_, br=br_, Wh=Wh_, bh=bh_, W1=W1_, b1=b1_,
                                  P=P[t], x3=None if from_p else X3[t], Z=Z[t], R=R[t], Ht=Ht[t], Hn=Hn[t], HR=HR[t], y=Y[t],
                                  clamp_mask=mask[t], w_fold=w_fold, b_fold=b_fold,
                                  fold_bound=f_bound, fold_status=fold_status)
        if _fold_refused(fold_status):
            return _TGCNDynWindow.forward(ctx, x0, steps, use_nid, lo, hi, Wcz, Wcr, Wch, bcz, bcr, bch, Wz, bz, Wr, br, Wh, bh, W1, b1)
        # the decoder + loss of every snapshot behind the last step, in one launch: a snapshot's loss feeds nothing in the next one
        kernels.link_decode_fwd_window([Y[t] for t in range(B)], [st["edges"] for st in steps], [st["targets"] for st in steps],
                                       [logits[t] for t in range(B)], [partial[t] for t in range(B)])
        cost = kernels.partial_sums_loss(partial, B, nparts, 1.0 / M)
        ctx.save_for_backward(x0, Wcat, Wz_, Wr_, Wh_, W1_, P, X3, Z, R, Ht, Hn, HR, Y, mask, logits)
        ctx.steps, ctx.use_nid, ctx.clamp = steps, use_nid, (float(lo), float(hi))
        return cost.reshape(())

    @staticmethod
    def backward(ctx, g_cost):
        from . import kernels
        x0, Wcat, Wz, Wr, Wh, W1, P, X3, Z, R, Ht, Hn, HR, Y, mask, logits = ctx.saved_tensors
        steps, lo, hi = ctx.steps, *ctx.clamp
        dev = x0.device
        B, N, C = Z.shape
        Fin, Fh = int(x0.shape[1]), int(Y.shape[2])
        g = g_cost.reshape(1).contiguous().float()
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
        from_p = X3 is None
        wide_d = from_p and kernels.STEP_WGRAD_ZR_TOGETHER      # as in _TGCNWindow.backward
        if wide_d:
            D3 = new(B, N, 3 * C)
            dzl, drl, dhl, da3 = D3[:, :, :C], D3[:, :, C:2 * C], D3[:, :, 2 * C:], None
        else:
            dzl, drl, dhl, da3 = new(B, N, C), new(B, N, C), new(B, N, C), (None if from_p else new(B, N, 3 * C))
        dyt = new(B, N, Fh)
        dH, zbuf = new(2, N, C), new(2, N, Fin)
        WzT, WrT, WhT, W1T = ctx.packed_T
        want_dx0 = ctx.needs_input_grad[0]
        with torch.cuda.device(dev):
            for st in steps:
                st["nc_b"] = kernels._edge_gathered(st["bwd"], "norm", st["norm"], st["bwd"].column_indices)
        for t in range(B - 1, -1, -1):
            st, last = steps[t], t == B - 1
            nxt = None if last else steps[t + 1]                 # the gather of z_{t+1} runs over snapshot t + 1's backward CSR
            # the node side of the link loss's backward runs inside the step launch (stg_tgcn_step_bwd's link_* fields)
            row_ptr, other, eid = st["incidence"]
            kw = {}
            if nxt is not None:
                b = nxt["bwd"]
                kw = dict(row_offsets=b.row_offset, column_indices=b.column_indices,
                          node_ids=None, norm_col_edge=nxt["nc_b"], ew_edge=None,
                          norm=nxt["normv"], zn=zbuf[(t + 1) & 1], dHn=dH[(t + 1) & 1])
            kernels.tgcn_step_bwd(N, C, Fin, Fh, 1, lo, hi, dev, link_edges=int(logits[t].shape[0]), g_cost=g, link_row_ptr=row_ptr,
                                  link_other=other, link_eid=eid, link_y=Y[t], link_logits=logits[t], link_target=st["targets"],
                                  Z=Z[t], R=R[t], Ht=Ht[t],
                                  H=None if t == 0 else Hn[t - 1], Hn=Hn[t], clamp_mask=mask[t], WzT=WzT, WrT=WrT, WhT=WhT,
                                  Wcat=Wcat, W1T=W1T, dzl=dzl[t], drl=drl[t], dhl=dhl[t], da3=None if from_p else da3[t], dH=dH[t & 1],
                                  z=zbuf[t & 1] if (t > 0 or want_dx0 or from_p) else None, dyt=dyt[t],
                                  w_fold_t=ctx.w_fold_t if from_p else None, ld_d=3 * C if wide_d else 0, **kw)
        dx0 = None
        if want_dx0:
            s0 = steps[0]
            dx0 = kernels.gcn_agg(zbuf[0], s0["norm"], s0["norm"], s0["bwd"], use_node_ids=ctx.use_nid)
        rng = range(B)
        Hprev = [_zeros(N, C, dev)] + [Hn[t] for t in range(B - 1)]
        sinks = _grad_sinks(ctx.params, ctx.needs_input_grad[5:19])          # see _TGCNWindow.backward
        dst = (lambda i, j: dict(out=sinks[i], colsum_out=sinks[j])) if sinks else (lambda i, j: {})  # noqa: E731
        gate = lambda d, k, second, i: dict(  # noqa: E731
            As=[d[t] for t in rng], Bs=[X3[t][:, k * C:(k + 1) * C] for t in rng], M=C, N=2 * C, B2s=second, nsplit=C,
            b_op=kernels.GEMM_B_CLAMP, lo=lo, hi=hi, colsum=True, **dst(i, i + 1))
        # five split-K contractions, ONE reduction launch (kernels.gemm_tn_form_batch); conv gradients as in _TGCNWindow.backward
        head_call = dict(As=[dyt[t] for t in rng], Bs=[Hn[t] for t in rng], M=Fh, N=C, b_op=kernels.GEMM_B_RELU, colsum=True, **dst(12, 13))
        if from_p:                                               # as in _TGCNWindow.backward
            Wcz, Wcr, Wch, bcz, bcr, bch, Wz_p, _, Wr_p, _, Wh_p, _ = ctx.params[:12]
            Rg, csg, nres = _gate_contractions(kernels, D3 if wide_d else None, (dzl, drl, dhl), Hprev, [HR[t] for t in rng],
                                               [P[t] for t in rng], rng, C, Fin, [head_call])
            gates = kernels.tgcn_unfold_gate_grads(
                Rg, csg, (Wcz, Wcr, Wch), (bcz, bcr, bch), (Wz_p, Wr_p, Wh_p),
                outs=[(sinks[6 + 2 * k], sinks[7 + 2 * k], sinks[k], sinks[3 + k]) for k in range(3)] if sinks else None)
            ctx.steps = None
            if sinks:
                return (dx0,) + (None,) * 18
            dW1, db1 = nres[0]
            return (dx0, None, None, None, None, *[g[2] for g in gates], *[g[3] for g in gates],
                    gates[0][0], gates[0][1], gates[1][0], gates[1][1], gates[2][0], gates[2][1], dW1, db1)
        conv = dict(As=[da3[t] for t in rng], Bs=[P[t] for t in rng], M=3 * C, N=Fin, colsum=True)
        if sinks:
            conv.update(out_blocks_t=list(sinks[0:3]), colsum_blocks=list(sinks[3:6]))
        res = kernels.gemm_tn_form_batch([
            gate(dzl, 0, Hprev, 6), gate(drl, 1, Hprev, 8), gate(dhl, 2, [HR[t] for t in rng], 10), conv, head_call])
        ctx.steps = None
        if sinks:
            return (dx0,) + (None,) * 18
        (dWz, dbz), (dWr, dbr), (dWh, dbh), (dWcT, db3), (dW1, db1) = res
        conv_w = [dWcT[k * C:(k + 1) * C].t() for k in range(3)]
        conv_b = [db3[k * C:(k + 1) * C] for k in range(3)]
        return (dx0, None, None, None, None, *conv_w, *conv_b, dWz, dbz, dWr, dbr, dWh, dbh, dW1, db1)


def dyn_window_usable(model, graph, x0) -> bool:
    from . import kernels
    from .nn.pytorch.static.gcn_conv import GCNConv
    if not (_FUSED_WINDOW and _FUSED_HEAD and isinstance(model, DynamicSTGraphTGCN) and type(model.temporal) is TGCN):
        return False
    tg = model.temporal
    convs = (tg.conv_z, tg.conv_r, tg.conv_h)
    return (x0.is_cuda and x0.dtype == torch.float32 and x0.dim() == 2 and hasattr(graph, "csr")
            and not kernels.reference_compat() and kernels._EDGE_CACHE
            and all(type(c) is GCNConv and c.bias is not None and c.activation is None for c in convs)
            and model.linear.bias is not None
            and kernels.tgcn_step_supported(tg.out_channels, tg.in_channels, model.linear.out_features)
            and model.linear.out_features == tg.in_channels and x0.shape[0] * 3 * tg.out_channels < (1 << 30))


def dyn_window_cost(model, graph, x0, steps) -> torch.Tensor:
    """``sum_t BCEWithLogitsLoss()(decode(y_t), targets_t)`` over ``steps`` (dicts with fwd, bwd, norm, edges, targets,
    incidence: one per snapshot, collected while the loop moved the graph forward)."""
    from . import kernels
    tg = model.temporal
    return _TGCNDynWindow.apply(
        x0, steps, kernels.rows_by_node_ids(graph.graph_type()), -1e6, 1e6,
        tg.conv_z.weight, tg.conv_r.weight, tg.conv_h.weight, tg.conv_z.bias, tg.conv_r.bias, tg.conv_h.bias,
        tg.linear_z.weight, tg.linear_z.bias, tg.linear_r.weight, tg.linear_r.bias, tg.linear_h.weight, tg.linear_h.bias,
        model.linear.weight, model.linear.bias)


def train_epoch_dynamic(model, graph, pos_neg_edges, pos_neg_targets, backprop_every: int, optimizer,
                        bucket: GradBucket, feat_size: int, epoch: int = 0, rank: int = 0, world: int = 1,
                        group=None, seed: int = 0, norm_fn=None, timed_comm: bool = False):
    """One epoch of the dynamic-temporal loop (dynamic-temporal-tgcn/seastar/train.py:179-231):
    per step ``graph.get_graph(t)``, ``norm`` from the snapshot's in-degrees unless already cached
    for that timestamp, un-weighted GCN kernels, link-prediction loss on ``pos_neg_edges[t]``;
    the last timestamp has no prediction target (``t >= T - 1`` stops).  Backward walks the
    snapshots in reverse through the executor's timestamp stack."""
    total = len(pos_neg_edges)
    if backprop_every == 0:
        backprop_every = total
    norm_fn = norm_fn or in_degree_norm
    n = graph.get_num_nodes()
    dev = pos_neg_targets[0].device
    losses = []
    graph.reset_graph()
    with _FoldGuard(optimizer, dev, world, group, owner=bucket) as guard:
        for _, w in windows_of_rank(total, backprop_every, rank, world):
            bucket.zero()
            if w is not None:
                cost = 0
                hidden = None
                y_hat = window_input(n, feat_size, epoch, w, dev, seed)
                graph.get_graph(w * backprop_every)
                ts = range(w * backprop_every, min((w + 1) * backprop_every, total - 1))
                fused = dyn_window_usable(model, graph, y_hat) and len(ts) > 0 and all(
                    pos_neg_edges[t].dtype == torch.int64 and pos_neg_edges[t].is_contiguous() and pos_neg_edges[t].dim() == 2
                    and pos_neg_edges[t].shape == pos_neg_edges[ts[0]].shape and pos_neg_edges[t].shape[1] > 0
                    and pos_neg_targets[t].dtype == torch.float32 and pos_neg_targets[t].is_contiguous()
                    and pos_neg_targets[t].numel() == pos_neg_edges[t].shape[1] for t in ts)
                steps = []
                for k in range(backprop_every):
                    t = w * backprop_every + k
                    if t >= total - 1:
                        break
                    graph.get_graph(t)
                    if graph.get_ndata("norm") is None:
                        graph.set_ndata("norm", norm_fn(graph))
                    if fused:                                   # one autograd node for the window: collect the snapshots
                        steps.append(dict(fwd=graph.csr("fwd"), bwd=graph.csr("bwd"), norm=graph.get_ndata("norm"),
                                          edges=pos_neg_edges[t], targets=pos_neg_targets[t],
                                          incidence=SF._incidence_of(pos_neg_edges[t], n)))
                    else:
                        cost, y_hat, hidden = model.step_loss(graph, y_hat, None, hidden, pos_neg_edges[t], pos_neg_targets[t], cost)
                if fused and steps:
                    cost = dyn_window_cost(model, graph, y_hat, steps)
                if not isinstance(cost, int):
                    cost = cost / (backprop_every + 1)
                    cost.backward()
                    losses.append(cost.detach())
            bucket.all_reduce_mean(world, group, timed_comm)
            optimizer.step()
    if guard.tripped():            # see train_epoch_static
        guard.restore()
        _fold_switch_off("the epoch")
        return train_epoch_dynamic(model, graph, pos_neg_edges, pos_neg_targets, backprop_every, optimizer, bucket, feat_size, epoch,
                                   rank, world, group, seed, norm_fn, timed_comm)
    return losses


def _capture_tail_graph(tail, in_graph, dev, bucket, group, what: str):
    """Capture ``tail()`` into a HIP graph.  ``in_graph[0]``: the tail contains the gradient all-reduce; if the communicator
    refuses stream capture the failure is LOGGED (warnings), ``in_graph[0]`` is cleared, the communicator and the stream are
    checked with one eager all-reduce of the bucket (it must come back: a broken communicator raises here, not steps later), and
    the tail is captured again without the collective."""
    import warnings
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    try:
        with _graph_capture(g):
            tail()
    except RuntimeError as err:
        if not in_graph[0]:
            raise
        warnings.warn(f"{what}: the process group refused to capture the gradient all-reduce into the optimizer-tail graph "
                      f"({type(err).__name__}: {str(err).splitlines()[0][:200]}); keeping the collective eager between the replays")
        in_graph[0] = False
        torch.cuda.synchronize(dev)
        probe = bucket.flat.clone()
        dist.all_reduce(probe, op=dist.ReduceOp.SUM, group=group)            # health check of communicator + stream
        torch.cuda.synchronize(dev)
        if not bool(torch.isfinite(probe).all()):
            raise RuntimeError(f"{what}: the eager all-reduce after the failed capture returned non-finite values") from err
        g = torch.cuda.CUDAGraph()
        with _graph_capture(g):
            tail()
    return g


class CapturedStaticWindow:
    """The compute of one full BPTT window of the static-temporal loop -- bucket.zero, ``backprop_every`` model steps,
    loss, backward through time -- captured ONCE into a HIP graph and replayed per window.

    Eagerly the window issues thousands of small kernels and is bound by host launch overhead; the graph replays the
    identical kernel sequence from device-side descriptors.  What changes per window is READ BY THE GRAPH from device
    memory: a window index ``widx`` selects the window's slice of ``targets`` and ``slot`` (= widx // world) its ``randn``
    input among this rank's pre-drawn inputs of the epoch (captured gathers), and the cost lands in slot ``widx`` of a
    per-epoch buffer.  Per optimizer step the host therefore issues: one graph replay, the all-reduce of the gradient
    bucket (eager by default: it keeps the N > 1 path free of stream-capture constraints on the communicator; skipped at
    one rank; ``allreduce_in_graph`` moves it into the second graph), and a second small graph holding ``grad /= world``,
    the optimizer step, ``widx += world`` and ``slot += 1`` when the optimizer is capturable
    (``torch.optim.Adam(..., capturable=True)``); with another optimizer that tail runs eagerly as before.
    """

    def __init__(self, model, graph, edge_weight, targets, backprop_every: int, optimizer,
                 bucket: GradBucket, feat_size: int, world: int = 1, rank: int = 0, group=None, warmup: int = 3,
                 allreduce_in_graph: bool = False):
        """``allreduce_in_graph`` (N > 1, RCCL only; default off): capture the gradient all-reduce INTO the second graph
        (all-reduce, / N, Adam, window index) instead of issuing it eagerly between the two replays -- one host operation
        fewer per optimizer step.  Falls back to the eager all-reduce if the communicator refuses stream capture."""
        self.B = B = backprop_every
        self.allreduce_in_graph = False
        # (a one-rank RCCL group passed explicitly also takes the captured collective: how the single-GPU test drives it)
        self._want_allreduce_in_graph = bool(allreduce_in_graph) and (world > 1 or group is not None)
        n = graph.get_num_nodes()
        dev = targets.device
        total = targets.shape[0]
        self.full_windows = total // B                      # windows replayed from the graph; a ragged tail runs eagerly
        self.bucket, self.world, self.rank, self.group, self.optimizer = bucket, world, rank, group, optimizer
        self.n, self.feat, self.dev = n, feat_size, dev
        self.num_windows = num_windows(total, B)
        self.my_windows = list(range(rank, self.num_windows, world))                   # window w runs on rank w mod R
        if len(self.my_windows) * n * feat_size * 4 > (8 << 30):
            raise ValueError("CapturedStaticWindow keeps its rank's window inputs of an epoch resident: more than 8 GiB here")
        # one resident slot per window of THIS rank (slot = w // world), refilled in place every epoch
        self.inputs = torch.zeros(max(len(self.my_windows), 1), n, feat_size, device=dev)
        # targets are sharded by owning rank: this object keeps the snapshots of ITS windows only -- slot
        # i = the i-th window of this rank, like the inputs -- so the caller may drop the full [T, N] tensor afterwards
        self.total = int(total)
        mine_full = [w for w in self.my_windows if w < self.full_windows]
        if world == 1:
            self.targets_w = targets[: self.full_windows * B].view(self.full_windows, B, *targets.shape[1:])   # a view: no copy
        else:
            self.targets_w = (torch.stack([targets[w * B:(w + 1) * B] for w in mine_full]) if mine_full
                              else targets.new_zeros((1, B) + tuple(targets.shape[1:])))
        tail_w = self.num_windows - 1
        self.targets_tail = (targets[tail_w * B:total].clone()
                             if self.num_windows > self.full_windows and tail_w in self.my_windows else None)
        self.widx = torch.zeros(1, dtype=torch.int64, device=dev)         # the window the next replay runs (global index)
        self.slot = torch.zeros(1, dtype=torch.int64, device=dev)         # its input slot = widx // world
        self.costs = torch.zeros(max(self.num_windows, 1), device=dev)
        self._epoch = None

        def body():
            bucket.zero()
            y0 = self.inputs.index_select(0, self.slot)[0]                            # captured gathers: no host copy
            tw = self.targets_w.index_select(0, self.slot)[0]             # slot == widx at one rank
            cost = window_cost_of(model, graph, y0, edge_weight, tw)
            cost = cost / (B + 1)
            with direct_param_grads():                       # the bucket was zeroed above: written, not accumulated
                cost.backward()
            self.costs.index_copy_(0, self.widx, cost.detach().reshape(1))

        self._body, self._warmup, self.graph = body, warmup, None
        self.recapture()

        # second graph: what follows the all-reduce
        self.step_graph = None
        if all(g.get("capturable", False) for g in optimizer.param_groups):
            self._capture_step(dev)

    def recapture(self) -> None:
        """(Re-)capture the window graph in the step formulation that is current NOW (kernels.STEP_FOLDED / STEP_WGRAD_FROM_P): at
        construction, and again by ``train_epoch_static_captured`` after an epoch met data the folded formulation refuses."""
        dev = self.dev
        self.graph = None                           # the old graph's pool goes back before the new one is built
        # the body gathers its input / target slot and scatters its cost through these device counters: after an epoch they point
        # one past the last window (an out-of-range gather is a device assertion) -- warm up and capture on window 0
        self.widx.zero_()
        self.slot.zero_()
        # Warm up on a side stream (allocator, lazily built per-edge caches, tracing).  The body
        # only writes gradients, so warming up and capturing leave the training state untouched.
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(self._warmup):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with _graph_capture(graph):
            self._body()
        self.graph = graph
        self.bucket.zero()

    def _capture_step(self, dev):
        """``grad /= world`` + ``optimizer.step()`` + ``widx += world`` as one graph.  Capturing runs the optimizer
        once, so its state and the parameters are snapshotted and restored around the capture (the first captured
        step would otherwise count as a training step on zero gradients)."""
        opt, bucket, world = self.optimizer, self.bucket, self.world
        params = [p for g in opt.param_groups for p in g["params"]]

        in_graph = [self._want_allreduce_in_graph]

        def tail():
            if in_graph[0]:
                dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM, group=self.group)
            if world > 1:
                bucket.flat.div_(world)
            opt.step()
            self.widx.add_(world)
            self.slot.add_(1)
        saved_p = [p.detach().clone() for p in params]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                tail()                                        # creates the optimizer state (device step counters)
        torch.cuda.current_stream(dev).wait_stream(side)
        g = _capture_tail_graph(tail, in_graph, dev, bucket, self.group, "CapturedStaticWindow")
        self.allreduce_in_graph = in_graph[0]
        with torch.no_grad():
            for p, q in zip(params, saved_p):
                p.copy_(q)
            for st in opt.state.values():                     # back to "no step taken yet"
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
            self.widx.zero_()
            self.slot.zero_()
        bucket.zero()
        self.step_graph = g

    def begin_epoch(self, epoch: int, seed: int = 0) -> None:
        """Draw THIS RANK's window inputs of the epoch in place and point ``widx`` / ``slot`` at its first window."""
        for i, w in enumerate(self.my_windows):
            window_input(self.n, self.feat, epoch, w, self.dev, seed, out=self.inputs[i])
        self.widx.fill_(self.rank)
        self.slot.zero_()
        self._epoch = epoch

    def run(self, window: int, timed_comm: bool = False) -> torch.Tensor:
        """Replay the window ``widx`` points at (== ``window``: the caller's loop and the device counter advance
        together), reduce, step.  Returns the slot of ``costs`` the window's cost is written to (a view: no copy)."""
        self.graph.replay()
        if self.step_graph is not None:
            if not self.allreduce_in_graph:
                self.bucket.all_reduce_mean(self.world, self.group, timed_comm, divide=False)
            self.step_graph.replay()
        else:
            self.bucket.all_reduce_mean(self.world, self.group, timed_comm)
            self.optimizer.step()
            self.widx.add_(self.world)
            self.slot.add_(1)
        return self.costs[window]


def train_epoch_static_captured(cw: CapturedStaticWindow, model, graph, edge_weight, targets, optimizer,
                                bucket: GradBucket, feat_size: int, epoch: int = 0, rank: int = 0,
                                world: int = 1, group=None, seed: int = 0, timed_comm: bool = False):
    """``train_epoch_static`` with full windows replayed from the captured graph; a trailing short
    window or a padding step (more ranks than windows left) runs eagerly.  Returns this rank's window costs as one
    tensor (views into the captured window's per-epoch buffer, cloned once at the end)."""
    total = cw.total                                     # (``targets`` may be None: the window object holds this rank's share)
    B = cw.B
    slots, eager = [], {}
    with _FoldGuard(optimizer, cw.dev, world, group, owner=cw) as guard:
        cw.begin_epoch(epoch, seed)
        for _, w in windows_of_rank(total, B, rank, world):
            if w is not None and w < cw.full_windows:
                cw.run(w, timed_comm)
                slots.append(w)
                continue
            bucket.zero()
            if w is not None:
                y_hat = cw.inputs[w // world]
                tw = cw.targets_tail if cw.targets_tail is not None else targets[w * B:min((w + 1) * B, total)]
                cost = window_cost_of(model, graph, y_hat, edge_weight, tw)
                cost = cost / (B + 1)
                cost.backward()
                eager[w] = cost.detach()
                slots.append(w)
            bucket.all_reduce_mean(world, group)
            optimizer.step()
            cw.widx.add_(world)
            cw.slot.add_(1)
    if guard.tripped():
        # a replayed window met data the folded formulation refuses (nothing can be read back inside a graph): every optimizer
        # step of this epoch is undone, the window graph is captured again in the reference formulation and the epoch rerun
        guard.restore()
        _fold_switch_off("the epoch (its window graph captured again)")
        cw.recapture()
        return train_epoch_static_captured(cw, model, graph, edge_weight, targets, optimizer, bucket, feat_size, epoch, rank,
                                           world, group, seed, timed_comm)
    out = cw.costs[slots].clone() if slots else cw.costs[:0].clone()
    for i, w in enumerate(slots):
        if w in eager:
            out[i] = eager[w]
    return list(out.unbind(0))


class CapturedDynamicWindows:
    """The dynamic-temporal loop with ONE HIP graph per BPTT window.

    A window of the dynamic loop differs from the next in its graphs, not in its shapes, so -- unlike the static
    case -- there is one captured graph per window rather than one for all: window ``w``'s graph holds bucket.zero, the
    move through its snapshots (for ``NaiveGraph(resident=False)`` the device CSR builds themselves, which therefore
    still run every epoch; for a resident NaiveGraph only pointer swaps, which cost nothing at replay), the in-degree
    norms, ``dyn_window_cost`` and the backward pass.  It is captured the first time the window is met -- AFTER at least
    one eager epoch (lazy initialisations, the label edges' incidence lists) -- and replayed from then on; per
    optimizer step the host issues one input copy, one replay, the all-reduce and the captured optimizer tail.
    Eagerly the window is ~ 60 launches per snapshot from Python and leaves the device idle 20-30 % of the time.

    The delta-based stores (PCSRGraph, GPMAGraph) qualify too (round 3): their update batches are packed and sorted at
    construction, every step of the graph protocol (merge, CSR emission in both orientations, degrees) is a device launch
    with shapes known on the host, and nothing synchronises -- so ``get_graph(t)`` itself is captured.  A window's graph
    then starts from the edge set the PREVIOUS window of this rank left (tensors of that window's pool, rewritten by its
    replay) or from the persistent base set after ``reset_graph()``; the host half of the protocol (current timestamp,
    which store object is current) is restored after every replay from what the capture recorded.  Windows must be
    replayed in increasing order within an epoch, as ``train_epoch_dynamic_captured`` does.
    Memory: a window's activations and (rebuild mode, stores) CSRs live in its graph's pool (~ 2 GB per window of 20
    snapshots at |V| = 25 K)."""

    def __init__(self, model, graph, pos_neg_edges, pos_neg_targets, backprop_every: int, optimizer,
                 bucket: GradBucket, feat_size: int, world: int = 1, rank: int = 0, group=None, norm_fn=None,
                 allreduce_in_graph: bool = False):
        """``allreduce_in_graph``: as for :class:`CapturedStaticWindow` -- the gradient all-reduce captured into the
        optimizer-tail graph (N > 1, or an explicit one-rank RCCL group); logged fallback to the eager collective."""
        from .graph.dynamic.dynamic_graph import DynamicGraph
        from .graph.dynamic.naive.naive_graph import NaiveGraph
        if not (isinstance(graph, DynamicGraph) and hasattr(graph, "csr")):
            raise TypeError("CapturedDynamicWindows needs a dynamic graph that hands out device CSRs (NaiveGraph, PCSRGraph, GPMAGraph)")
        self._store = not isinstance(graph, NaiveGraph)
        if not self._store and not graph._resident:
            from . import kernels
            kernels.pin_build_counters(self, pos_neg_targets[0].device, graph.get_num_nodes())   # rebuilds inside the window graphs
        self._end_state = {}
        self._side = None
        # rebuild mode: snapshot builds on parallel branches of the window's graph.  Off: measured 30.2 against 31.7 epochs/s
        # at cfg5 (profiles/r03 notes in DESIGN.md) -- hipGraphLaunch did not overlap the branches' 5-launch chains
        self.parallel_builds = False
        # rebuild mode: the window's snapshots as ONE batched build (stg_graph_build_direct2_batch_device)
        self.batched_builds = True
        self.deferred_emission = True
        # rebuild mode, OPTION (off): a window's snapshot builds as a HIP graph of their own, replayed on a SECOND STREAM one window
        # ahead of the training graph that reads them.  The builds read the edge lists only, so the builds of this rank's next window
        # (the first window of the next epoch after the last) run beside the current window's step launches; every snapshot is
        # still built once per epoch.  Bit-identical results (tests/test_gpu_window.py).  MEASURED at cfg5 (round 5): 80 % of the
        # builds' device time does run beside another launch (profiles/r05_dyn_build_overlap.json), but a step launch wants every
        # register of a CU for its workgroup and waits for whatever build waves sit there -- forward step launches beside a build take
        # 55 us against 24 alone, whatever the build stream's priority -- so the epoch gains between -1.5 and +2.5 %
        # (bench.py, dynamic.rebuild_prefetch): not a default.
        self.prefetch_builds = False
        self.build_stream_low_priority = True
        self._build_graphs, self._built, self._build_done, self._train_done, self._build_pending = {}, {}, {}, {}, {}
        self._build_stream = None
        self._order = None
        self.model, self.graph, self.edges, self.targets = model, graph, pos_neg_edges, pos_neg_targets
        self.total = len(pos_neg_edges)
        self.B = backprop_every or self.total
        self.optimizer, self.bucket, self.feat = optimizer, bucket, feat_size
        self.world, self.rank, self.group = world, rank, group
        self.norm_fn = norm_fn or in_degree_norm
        self.n = graph.get_num_nodes()
        self.dev = pos_neg_targets[0].device
        self.graphs, self.inputs, self.costs = {}, {}, {}
        self.step_graph = None
        self._tail_ready = False
        self._probe = None
        self.allreduce_in_graph = False
        self._want_allreduce_in_graph = bool(allreduce_in_graph) and (world > 1 or group is not None)

    def __del__(self):
        # a build replayed ahead may still be writing into its graph's pool when the last reference goes
        st = getattr(self, "_build_stream", None)
        if st is not None:
            try:
                st.synchronize()
            except Exception:                                    # interpreter shutdown
                pass

    def invalidate(self) -> None:
        """Drop every captured window graph (they are captured again the next time their window is met): after a change of the
        step formulation."""
        if self._build_graphs:
            torch.cuda.synchronize(self.dev)                     # a build replay may still be running on the second stream
        self._build_graphs.clear()
        self._built.clear()
        self._build_done.clear()
        self._train_done.clear()
        self._build_pending.clear()
        self.graphs.clear()
        self.costs.clear()
        self.inputs.clear()
        self._end_state.clear()
        if not self._store and not self.graph._resident:
            self.graph._snapshots.clear()            # tensors of the dropped graphs' pools
        self.graph._ndata.clear()

    def timestamps(self, w: int):
        return range(w * self.B, min((w + 1) * self.B, self.total - 1))

    def usable(self, w: int) -> bool:
        ts = self.timestamps(w)
        if len(ts) == 0:
            return False
        e, t0 = self.edges, ts[0]
        if self._probe is None:
            self._probe = torch.empty(self.n, self.feat, device=self.dev)      # shape / dtype / device stand-in for a window input
        return dyn_window_usable(self.model, self.graph, self._probe) and all(
            e[t].dtype == torch.int64 and e[t].is_contiguous() and e[t].dim() == 2 and e[t].shape == e[t0].shape
            and e[t].shape[1] > 0 and self.targets[t].dtype == torch.float32 and self.targets[t].is_contiguous()
            and self.targets[t].numel() == e[t].shape[1] for t in ts)

    def _prebuild(self, w: int) -> None:
        """Rebuild mode: the window's snapshot builds are independent of each other and of the model, and each is a
        chain of five small launches that leaves most of the chip idle -- so they are issued round-robin on
        ``kernels.BUILD_SLOTS`` side streams (parallel branches of the window's HIP graph, one counter buffer each) and
        joined before the first step.  Bit-identical CSRs; ~ 4 builds in flight instead of 1."""
        from . import kernels
        g = self.graph
        cur = torch.cuda.current_stream(self.dev)
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.dev) for _ in range(min(4, kernels.BUILD_SLOTS))]
        used = []
        for i, t in enumerate(self.timestamps(w)):
            if t in g._snapshots or g._built_by.get(t) != "direct":
                continue
            k = i % len(self._side)
            st = self._side[k]
            if st not in used:
                st.wait_stream(cur)
                used.append(st)
            with torch.cuda.stream(st):
                g._snapshot(t, counters_slot=k)
        for st in used:
            cur.wait_stream(st)

    def _build_body(self, w: int) -> None:
        """Rebuild mode, ``prefetch_builds``: every snapshot of window ``w`` (both CSRs, degrees, norms, per-edge coefficients)."""
        from . import kernels
        g = self.graph
        base = kernels._C.BUILD_BATCH_MAX                        # counter slots of their own: beside eager builds on another stream
        while g.prebuild(self.timestamps(w), counters_base=base):
            pass
        for t in self.timestamps(w):
            g._snapshot(t, counters_slot=base)                   # what the batched build does not cover

    def _next_window(self, w: int):
        """This rank's window after ``w`` (cyclically: the first window of the next epoch after the last)."""
        if self._order is None:
            self._order = [v for _, v in windows_of_rank(self.total, self.B, self.rank, self.world) if v is not None and self.usable(v)]
        if w not in self._order or len(self._order) < 2:
            return None
        return self._order[(self._order.index(w) + 1) % len(self._order)]

    def _launch_build(self, w: int) -> None:
        if self._build_stream is None:
            # LOWEST priority: a step launch wants every register of a CU for its workgroup, and waits for whatever build waves sit
            # there.  Measured with equal priorities: forward step launches beside a build 55 us against 24 alone (the builds'
            # thousands of short workgroups keep taking the slots a step workgroup waits for); the builds are to fill gaps only.
            least = torch.cuda.Stream.priority_range()[0]
            self._build_stream = torch.cuda.Stream(device=self.dev, priority=least if self.build_stream_low_priority else 0)
        st = self._build_stream
        if w in self._train_done:
            st.wait_event(self._train_done[w])                   # the previous reader of these buffers
        with torch.cuda.stream(st):
            self._build_graphs[w].replay()
            ev = self._build_done.get(w)
            if ev is None:
                ev = self._build_done[w] = torch.cuda.Event()
            ev.record(st)
        self._build_pending[w] = True

    def _body(self, w: int) -> torch.Tensor:
        from .nn import functional as SF
        g = self.graph
        self.bucket.zero()
        if not self._store and not g._resident and w not in self._build_graphs:
            if self.parallel_builds:
                self._prebuild(w)
            elif self.batched_builds:
                while g.prebuild(self.timestamps(w)):        # the window's snapshots in the launches of one build
                    pass                                     # (a window longer than STG_BUILD_BATCH_MAX: of two, ...)
        steps = []
        import contextlib
        # delta stores: nothing reads a snapshot's columns before the window's first model step, so the emissions ride in the
        # next timestamp's merge launch (one launch per timestamp)
        defer = g.deferred_emission() if self._store and self.deferred_emission and hasattr(g, "deferred_emission") else contextlib.nullcontext()
        with defer:
            for t in self.timestamps(w):
                g.get_graph(t)
                steps.append(dict(fwd=g.csr("fwd"), bwd=g.csr("bwd"), norm=self.norm_fn(g), edges=self.edges[t],
                                  targets=self.targets[t], incidence=SF._incidence_of(self.edges[t], self.n)))
        cost = dyn_window_cost(self.model, g, self.inputs[w], steps) / (self.B + 1)
        with direct_param_grads():                           # the bucket was zeroed above: written, not accumulated
            cost.backward()
        return cost.detach()

    def _capture(self, w: int) -> None:
        import copy
        g = self.graph
        self.inputs[w] = torch.zeros(self.n, self.feat, device=self.dev)
        rebuild = not self._store and not g._resident
        if rebuild:
            for t in self.timestamps(w):         # the builds must be IN the graph: forget snapshots an eager epoch left
                g._snapshots.pop(t, None)
        if self._store:
            g._ndata.clear()                     # per-timestamp norms of an eager epoch: recomputed inside the graph
        torch.cuda.synchronize(self.dev)
        fits = g._max_cached is None or g._max_cached >= len(self.timestamps(w)) if rebuild else False
        if rebuild and self.prefetch_builds and not self.parallel_builds and fits:      # (all of the window's snapshots alive at once)
            cb = torch.cuda.CUDAGraph()
            with _graph_capture(cb):
                self._build_body(w)
            # the snapshots are tensors of the build graph's pool: kept (the training graph below reads them on every replay)
            self._built[w] = {t: g._snapshots[t] for t in self.timestamps(w)}
            self._build_graphs[w] = cb
        cg = torch.cuda.CUDAGraph()
        with _graph_capture(cg):
            self.costs[w] = self._body(w)
        if rebuild:
            for t in self.timestamps(w):         # tensors of the graph's private pool: not for eager readers
                g._snapshots.pop(t, None)
        if self._store:                          # where the walk ended: restored on the host after every replay
            self._end_state[w] = (copy.copy(g._forward_graph), g.current_timestamp)
        self.graphs[w] = cg
        self.bucket.zero()

    def _capture_tail(self) -> None:
        """``grad /= world`` + ``optimizer.step()`` as one graph (capturable optimizers; the optimizer has taken real
        steps in the eager epoch before, so its state exists and capturing does not run it)."""
        self._tail_ready = True
        opt, bucket, world = self.optimizer, self.bucket, self.world
        if not all(g.get("capturable", False) for g in opt.param_groups) or not opt.state:
            return

        in_graph = [self._want_allreduce_in_graph]

        def tail():
            if in_graph[0]:
                dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM, group=self.group)
            if world > 1:
                bucket.flat.div_(world)
            opt.step()
        # capturing RUNS nothing, but a refused collective capture falls back to an eager health check: keep the gradients
        saved = bucket.flat.clone()
        self.step_graph = _capture_tail_graph(tail, in_graph, self.dev, bucket, self.group, "CapturedDynamicWindows")
        bucket.flat.copy_(saved)
        self.allreduce_in_graph = in_graph[0]

    def run(self, w: int, epoch: int, seed: int = 0, timed_comm: bool = False) -> torch.Tensor:
        if not self._tail_ready:
            self._capture_tail()
        if w not in self.graphs:
            self._capture(w)
        self.inputs[w].copy_(window_input(self.n, self.feat, epoch, w, self.dev, seed))
        if w in self._build_graphs:
            cur = torch.cuda.current_stream(self.dev)
            if not self._build_pending.get(w):                   # nobody built it ahead (first use, one-window rank)
                self._launch_build(w)
            cur.wait_event(self._build_done[w])
            self._build_pending[w] = False
            nxt = self._next_window(w)
            if nxt is not None and nxt in self._build_graphs and not self._build_pending.get(nxt):
                self._launch_build(nxt)                          # beside this window's training graph
            self.graphs[w].replay()
            ev = self._train_done.get(w)
            if ev is None:
                ev = self._train_done[w] = torch.cuda.Event()
            ev.record(cur)
        else:
            self.graphs[w].replay()
        if self._store:
            import copy
            fg, ts = self._end_state[w]
            g = self.graph
            g._forward_graph, g.current_timestamp, g._is_backprop_state = copy.copy(fg), ts, False
            g._get_graph_csr_ptrs()
        if self.step_graph is not None:
            if not self.allreduce_in_graph:
                self.bucket.all_reduce_mean(self.world, self.group, timed_comm, divide=False)
            self.step_graph.replay()
        else:
            self.bucket.all_reduce_mean(self.world, self.group, timed_comm)
            self.optimizer.step()
        return self.costs[w]


def train_epoch_dynamic_captured(cd: CapturedDynamicWindows, epoch: int = 0, seed: int = 0, timed_comm: bool = False):
    """``train_epoch_dynamic`` with every usable window replayed from its own HIP graph (call after at least one eager
    epoch on the same objects).  Returns this rank's window costs (views of the graphs' output buffers: valid until
    the window is replayed again)."""
    losses = []
    cd.graph.reset_graph()
    with _FoldGuard(cd.optimizer, cd.dev, cd.world, cd.group, owner=cd) as guard:
        for _, w in windows_of_rank(cd.total, cd.B, cd.rank, cd.world):
            if w is not None and cd.usable(w):
                losses.append(cd.run(w, epoch, seed, timed_comm))
                continue
            cd.bucket.zero()
            if w is not None and len(cd.timestamps(w)) > 0:
                raise RuntimeError("CapturedDynamicWindows: window %d is not covered by the fused window path" % w)
            cd.bucket.all_reduce_mean(cd.world, cd.group, timed_comm)
            cd.optimizer.step()
    if guard.tripped():
        # as train_epoch_static_captured; the per-window graphs are dropped (they are captured again, lazily, by the next
        # captured epoch) and THIS epoch is rerun eagerly -- the eager epoch the lazy captures expect to have come before them
        guard.restore()
        _fold_switch_off("the epoch (eagerly; the window graphs are captured again)")
        cd.invalidate()
        return train_epoch_dynamic(cd.model, cd.graph, cd.edges, cd.targets, cd.B, cd.optimizer, cd.bucket, cd.feat, epoch, cd.rank,
                                   cd.world, cd.group, seed, cd.norm_fn, timed_comm)
    return losses


def in_degree_norm(graph) -> torch.Tensor:
    """norm = in_deg^-0.5, inf -> 0, computed on the device from the current snapshot (one launch: kernels.degree_norm;
    the torch composition -- five launches -- for tensors that are not int32 on a GPU)."""
    from . import kernels
    if hasattr(graph, "in_degree_norm_tensor"):
        pre = graph.in_degree_norm_tensor()          # made by the graph step itself (delta stores, fused rebuild)
        if pre is not None:
            return pre
    if hasattr(graph, "in_degrees_tensor"):
        deg = graph.in_degrees_tensor()
        if deg.is_cuda and deg.dtype == torch.int32 and deg.is_contiguous():
            return kernels.degree_norm(degrees=deg)
        deg = deg.float()
    else:
        ro = graph.csr("fwd").row_offset
        if ro.is_cuda and ro.dtype == torch.int32 and ro.is_contiguous():
            return kernels.degree_norm(row_offsets=ro)
        deg = (ro[1:] - ro[:-1]).float()
    norm = torch.pow(deg, -0.5)
    norm[torch.isinf(norm)] = 0
    return norm.unsqueeze(1)
