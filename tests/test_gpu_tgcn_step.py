"""stg_tgcn_step_fwd / _bwd (csrc/tgcn_step.hip: one TGCN training step per launch) against an fp64 torch restatement
of reference nn/pytorch/temporal/tgcn.py:21-55 + benchmarking/static-temporal-tgcn/seastar/model.py:6-18 with
autograd, two chained steps (so the backward gather of the next step's input gradient is exercised), and against
the kernels it replaces (gcn_agg: P bit-identical)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

C, FIN, FH = 64, 32, 32
LO, HI = -1e6, 1e6


def _graph(cuda, n, e, seed):
    from stgraph_amd import kernels
    rng = np.random.default_rng(seed)
    keys = rng.choice(n * n, size=e, replace=False)
    src, dst = (keys // n).astype(np.int32), (keys % n).astype(np.int32)
    if n > 40:                                   # a hub (long row) and an isolated vertex
        hub = np.arange(1, 38, dtype=np.int32)
        src, dst = np.concatenate([src, hub]), np.concatenate([dst, np.zeros_like(hub)])
        keep = (src != n - 1) & (dst != n - 1)
        src, dst = src[keep], dst[keep]
        k = np.unique(src.astype(np.int64) * n + dst, return_index=True)[1]
        src, dst = src[k], dst[k]
    g = kernels.build_graph_csr(src, dst, n, cuda)
    return g, len(src)


def _params(cuda, seed, scale=0.3):
    gen = torch.Generator(device=cuda).manual_seed(seed)
    r = lambda *s: (torch.randn(*s, device=cuda, generator=gen) * scale)  # noqa: E731
    return dict(Wcat=r(FIN, 3 * C), b3=r(3 * C), Wz=r(C, 2 * C), bz=r(C), Wr=r(C, 2 * C), br=r(C), Wh=r(C, 2 * C), bh=r(C),
                W1=r(FH, C), b1=r(FH), W2=r(1, FH), b2=r(1))


def _dense_adj(g, norm, ew, n, fwd=True):
    """A_hat (fp64, dense) of the forward CSR: out[r] = norm[r] * sum_e norm[col[e]] * w[eid[e]] * x[col[e]]."""
    csr = g.fwd if fwd else g.bwd
    ro, col, eid = csr.row_offset.long(), csr.column_indices.long(), csr.eids.long()
    rows = torch.repeat_interleave(torch.arange(n, device=ro.device), ro[1:] - ro[:-1])
    val = norm.double().view(-1)[rows] * norm.double().view(-1)[col]
    if ew is not None:
        val = val * ew.double().view(-1)[eid]
    A = torch.zeros(n, n, dtype=torch.float64, device=ro.device)
    A.index_put_((rows, col), val, accumulate=True)
    return A


def _ref_step(A, x, H, p, target, lo=LO, hi=HI, like=None):
    """One step in fp64.  ``like`` = the kernel's saved tensors of the same step: the reference then differentiates on the KERNEL's side
    of the two kinks (clamp of x3, relu of Hn) -- among |V| x 64 hidden values a few lie within fp32 rounding of zero, and an
    independent evaluation lands on either side of it (the values themselves agree to rounding either way)."""
    keep = {}
    P = A @ x
    x3 = P @ p["Wcat"] + p["b3"]
    x3.retain_grad()
    if like is None:
        h3 = torch.clamp(x3, lo, hi)
    else:
        k3 = like["x3"].double()
        h3 = torch.where((k3 >= lo) & (k3 <= hi), x3, torch.clamp(x3, lo, hi).detach())
    hz, hr, hh = h3[:, :C], h3[:, C:2 * C], h3[:, 2 * C:]
    zl = torch.cat([hz, H], 1) @ p["Wz"].t() + p["bz"]
    rl = torch.cat([hr, H], 1) @ p["Wr"].t() + p["br"]
    Z, R = torch.sigmoid(zl), torch.sigmoid(rl)
    hl = torch.cat([hh, H * R], 1) @ p["Wh"].t() + p["bh"]
    Ht = torch.tanh(hl)
    Hn = Z * H + (1 - Z) * Ht
    act = torch.relu(Hn) if like is None else torch.where(like["Hn"] > 0, Hn, torch.relu(Hn).detach())
    y = act @ p["W1"].t() + p["b1"]
    y_out = y @ p["W2"].t() + p["b2"]
    loss = torch.mean((y_out.view(-1) - target) ** 2)
    for k, v in (("zl", zl), ("rl", rl), ("hl", hl), ("y", y), ("y_out", y_out)):
        v.retain_grad()
        keep[k] = v
    keep.update(P=P, x3=x3, Z=Z, R=R, Ht=Ht, Hn=Hn, HR=H * R, loss=loss)
    return keep


def _alloc(cuda, n):
    new = lambda *s: torch.full(s, float("nan"), device=cuda)  # noqa: E731
    return dict(P=new(n, FIN), x3=new(n, 3 * C), Z=new(n, C), R=new(n, C), Ht=new(n, C), Hn=new(n, C), HR=new(n, C),
                y=new(n, FH), y_out=new(n), loss_partial=new(-(-n // 16)),
                clamp_mask=torch.zeros(n, 12, dtype=torch.int32, device=cuda))


def _fwd(cuda, g, norm, ew, p, x, H, target, n, head=2, node_ids=False, lo=LO, hi=HI, x3form=False):
    from stgraph_amd import kernels
    out = _alloc(cuda, n)
    if x3form == "folded32":                  # the conv folded into the gate Linears: the FOLD form of csrc/tgcn_step_fwd.hip
        Wc = [p["Wcat"][:, k * C:(k + 1) * C].contiguous() for k in range(3)]
        bc = [p["b3"][k * C:(k + 1) * C].contiguous() for k in range(3)]
        out["w_fold"], out["b_fold"], bound, w_fold_t = kernels.tgcn_fold_weights(*Wc, *bc, p["Wz"], p["bz"], p["Wr"], p["br"], p["Wh"],
                                                                                  p["bh"], with_bound=True)
        assert torch.equal(w_fold_t.view(3, FIN, C), out["w_fold"].view(3, C, FIN + C)[:, :, :FIN].transpose(1, 2))
        assert out["w_fold"].shape == (3 * C, FIN + C) and out["b_fold"].shape == (3 * C,)
        assert float(bound[0]) == float(p["Wcat"].abs().max()) and float(bound[1]) == float(p["b3"].abs().max())
        out["fold_bound"], out["x3"] = bound, None     # x3 is not formed (nor asked for), the clamp is bounded instead of looked at
        out["clamp_mask"] = kernels.step_ones_mask(n, cuda)
    nc = kernels._edge_gathered(g.fwd, "norm", norm, g.fwd.column_indices)
    ew_e = None if ew is None else kernels._edge_gathered(g.fwd, "ew", ew, g.fwd.eids)
    kernels.tgcn_step_fwd(n, C, FIN, FH, head, lo, hi, cuda, row_offsets=g.fwd.row_offset, column_indices=g.fwd.column_indices,
                          node_ids=g.fwd.node_ids if node_ids else None, norm_col_edge=nc, ew_edge=ew_e, norm=norm.view(-1),
                          x=x, H=H, target=target, WcatT=p["Wcat"].t().contiguous(), b3=p["b3"], Wz=p["Wz"], bz=p["bz"],
                          Wr=p["Wr"], br=p["br"], Wh=p["Wh"], bh=p["bh"], W1=p["W1"], b1=p["b1"],
                          W2=p["W2"].view(-1).contiguous(), b2=p["b2"], **out)
    for k in ("w_fold", "b_fold", "fold_bound"):
        out.pop(k, None)
    if x3form == "folded32":                   # the launch formed no x3: what the checks downstream read is P Wcat + b3
        out["x3"] = out["P"] @ p["Wcat"] + p["b3"]
    return out


def _bwd(cuda, g, norm, ew, p, saved, H, target, n, zn, dHn, g_cost, want_z=True, head=2, node_ids=False, lo=LO, hi=HI,
         use_mask=False, x3form=False):
    from stgraph_amd import kernels
    new = lambda *s: torch.full(s, float("nan"), device=cuda)  # noqa: E731
    out = dict(dzl=new(n, C), drl=new(n, C), dhl=new(n, C), da3=new(n, 3 * C), dH=new(n, C), dyt=new(n, FH), dyo=new(n),
               z=new(n, FIN) if want_z else None)
    if x3form == "folded32":                     # the folded backward launch: z from d_g and the folded weights, no da3, no mask read
        Wc = [p["Wcat"][:, k * C:(k + 1) * C].contiguous() for k in range(3)]
        bc = [p["b3"][k * C:(k + 1) * C].contiguous() for k in range(3)]
        out["w_fold_t"] = kernels.tgcn_fold_weights(*Wc, *bc, p["Wz"], p["bz"], p["Wr"], p["br"], p["Wh"], p["bh"], with_bound=True)[3]
        out["da3"], use_mask = None, True
        if not want_z:
            out["z"] = new(n, FIN)
    nc = kernels._edge_gathered(g.bwd, "norm", norm, g.bwd.column_indices)
    ew_e = None if ew is None else kernels._edge_gathered(g.bwd, "ew", ew, g.bwd.eids)
    kernels.tgcn_step_bwd(n, C, FIN, FH, head, lo, hi, cuda, row_offsets=g.bwd.row_offset, column_indices=g.bwd.column_indices,
                          node_ids=g.bwd.node_ids if node_ids else None, norm_col_edge=nc, ew_edge=ew_e, norm=norm.view(-1),
                          zn=zn, g_y=None, dHn=dHn, g_cost=g_cost, Z=saved["Z"], R=saved["R"], Ht=saved["Ht"], H=H,
                          Hn=saved["Hn"], x3=None if use_mask else saved["x3"],
                          clamp_mask=saved["clamp_mask"] if use_mask else None, y_out=saved["y_out"], target=target,
                          WzT=p["Wz"].t().contiguous(), WrT=p["Wr"].t().contiguous(), WhT=p["Wh"].t().contiguous(),
                          Wcat=p["Wcat"], W1T=p["W1"].t().contiguous(), W2=p["W2"].view(-1).contiguous(), **out)
    if out.pop("w_fold_t", None) is not None:    # the launch formed no da3: what the checks downstream read is d_g Wg[:, :C]
        out["da3"] = torch.cat([out["dzl"] @ p["Wz"][:, :C], out["drl"] @ p["Wr"][:, :C], out["dhl"] @ p["Wh"][:, :C]], 1)
        if not want_z:
            out["z"] = None
    return out


def _close(got, want, what, tol=2e-5):
    want = want.to(torch.float64)
    err = (got.double() - want).abs().max().item()
    scale = want.abs().max().item() + 1e-30
    assert err <= tol * scale + 1e-7, (what, err, scale)


@pytest.mark.parametrize("n,e,use_ew,node_ids", [(300, 2400, True, False), (3001, 30000, True, True), (1000, 9000, False, False),
                                                 (17, 60, True, False), (50_000, 500_000, True, False),
                                                 # 7501 tiles on 3072 wave slots, one row in the last tile: every wave takes
                                                 # further tiles off the workgroup's counter
                                                 (120_001, 1_000_000, False, False)])
@pytest.mark.parametrize("x3form", [False, "folded32"])
def test_two_chained_steps_match_fp64_autograd(cuda, n, e, use_ew, node_ids, x3form):
    """``x3form`` "folded32": the folded form of both launches (the conv folded into the gate Linears, no x3 / da3 formed) -- the
    SAME fp64 reference and the SAME tolerances as the reference formulation; its status word stays clear."""
    from stgraph_amd import kernels
    kernels.step_fold_status_word(cuda).zero_()
    g, e = _graph(cuda, n, e, seed=n)
    gen = torch.Generator(device=cuda).manual_seed(n + 1)
    deg = (g.fwd.row_offset[1:] - g.fwd.row_offset[:-1]).float()
    norm = torch.where(deg > 0, deg.clamp(min=1) ** -0.5, torch.zeros_like(deg)).view(-1, 1)
    ew = (torch.rand(e, 1, device=cuda, generator=gen) + 0.5) if use_ew else None
    p = _params(cuda, n + 2)
    x0 = torch.randn(n, FIN, device=cuda, generator=gen)
    t0, t1 = torch.randn(n, device=cuda, generator=gen), torch.randn(n, device=cuda, generator=gen)
    g_cost = torch.tensor([0.37], device=cuda)

    s0 = _fwd(cuda, g, norm, ew, p, x0, None, t0, n, node_ids=node_ids, x3form=x3form)          # H = None: zeros
    s1 = _fwd(cuda, g, norm, ew, p, s0["y"], s0["Hn"], t1, n, node_ids=node_ids, x3form=x3form)
    # P is the aggregation kernel's own arithmetic, bit for bit
    assert torch.equal(s0["P"], kernels.gcn_agg(x0, norm, norm, g.fwd, ew=ew))
    assert torch.equal(s1["P"], kernels.gcn_agg(s0["y"], norm, norm, g.fwd, ew=ew))

    b1 = _bwd(cuda, g, norm, ew, p, s1, s0["Hn"], t1, n, zn=None, dHn=None, g_cost=g_cost, node_ids=node_ids, x3form=x3form)
    b0 = _bwd(cuda, g, norm, ew, p, s0, None, t0, n, zn=b1["z"], dHn=b1["dH"], g_cost=g_cost, node_ids=node_ids, x3form=x3form)
    b0_again = _bwd(cuda, g, norm, ew, p, s0, None, t0, n, zn=b1["z"], dHn=b1["dH"], g_cost=g_cost, node_ids=node_ids, x3form=x3form)
    assert all(torch.equal(b0[k], b0_again[k]) for k in b0)            # deterministic: no atomics, fixed orders
    assert int(kernels.step_fold_status_word(cuda).item()) == 0
    if x3form == "folded32":
        s0_again = _fwd(cuda, g, norm, ew, p, x0, None, t0, n, node_ids=node_ids, x3form=x3form)
        assert all(torch.equal(s0[k], s0_again[k]) for k in s0)
        assert bool((s0["clamp_mask"] == 0xffff).all())

    if n > 20_000:
        A = None                                   # dense A_hat would be 20 GB: aggregate with the (tested) kernel in fp32
        agg = lambda v, csr: kernels.gcn_agg(v.float().contiguous(), norm, norm, csr, ew=ew).double()  # noqa: E731

        class _Agg(torch.autograd.Function):
            @staticmethod
            def forward(ctx, v):
                return agg(v, g.fwd)

            @staticmethod
            def backward(ctx, gr):
                return agg(gr, g.bwd)
        mul = _Agg.apply
    else:
        A = _dense_adj(g, norm, ew, n)
        mul = lambda v: A @ v  # noqa: E731
    pd = {k: v.double().requires_grad_(True) for k, v in p.items()}
    xr = x0.double().requires_grad_(True)
    H0 = torch.zeros(n, C, dtype=torch.float64, device=cuda, requires_grad=True)

    class _A:                                       # A @ x through `mul`
        def __matmul__(self, v):
            return mul(v)
    r0 = _ref_step(_A(), xr, H0, pd, t0.double(), like=s0)
    r1 = _ref_step(_A(), r0["y"], r0["Hn"], pd, t1.double(), like=s1)
    ((r0["loss"] + r1["loss"]) * 0.37).backward()

    tol = 2e-5 if n <= 20_000 else 2e-4             # the fp32 aggregation inside the large reference
    for k in ("x3", "Z", "R", "Ht", "Hn", "HR", "y"):
        _close(s0[k], r0[k].detach(), "s0." + k, tol)
        _close(s1[k], r1[k].detach(), "s1." + k, tol)
    _close(s0["y_out"], r0["y_out"].detach().view(-1), "y_out0", tol)
    cost = kernels.tgcn_window_loss(torch.stack([s0["loss_partial"], s1["loss_partial"]]), 2, n)
    _close(cost, (r0["loss"] + r1["loss"]).detach().view(1), "cost", tol)

    btol = 10 * tol
    for b, r, tag in ((b1, r1, "1"), (b0, r0, "0")):
        _close(b["dzl"], r["zl"].grad, "dzl" + tag, btol)
        _close(b["drl"], r["rl"].grad, "drl" + tag, btol)
        _close(b["dhl"], r["hl"].grad, "dhl" + tag, btol)
        _close(b["da3"], r["x3"].grad, "da3" + tag, btol)
        _close(b["dyt"], r["y"].grad, "dyt" + tag, btol)
        _close(b["dyo"], r["y_out"].grad.view(-1), "dyo" + tag, btol)
    _close(b0["dH"], H0.grad, "dH0", btol)
    dx = kernels.gcn_agg(b0["z"], norm, norm, g.bwd, ew=ew)       # the input gradient = A_hat^T z
    _close(dx, xr.grad, "dx0", btol)

    # the saved tensors are enough for every weight gradient (what gemm_tn contracts once per window)
    def wg(dpre, left, right):
        return dpre.double().t() @ torch.cat([left.double(), right.double()], 1)
    Hz = torch.zeros(n, C, device=cuda)
    clamp = lambda s, gate: s["x3"][:, gate * C:(gate + 1) * C].clamp(LO, HI)  # noqa: E731
    _close(wg(b0["dzl"], clamp(s0, 0), Hz) + wg(b1["dzl"], clamp(s1, 0), s0["Hn"]), pd["Wz"].grad, "dWz", btol)
    _close(wg(b0["drl"], clamp(s0, 1), Hz) + wg(b1["drl"], clamp(s1, 1), s0["Hn"]), pd["Wr"].grad, "dWr", btol)
    _close(wg(b0["dhl"], clamp(s0, 2), s0["HR"]) + wg(b1["dhl"], clamp(s1, 2), s1["HR"]), pd["Wh"].grad, "dWh", btol)
    _close(s0["P"].double().t() @ b0["da3"].double() + s1["P"].double().t() @ b1["da3"].double(), pd["Wcat"].grad, "dWcat", btol)
    _close(b0["dyt"].double().t() @ s0["Hn"].double().relu() + b1["dyt"].double().t() @ s1["Hn"].double().relu(),
           pd["W1"].grad, "dW1", btol)
    _close((b0["dyo"].double() @ s0["y"].double() + b1["dyo"].double() @ s1["y"].double()).view(1, -1), pd["W2"].grad, "dW2", btol)


def test_folded_form_reports_a_clamp_that_may_bite(cuda):
    """The folded forward launch does not form x3: it refuses on its BOUND |P_r|_1 max |Wcat| + max |b3| (conservative) by raising the
    sticky status word that check_step_fold_status() turns into an error (the epoch functions of stgraph_amd.temporal read the
    same word and rerun instead) -- and stays silent where the bound holds (the layer's own +-1e6)."""
    from stgraph_amd import kernels
    n, lo, hi = 200, -0.25, 0.4
    g, e = _graph(cuda, n, 1500, seed=5)
    deg = (g.fwd.row_offset[1:] - g.fwd.row_offset[:-1]).float()
    norm = torch.where(deg > 0, deg.clamp(min=1) ** -0.5, torch.zeros_like(deg)).view(-1, 1)
    p = _params(cuda, 9)
    x0, t0, H = torch.randn(n, FIN, device=cuda), torch.randn(n, device=cuda), torch.randn(n, C, device=cuda) * 0.3
    kernels.step_fold_status_word(cuda).zero_()
    kernels.check_step_fold_status(cuda)
    _fwd(cuda, g, norm, None, p, x0, H, t0, n, lo=lo, hi=hi, x3form="folded32")
    with pytest.raises(RuntimeError, match="folded step form"):
        kernels.check_step_fold_status(cuda)
    kernels.check_step_fold_status(cuda)             # cleared by the failed check
    _fwd(cuda, g, norm, None, p, x0, H, t0, n, x3form="folded32")
    kernels.check_step_fold_status(cuda)
    # the un-folded form given the status word alone reports an ACTUAL clamp the same way
    out = _alloc(cuda, n)
    out["x3"] = None
    nc = kernels._edge_gathered(g.fwd, "norm", norm, g.fwd.column_indices)
    kernels.tgcn_step_fwd(n, C, FIN, FH, 2, lo, hi, cuda, row_offsets=g.fwd.row_offset, column_indices=g.fwd.column_indices, norm_col_edge=nc,
                          norm=norm.view(-1), x=x0, H=H, target=t0, WcatT=p["Wcat"].t().contiguous(), b3=p["b3"], Wz=p["Wz"], bz=p["bz"],
                          Wr=p["Wr"], br=p["br"], Wh=p["Wh"], bh=p["bh"], W1=p["W1"], b1=p["b1"], W2=p["W2"].view(-1).contiguous(),
                          b2=p["b2"], fold_status=kernels.step_fold_status_word(cuda), **out)
    with pytest.raises(RuntimeError, match="folded step form"):
        kernels.check_step_fold_status(cuda)


def test_retired_forms_are_refused(cuda):
    """ABI 26: `w_image` (the bf16-split form) must be NULL; w_fold needs fold_bound, x3 == NULL, head >= 1 -- errors, not fallbacks."""
    import ctypes
    from stgraph_amd import _C
    a = _C.TgcnStepFwdArgs()
    a.N, a.C, a.Fin, a.Fh, a.head = 16, C, FIN, FH, 2
    buf = torch.zeros(64, device=cuda)
    a.w_image = buf.data_ptr()
    with pytest.raises(_C.StgError, match="retired"):
        _C.check(_C.lib.stg_tgcn_step_fwd(ctypes.byref(a), None))
    b = _C.TgcnStepBwdArgs()
    b.N, b.C, b.Fin, b.Fh, b.head = 16, C, FIN, FH, 2
    b.w_image = buf.data_ptr()
    with pytest.raises(_C.StgError, match="retired"):
        _C.check(_C.lib.stg_tgcn_step_bwd(ctypes.byref(b), None))
    a.w_image = None
    a.w_fold = a.x = buf.data_ptr()                    # without b_fold / fold_bound / fold_status
    with pytest.raises(_C.StgError, match="folded form needs"):
        _C.check(_C.lib.stg_tgcn_step_fwd(ctypes.byref(a), None))
    with pytest.raises(_C.StgError):
        _C.set_tuning("step_impl", 1)


@pytest.mark.parametrize("x3form", [False])
def test_clamp_is_honoured(cuda, x3form):
    """A clamp that bites ([-0.25, 0.4] instead of the layer's +-1e6): forward clamps, x3 is kept unclamped, backward
    blocks the gradient exactly where x3 is outside [lo, hi]."""
    n, lo, hi = 200, -0.25, 0.4
    g, e = _graph(cuda, n, 1500, seed=5)
    deg = (g.fwd.row_offset[1:] - g.fwd.row_offset[:-1]).float()
    norm = torch.where(deg > 0, deg.clamp(min=1) ** -0.5, torch.zeros_like(deg)).view(-1, 1)
    p = _params(cuda, 9)
    x0 = torch.randn(n, FIN, device=cuda)
    t0 = torch.randn(n, device=cuda)
    H = torch.randn(n, C, device=cuda) * 0.3
    s0 = _fwd(cuda, g, norm, None, p, x0, H, t0, n, lo=lo, hi=hi, x3form=x3form)
    b0 = _bwd(cuda, g, norm, None, p, s0, H, t0, n, zn=None, dHn=None, g_cost=torch.ones(1, device=cuda), lo=lo, hi=hi, x3form=x3form)
    bm = _bwd(cuda, g, norm, None, p, s0, H, t0, n, zn=None, dHn=None, g_cost=torch.ones(1, device=cuda), lo=lo, hi=hi,
              use_mask=True, x3form=x3form)      # the mask the forward launch left instead of x3: same gradients, bit for bit
    assert all(torch.equal(b0[k], bm[k]) for k in ("da3", "dH", "dzl", "drl", "dhl", "z"))
    blocked = (s0["x3"] > hi) | (s0["x3"] < lo)
    assert 0.2 < blocked.float().mean() < 0.9            # the clamp really is active, x3 itself is kept unclamped
    assert not b0["da3"][blocked].any() and b0["da3"][~blocked].abs().sum() > 0
    A = _dense_adj(g, norm, None, n)
    pd = {k: v.double().requires_grad_(True) for k, v in p.items()}
    Hd = H.double().requires_grad_(True)
    r0 = _ref_step(A, x0.double(), Hd, pd, t0.double(), lo, hi)
    r0["loss"].backward()
    _close(s0["Hn"], r0["Hn"].detach(), "Hn")
    _close(b0["da3"], r0["x3"].grad, "da3", 2e-4)
    _close(b0["dH"], Hd.grad, "dH", 2e-4)


def test_cell_only_mode_matches_the_gather_mode(cuda):
    """x = NULL: a3 = A_hat (x Wcat) is given (the generic layer path) -- same cell, same outputs to rounding."""
    from stgraph_amd import kernels
    n = 1234
    g, e = _graph(cuda, n, 9000, seed=3)
    deg = (g.fwd.row_offset[1:] - g.fwd.row_offset[:-1]).float()
    norm = torch.where(deg > 0, deg.clamp(min=1) ** -0.5, torch.zeros_like(deg)).view(-1, 1)
    p = _params(cuda, 4)
    x0, H = torch.randn(n, FIN, device=cuda), torch.randn(n, C, device=cuda) * 0.5
    full = _fwd(cuda, g, norm, None, p, x0, H, torch.zeros(n, device=cuda), n, head=0)
    a3 = kernels.gcn_agg(x0, norm, norm, g.fwd) @ p["Wcat"]
    out = _alloc(cuda, n)
    kernels.tgcn_step_fwd(n, C, FIN, FH, 0, LO, HI, cuda, a3=a3.contiguous(), H=H, b3=p["b3"], Wz=p["Wz"], bz=p["bz"], Wr=p["Wr"],
                          br=p["br"], Wh=p["Wh"], bh=p["bh"], x3=out["x3"], Z=out["Z"], R=out["R"], Ht=out["Ht"], Hn=out["Hn"],
                          HR=out["HR"], clamp_mask=None)
    for k in ("x3", "Z", "R", "Ht", "Hn", "HR"):
        _close(out[k], full[k], k, 1e-5)


@pytest.mark.parametrize("x3form", [False])
@pytest.mark.parametrize("n,e,m", [(3001, 30000, 5000), (17, 60, 40), (25_000, 250_000, 12_500)])
def test_link_loss_backward_inside_the_step_launch(cuda, n, e, m, x3form):
    """head == 1 with the link_* fields: the node side of the link-prediction loss's backward (stg_link_decode_bwd) taken inside
    the backward step launch -- every output bit for bit what the two launches produce (same terms, same order)."""
    from stgraph_amd import kernels
    g, e = _graph(cuda, n, e, seed=n + 3)
    gen = torch.Generator(device=cuda).manual_seed(n + 4)
    deg = (g.fwd.row_offset[1:] - g.fwd.row_offset[:-1]).float()
    norm = torch.where(deg > 0, deg.clamp(min=1) ** -0.5, torch.zeros_like(deg)).view(-1, 1)
    p = _params(cuda, n + 5)
    x0 = torch.randn(n, FIN, device=cuda, generator=gen)
    H = torch.randn(n, C, device=cuda, generator=gen) * 0.3
    s = _fwd(cuda, g, norm, None, p, x0, H, torch.zeros(n, device=cuda), n, head=1, x3form=x3form)
    edge_index = torch.randint(0, n, (2, m), device=cuda, generator=gen)
    target = (torch.rand(m, device=cuda, generator=gen) < 0.5).float()
    y = s["y"]
    logits = (y[edge_index[0]] * y[edge_index[1]]).sum(1).contiguous()
    inc = kernels.link_incidence(edge_index, n)
    g_cost = torch.tensor([0.61], device=cuda)
    zn = torch.randn(n, FIN, device=cuda, generator=gen)
    dHn = torch.randn(n, C, device=cuda, generator=gen)
    nc = kernels._edge_gathered(g.bwd, "norm", norm, g.bwd.column_indices)

    def run(**kw):
        new = lambda *sh: torch.full(sh, float("nan"), device=cuda)  # noqa: E731
        out = dict(dzl=new(n, C), drl=new(n, C), dhl=new(n, C), da3=new(n, 3 * C), dH=new(n, C), dyt=new(n, FH), z=new(n, FIN))
        kernels.tgcn_step_bwd(n, C, FIN, FH, 1, LO, HI, cuda, row_offsets=g.bwd.row_offset, column_indices=g.bwd.column_indices,
                              norm_col_edge=nc, norm=norm.view(-1), zn=zn, dHn=dHn, Z=s["Z"], R=s["R"], Ht=s["Ht"], H=H,
                              Hn=s["Hn"], clamp_mask=s["clamp_mask"], WzT=p["Wz"].t().contiguous(), WrT=p["Wr"].t().contiguous(),
                              WhT=p["Wh"].t().contiguous(), Wcat=p["Wcat"], W1T=p["W1"].t().contiguous(), **out, **kw)
        return out
    dy = torch.empty(n, FH, device=cuda)
    kernels.link_decode_bwd(g_cost, y, logits, target, inc, dy)
    two = run(g_y=dy)
    one = run(link_edges=m, g_cost=g_cost, link_row_ptr=inc[0], link_other=inc[1], link_eid=inc[2], link_y=y, link_logits=logits,
              link_target=target)
    for k in two:
        assert torch.equal(one[k], two[k]), k


def test_pack_weights(cuda):
    """stg_tgcn_pack_weights: the cat / transposes of a window's weights in one launch, element for element."""
    from stgraph_amd import kernels
    p = _params(cuda, 3)
    Wc = [p["Wcat"][:, k * C:(k + 1) * C].contiguous() for k in range(3)]
    bc = [p["b3"][k * C:(k + 1) * C].contiguous() for k in range(3)]
    Wcat, WcatT, b3, WzT, WrT, WhT, W1T = kernels.tgcn_pack_weights(*Wc, *bc, p["Wz"], p["Wr"], p["Wh"], p["W1"])
    assert torch.equal(Wcat, p["Wcat"]) and torch.equal(WcatT, p["Wcat"].t().contiguous()) and torch.equal(b3, p["b3"])
    for got, w in ((WzT, p["Wz"]), (WrT, p["Wr"]), (WhT, p["Wh"]), (W1T, p["W1"])):
        assert torch.equal(got, w.t().contiguous())


def _bf16_terms_to_f64(words):
    """[..., 3 terms, 8] uint16 bf16 bit patterns -> the fp64 sum of the three terms."""
    u = words.astype(np.uint32) << 16
    return u.view(np.float32).astype(np.float64).sum(-2)


def test_unfold_gate_grads_kernel_against_its_torch_statement(cuda):
    """stg_tgcn_unfold_gate_grads (the window nodes' gate / conv parameter gradients from d_g^T [Hx | P] and the column sums of d_g)
    against temporal._unfold_gate_grads in fp64, and that statement against the direct products d_g^T [x3_g | Hx], P^T (d_g Wg[:, :C])
    on random rows."""
    from stgraph_amd import kernels, temporal
    gen = torch.Generator(device=cuda).manual_seed(3)
    n = 700
    r = lambda *s: torch.randn(*s, device=cuda, generator=gen)  # noqa: E731
    Rs, css, Wcs, bcs, Wgs, want = [], [], [], [], [], []
    for g in range(3):
        d, Hx, P = r(n, C), r(n, C), r(n, FIN)
        Wc, bc, Wg = r(FIN, C) * 0.3, r(C) * 0.3, r(C, 2 * C) * 0.3
        Rs.append((d.double().t() @ torch.cat([Hx, P], 1).double()).float())
        css.append(d.double().sum(0).float())
        Wcs.append(Wc), bcs.append(bc), Wgs.append(Wg)
        x3 = P.double() @ Wc.double() + bc.double()
        direct = (d.double().t() @ torch.cat([x3, Hx.double()], 1), d.double().sum(0),
                  P.double().t() @ (d.double() @ Wg.double()[:, :C]), (d.double() @ Wg.double()[:, :C]).sum(0))
        stated = temporal._unfold_gate_grads(Rs[g][:, C:].double(), css[g].double(), Rs[g][:, :C].double(), Wc.double(), bc.double(), Wg.double())
        for a, b in zip(stated, direct):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5 * float(b.abs().max()))
        want.append(stated)
    got = kernels.tgcn_unfold_gate_grads(Rs, css, Wcs, bcs, Wgs)
    outs = [tuple(torch.full_like(t, float("nan")) for t in got[g]) for g in range(3)]
    assert kernels.tgcn_unfold_gate_grads(Rs, css, Wcs, bcs, Wgs, outs=outs) is outs
    for g in range(3):
        for a, o, b in zip(got[g], outs[g], want[g]):
            torch.testing.assert_close(a.double(), b, rtol=1e-5, atol=1e-5 * float(b.abs().max()))
            assert torch.equal(a, o)


@pytest.mark.parametrize("n,waves", [(50_000, 12), (49_990, 12), (70_001, 16), (66_000, 16)])
def test_shared_tiles_of_the_partial_round_are_bit_identical(cuda, n, waves):
    """Tiles of the last, partial round shared by four waves (knob "step_coop" 0, default) against one wave each (1): every output
    of the folded forward and backward launches, bit for bit.  |V| = 50 000 on 12-wave workgroups: 53 workgroups with one such
    tile (49 990: the last tile ragged); 16-wave workgroups at 66-70 K rows: up to two shared tiles per workgroup, two groups."""
    from stgraph_amd import _C, kernels
    g, e = _graph(cuda, n, 10 * n, seed=n)
    gen = torch.Generator(device=cuda).manual_seed(n + 1)
    deg = (g.fwd.row_offset[1:] - g.fwd.row_offset[:-1]).float()
    norm = torch.where(deg > 0, deg.clamp(min=1) ** -0.5, torch.zeros_like(deg)).view(-1, 1)
    ew = torch.rand(e, 1, device=cuda, generator=gen) + 0.5
    p = _params(cuda, n + 2)
    x0 = torch.randn(n, FIN, device=cuda, generator=gen)
    H = torch.randn(n, C, device=cuda, generator=gen) * 0.3
    t0 = torch.randn(n, device=cuda, generator=gen)
    zn, dHn = torch.randn(n, FIN, device=cuda, generator=gen), torch.randn(n, C, device=cuda, generator=gen)
    g_cost = torch.tensor([0.37], device=cuda)
    res = []
    _C.set_tuning("step_waves", waves)
    try:
        for coop_off in (0, 1):
            _C.set_tuning("step_coop", coop_off)
            s = _fwd(cuda, g, norm, ew, p, x0, H, t0, n, x3form="folded32")
            b = _bwd(cuda, g, norm, ew, p, s, H, t0, n, zn=zn, dHn=dHn, g_cost=g_cost, x3form="folded32")
            res.append({**{k: v for k, v in s.items() if torch.is_tensor(v)}, **{"b_" + k: v for k, v in b.items() if torch.is_tensor(v)}})
    finally:
        _C.set_tuning("step_coop", 0)
        _C.set_tuning("step_waves", 0)
    assert int(kernels.step_fold_status_word(cuda).item()) == 0
    for k in res[0]:
        assert not bool(torch.isnan(res[0][k].float()).any()), k
        assert torch.equal(res[0][k], res[1][k]), k
