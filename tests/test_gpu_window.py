"""temporal.window_cost (one fused autograd node per BPTT window over the one-launch TGCN step kernels) against the
per-snapshot formulation with every fusion switched off (the reference's loop spelled in torch + the aggregation
kernels), at a small size with a ragged last window, and at the full BASELINE configs[3] size through the captured
window (HIP graph) with Adam steps."""
import contextlib

import pytest
import torch

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def _unfused():
    """One autograd node per torch op: no fused window, head, cell or gate aggregation, per-step weight gradients."""
    from stgraph_amd import temporal
    from stgraph_amd.nn import functional as SF
    from stgraph_amd.nn.pytorch.temporal import cell
    from stgraph_amd.nn.pytorch.temporal.tgcn import TGCN
    old = (TGCN.fuse_cell, TGCN.fuse_gates)
    temporal.set_fused_window(False)
    temporal.set_fused_head(False)
    cell.set_fused_forward(False)
    cell.set_fused_backward(False)
    SF.set_deferred_weight_grads(False)
    TGCN.fuse_cell = TGCN.fuse_gates = False
    try:
        yield
    finally:
        temporal.set_fused_window(True)
        temporal.set_fused_head(True)
        cell.set_fused_forward(True)
        cell.set_fused_backward(True)
        SF.set_deferred_weight_grads(True)
        TGCN.fuse_cell, TGCN.fuse_gates = old


def _setup(cuda, n, e, T, seed):
    from stgraph_amd import temporal
    from stgraph_amd.graph import StaticGraph
    from tests.util import random_graph
    src, dst = random_graph(seed, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    g.set_ndata("norm", temporal.in_degree_norm(g))
    gen = torch.Generator(device=cuda).manual_seed(seed)
    ew = torch.rand(len(src), 1, device=cuda, generator=gen) + 0.5
    targets = torch.randn(T, n, 1, device=cuda, generator=gen)
    return g, ew, targets, gen


@pytest.mark.parametrize("graph_type_ids", [False, True])
def test_window_cost_matches_the_unfused_loop(cuda, graph_type_ids):
    from stgraph_amd import temporal
    n, e, B = 4321, 40000, 5
    g, ew, targets, gen = _setup(cuda, n, e, B, 11)
    x0 = torch.randn(n, 32, device=cuda, generator=gen).requires_grad_(True)
    torch.manual_seed(3)
    model = temporal.STGraphTGCN(32, 64, 1).to(cuda)
    if graph_type_ids:
        g.graph_type = lambda: "csr"               # rows through node_ids (reference tpl_fa_csr.jinja:13-18)
    assert temporal.window_cost_usable(model, g, x0, ew, targets)
    cost = temporal.window_cost_of(model, g, x0, ew, targets) / (B + 1)
    cost.backward()
    got = (cost.detach().clone(), x0.grad.clone(), {k: p.grad.clone() for k, p in model.named_parameters()})
    model.zero_grad()
    x0.grad = None
    with _unfused():
        assert not temporal.window_cost_usable(model, g, x0, ew, targets)
        ref = temporal.window_cost_of(model, g, x0, ew, targets) / (B + 1)
        ref.backward()
    torch.testing.assert_close(got[0], ref.detach(), rtol=1e-5, atol=1e-6)
    scale = lambda t: float(t.abs().max()) + 1e-12  # noqa: E731
    assert float((got[1] - x0.grad).abs().max()) <= 1e-4 * scale(x0.grad)
    for k, p in model.named_parameters():
        assert float((got[2][k] - p.grad).abs().max()) <= 1e-4 * scale(p.grad) + 1e-7, k


def test_epoch_with_a_ragged_last_window_matches(cuda):
    """T = 11, backprop_every = 4: windows of 4, 4, 3 snapshots; losses and parameters after two epochs of Adam."""
    from stgraph_amd import temporal
    n, e, T, B = 4100, 33000, 11, 4
    res = []
    for fused in (True, False):
        g, ew, targets, gen = _setup(cuda, n, e, T, 5)
        torch.manual_seed(1)
        model = temporal.STGraphTGCN(32, 64, 1).to(cuda)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        bucket = temporal.GradBucket(model.parameters())
        ctx = contextlib.nullcontext() if fused else _unfused()
        with ctx:
            losses = []
            for ep in range(2):
                losses += temporal.train_epoch_static(model, g, ew, targets, B, opt, bucket, 32, epoch=ep)
        res.append((torch.stack(losses), [p.detach().clone() for p in model.parameters()]))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-4, atol=1e-6)
    for a, b in zip(res[0][1], res[1][1]):
        torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-5)       # after 6 Adam steps (sign-sensitive updates)


def test_captured_window_at_full_cfg4_size_matches_the_unfused_eager_loop(cuda):
    """BASELINE configs[3] shape (|V| = 50 K, |E| = 500 K, feat 32, hidden 64, backprop_every 25): two windows replayed
    from the captured HIP graph with Adam steps == the same two windows run eagerly with every fusion off."""
    from stgraph_amd import temporal
    n, e, B = 50_000, 500_000, 25
    T = 2 * B
    res = []
    for fused in (True, False):
        g, ew, targets, gen = _setup(cuda, n, e, T, 3)
        torch.manual_seed(3)
        model = temporal.STGraphTGCN(32, 64, 1).to(cuda)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        bucket = temporal.GradBucket(model.parameters())
        if fused:
            cw = temporal.CapturedStaticWindow(model, g, ew, targets, B, opt, bucket, 32)
            losses = temporal.train_epoch_static_captured(cw, model, g, ew, targets, opt, bucket, 32, epoch=0)
        else:
            with _unfused():
                losses = temporal.train_epoch_static(model, g, ew, targets, B, opt, bucket, 32, epoch=0)
        res.append((torch.stack(losses), [p.detach().clone() for p in model.parameters()]))
        del model, opt, bucket, g
        torch.cuda.empty_cache()
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-4, atol=1e-6)
    for a, b in zip(res[0][1], res[1][1]):
        # after two Adam steps of 1e-2: a gradient entry near zero moves its parameter by up to lr either way
        torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-4)


def test_captured_optimizer_tail_matches_the_eager_tail(cuda):
    """capturable Adam: the second graph (grad / world, optimizer step, window index) == the eager tail, over two epochs
    with a ragged last window (T = 11, B = 4); the window inputs come from the epoch's pre-drawn chunk in both."""
    from stgraph_amd import temporal
    n, e, T, B = 4100, 33000, 11, 4
    res = []
    for mode in ("captured_tail", "eager_tail", "eager_loop"):
        g, ew, targets, gen = _setup(cuda, n, e, T, 6)
        torch.manual_seed(2)
        model = temporal.STGraphTGCN(32, 64, 1).to(cuda)
        opt = (torch.optim.Adam(model.parameters(), lr=1e-2, capturable=True, fused=True) if mode == "captured_tail"
               else torch.optim.Adam(model.parameters(), lr=1e-2))
        bucket = temporal.GradBucket(model.parameters())
        losses = []
        if mode != "eager_loop":
            cw = temporal.CapturedStaticWindow(model, g, ew, targets, B, opt, bucket, 32)
            assert (cw.step_graph is not None) == (mode == "captured_tail")
        for ep in range(2):
            if mode == "eager_loop":
                losses += temporal.train_epoch_static(model, g, ew, targets, B, opt, bucket, 32, epoch=ep)
            else:
                losses += temporal.train_epoch_static_captured(cw, model, g, ew, targets, opt, bucket, 32, epoch=ep)
        res.append((torch.stack(losses), [p.detach().clone() for p in model.parameters()]))
    for other in (1, 2):
        torch.testing.assert_close(res[0][0], res[other][0], rtol=1e-4, atol=1e-6)
        for a, b in zip(res[0][1], res[other][1]):
            # six Adam steps with the gradients recomputed from the parameters: where a gradient entry is within rounding of
            # zero the fused capturable kernel and the default Adam move it by up to lr in either direction (both correct), so
            # the bulk of every tensor is held to (2e-3, 1e-4) and every entry to a few steps of lr; the tails themselves are
            # compared on FIXED gradients, strictly, in test_captured_tail_equals_the_eager_tail_on_fixed_gradients
            bad = (a - b).abs() > 1e-4 + 2e-3 * b.abs()
            assert bad.float().mean() <= 0.01 and float((a - b).abs().max()) <= 3e-2, (float(bad.float().mean()), float((a - b).abs().max()))


def test_captured_tail_equals_the_eager_tail_on_fixed_gradients(cuda):
    """The second graph of CapturedStaticWindow (grad / world, fused capturable Adam, window index) against the default Adam
    stepping eagerly, both fed the SAME sequence of well-conditioned gradients (|g| in [1e-3, 1], nothing near Adam's eps):
    parameters agree to 1e-6 after five steps -- the tails are the same update rule."""
    from stgraph_amd import temporal
    n, e, T, B = 2000, 16000, 8, 4
    g, ew, targets, gen = _setup(cuda, n, e, T, 6)
    res = []
    for captured in (True, False):
        torch.manual_seed(2)
        model = temporal.STGraphTGCN(32, 64, 1).to(cuda)
        opt = (torch.optim.Adam(model.parameters(), lr=1e-2, capturable=True, fused=True) if captured
               else torch.optim.Adam(model.parameters(), lr=1e-2))
        bucket = temporal.GradBucket(model.parameters())
        if captured:
            cw = temporal.CapturedStaticWindow(model, g, ew, targets, B, opt, bucket, 32)
            assert cw.step_graph is not None
        gg = torch.Generator(device=cuda).manual_seed(11)
        for step in range(5):
            mag = torch.rand(bucket.flat.shape, device=cuda, generator=gg) * (1 - 1e-3) + 1e-3
            sign = torch.where(torch.rand(bucket.flat.shape, device=cuda, generator=gg) < 0.5, -1.0, 1.0)
            bucket.flat.copy_(mag * sign)
            if captured:
                cw.step_graph.replay()
            else:
                opt.step()
        res.append([p.detach().clone() for p in model.parameters()])
    for a, b in zip(*res):
        torch.testing.assert_close(a, b, rtol=0, atol=1e-6)


@pytest.mark.parametrize("kind,full", [("naive_resident", False), ("naive_rebuild", False), ("pcsr", False),
                                       ("gpma", False), ("naive_resident", True), ("pcsr", True)])
def test_dynamic_window_matches_the_per_snapshot_loop(cuda, kind, full):
    """temporal.dyn_window_cost (one autograd node per window of the dynamic-temporal loop, every snapshot its own
    graph) == the per-snapshot loop, on every dynamic graph class: losses and parameters after two epochs of SGD.
    ``full``: at BASELINE configs[4]'s sizes (|V| = 25 K, 250 K edges +- 6250 per step, windows of 20), two windows."""
    import numpy as np
    from stgraph_amd import temporal
    from stgraph_amd.graph import GPMAGraph, NaiveGraph, PCSRGraph
    n, e0, churn, T, B, feat, hid, m = ((25_000, 250_000, 6_250, 41, 20, 32, 64, 10_000) if full
                                        else (4000, 30000, 800, 10, 4, 32, 64, 1500))
    rng = np.random.default_rng(7)
    stream = rng.choice(n * n, size=e0 + churn * T, replace=False)
    out = []
    for fused in (True, False):
        temporal.set_fused_window(fused)
        try:
            snaps, pn_edges, pn_targets = [], [], []
            gen = torch.Generator(device=cuda).manual_seed(4)
            for t in range(T):
                keys = stream[t * churn: t * churn + e0]
                s, d = (keys // n).astype(np.int32), (keys % n).astype(np.int32)
                snaps.append((torch.from_numpy(s).to(cuda), torch.from_numpy(d).to(cuda)))
                pos = torch.from_numpy(np.stack([s[:m], d[:m]]).astype(np.int64)).to(cuda)
                neg = torch.randint(0, n, (2, m), device=cuda, generator=gen)
                pn_edges.append(torch.cat([pos, neg], 1))
                pn_targets.append(torch.cat([torch.ones(m, device=cuda), torch.zeros(m, device=cuda)]))
            if kind.startswith("naive"):
                G = NaiveGraph(snaps, n, device=cuda, sort_inplace=False, resident=kind == "naive_resident", max_cached=B + 1)
            else:
                G = (PCSRGraph if kind == "pcsr" else GPMAGraph)(snaps, n, device=cuda)
            torch.manual_seed(4)
            model = temporal.DynamicSTGraphTGCN(feat, hid).to(cuda)
            opt = torch.optim.SGD(model.parameters(), lr=1e-2)
            bucket = temporal.GradBucket(model.parameters())
            losses = []
            for ep in range(2):
                if kind == "naive_rebuild":
                    G._snapshots.clear()
                G._ndata.clear()
                losses += temporal.train_epoch_dynamic(model, G, pn_edges, pn_targets, B, opt, bucket, feat, epoch=ep)
            out.append((torch.stack(losses), [p.detach().clone() for p in model.parameters()]))
        finally:
            temporal.set_fused_window(True)
    torch.testing.assert_close(out[0][0], out[1][0], rtol=2e-4, atol=1e-6)
    for a, b in zip(out[0][1], out[1][1]):
        torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("kind,optim", [("naive_resident", "adam"), ("naive_rebuild", "adam"), ("naive_resident", "sgd"),
                                        ("pcsr", "adam"), ("gpma", "adam"), ("pcsr", "sgd")])
def test_captured_dynamic_windows_match_the_eager_loop(cuda, kind, optim):
    """temporal.CapturedDynamicWindows (one HIP graph per window: snapshot moves / builds, norms, window cost, backward;
    captured optimizer tail when the optimizer is capturable) == train_epoch_dynamic on the same objects: per-window
    costs and parameters after one eager + three replayed epochs (the first of them captures)."""
    import numpy as np
    from stgraph_amd import temporal
    from stgraph_amd.graph import GPMAGraph, NaiveGraph, PCSRGraph
    n, e0, churn, T, B, feat, hid, m = 4000, 30000, 800, 13, 4, 32, 64, 1500
    rng = np.random.default_rng(11)
    stream = rng.choice(n * n, size=e0 + churn * T, replace=False)
    snaps, pn_edges, pn_targets = [], [], []
    gen = torch.Generator(device=cuda).manual_seed(4)
    for t in range(T):
        keys = stream[t * churn: t * churn + e0]
        s, d = (keys // n).astype(np.int32), (keys % n).astype(np.int32)
        snaps.append((torch.from_numpy(s).to(cuda), torch.from_numpy(d).to(cuda)))
        pos = torch.from_numpy(np.stack([s[:m], d[:m]]).astype(np.int64)).to(cuda)
        neg = torch.randint(0, n, (2, m), device=cuda, generator=gen)
        pn_edges.append(torch.cat([pos, neg], 1))
        pn_targets.append(torch.cat([torch.ones(m, device=cuda), torch.zeros(m, device=cuda)]))
    out = []
    for captured in (True, False):
        if kind.startswith("naive"):
            G = NaiveGraph(snaps, n, device=cuda, sort_inplace=False, resident=kind == "naive_resident", max_cached=B + 1)
        else:                                   # the delta stores: get_graph(t) itself (merge + CSR emission) is in the graph
            G = (PCSRGraph if kind == "pcsr" else GPMAGraph)(snaps, n, device=cuda)
        torch.manual_seed(4)
        model = temporal.DynamicSTGraphTGCN(feat, hid).to(cuda)
        opt = (torch.optim.Adam(model.parameters(), lr=1e-2, capturable=True, fused=True) if optim == "adam"
               else torch.optim.SGD(model.parameters(), lr=1e-2))
        bucket = temporal.GradBucket(model.parameters())
        cd, losses = None, []
        for ep in range(4):
            if kind == "naive_rebuild":
                G._snapshots.clear()
            G._ndata.clear()
            if captured and ep >= 1:
                cd = cd or temporal.CapturedDynamicWindows(model, G, pn_edges, pn_targets, B, opt, bucket, feat)
                ls = temporal.train_epoch_dynamic_captured(cd, epoch=ep)
                losses += [x.clone() for x in ls]
            else:
                losses += temporal.train_epoch_dynamic(model, G, pn_edges, pn_targets, B, opt, bucket, feat, epoch=ep)
        if captured:
            assert cd is not None and len(cd.graphs) == 3 and (cd.step_graph is not None) == (optim == "adam")   # window 3 = {t = 12}: no target
            assert len(cd._build_graphs) == 0             # (rebuild mode's builds-ahead option is off by default)
            if not kind.startswith("naive"):
                G.check()                       # the store's stream contract held through the replays
                assert G.current_timestamp == 11
        out.append((torch.stack(losses), [p.detach().clone() for p in model.parameters()]))
    torch.testing.assert_close(out[0][0], out[1][0], rtol=1e-5, atol=1e-7)
    for a, b in zip(out[0][1], out[1][1]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)


def test_builds_one_window_ahead_on_a_second_stream_change_nothing(cuda):
    """CapturedDynamicWindows.prefetch_builds (rebuild mode: a window's snapshot builds replayed on a second stream while the
    previous window trains) against the same builds at the head of the window's training graph: bit-identical costs and parameters over
    five replayed epochs of three windows -- an ordering mistake between the two streams (a build overwriting CSRs a training graph
    still reads, a training graph starting before its builds ended) would show here."""
    import numpy as np
    from stgraph_amd import temporal
    from stgraph_amd.graph import NaiveGraph
    n, e0, churn, T, B, feat, hid, m = 6000, 60000, 1500, 13, 4, 32, 64, 1500
    rng = np.random.default_rng(12)
    stream = rng.choice(n * n, size=e0 + churn * T, replace=False)
    snaps, pn_edges, pn_targets = [], [], []
    gen = torch.Generator(device=cuda).manual_seed(5)
    for t in range(T):
        keys = stream[t * churn: t * churn + e0]
        s, d = (keys // n).astype(np.int32), (keys % n).astype(np.int32)
        snaps.append((torch.from_numpy(s).to(cuda), torch.from_numpy(d).to(cuda)))
        pos = torch.from_numpy(np.stack([s[:m], d[:m]]).astype(np.int64)).to(cuda)
        neg = torch.randint(0, n, (2, m), device=cuda, generator=gen)
        pn_edges.append(torch.cat([pos, neg], 1))
        pn_targets.append(torch.cat([torch.ones(m, device=cuda), torch.zeros(m, device=cuda)]))
    out = []
    for prefetch in (True, False):
        G = NaiveGraph(snaps, n, device=cuda, sort_inplace=False, resident=False, max_cached=B + 1)
        torch.manual_seed(4)
        model = temporal.DynamicSTGraphTGCN(feat, hid).to(cuda)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=True, fused=True)
        bucket = temporal.GradBucket(model.parameters())
        cd, losses = None, []
        for ep in range(6):
            G._snapshots.clear()
            G._ndata.clear()
            if ep >= 1:
                if cd is None:
                    cd = temporal.CapturedDynamicWindows(model, G, pn_edges, pn_targets, B, opt, bucket, feat)
                    cd.prefetch_builds = prefetch
                losses += [x.clone() for x in temporal.train_epoch_dynamic_captured(cd, epoch=ep)]
            else:
                losses += temporal.train_epoch_dynamic(model, G, pn_edges, pn_targets, B, opt, bucket, feat, epoch=ep)
        assert len(cd._build_graphs) == (3 if prefetch else 0)
        if prefetch:
            assert cd._build_pending.get(0) is True              # the last window of an epoch built the next epoch's first
        torch.cuda.synchronize()
        out.append((torch.stack(losses), [p.detach().clone() for p in model.parameters()]))
    assert torch.equal(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert torch.equal(a, b)


def test_window_on_data_the_fold_refuses_falls_back_to_the_reference_formulation(cuda):
    """Inputs of size 3e5: the folded step's BOUND on the conv output leaves the clamp range (the conv output itself need not).  The
    eager window node notices (one read-back per window outside a capture), switches the folded formulation off for the process with
    a warning and recomputes the window in the reference formulation: bit for bit what the node gives with the folded formulation off
    from the start, no exception."""
    import warnings
    from stgraph_amd import kernels, temporal
    n, e, B = 2000, 16000, 3
    g, ew, targets, gen = _setup(cuda, n, e, B, 5)
    x0 = (torch.randn(n, 32, device=cuda, generator=gen) * 3e5).requires_grad_(True)
    torch.manual_seed(3)
    model = temporal.STGraphTGCN(32, 64, 1).to(cuda)
    was = kernels.STEP_FOLDED, kernels.STEP_WGRAD_FROM_P
    kernels.set_step_folded(True), kernels.set_step_wgrad_from_p(True)
    kernels.step_fold_status_word(cuda).zero_()
    try:
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            cost = temporal.window_cost_of(model, g, x0, ew, targets) / (B + 1)
        assert any("folded step formulation is switched off" in str(w.message) for w in rec)
        assert not kernels.STEP_FOLDED and not kernels.STEP_WGRAD_FROM_P
        assert int(kernels.step_fold_status_word(cuda).item()) == 0
        cost.backward()
        got = (cost.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()})
        model.zero_grad()
        x0.grad = None
        ref = temporal.window_cost_of(model, g, x0, ew, targets) / (B + 1)     # the reference formulation from the start: the same launches
        ref.backward()
        assert torch.equal(got[0], ref.detach())
        for k, p in model.named_parameters():
            assert torch.equal(got[1][k], p.grad), k
    finally:
        kernels.set_step_folded(was[0]), kernels.set_step_wgrad_from_p(was[1])
        kernels.step_fold_status_word(cuda).zero_()


def test_gate_gradients_as_one_matrix_give_the_same_weight_gradients(cuda):
    """kernels.STEP_WGRAD_ZR_TOGETHER: the backward step launches write d_z | d_r | d_h as the column blocks of ONE [N, 3C]
    matrix (stg_tgcn_step_bwd_args::ld_d) and the window contracts [d_z | d_r] against [H | P] as one operand -- against one
    contraction per gate over three [N, C] matrices: the same sums in the same split-K order, bit for bit."""
    from stgraph_amd import kernels, temporal
    n, e, B = 70_001, 500_000, 5
    g, ew, targets, gen = _setup(cuda, n, e, B, 13)
    x0 = torch.randn(n, 32, device=cuda, generator=gen).requires_grad_(True)
    torch.manual_seed(3)
    model = temporal.STGraphTGCN(32, 64, 1).to(cuda)
    res = []
    for on in (True, False):
        kernels.set_step_wgrad_zr_together(on)
        try:
            model.zero_grad()
            x0.grad = None
            rec = []
            kernels.enable_launch_timing(rec)
            cost = temporal.window_cost_of(model, g, x0, ew, targets) / (B + 1)
            cost.backward()
        finally:
            kernels.enable_launch_timing(None)
            kernels.set_step_wgrad_zr_together(True)
        kernels.check_step_fold_status(cuda)
        assert sum(r[0] == "gemm_tn_form" for r in rec) == (4 if on else 5), [r[0] for r in rec]
        res.append((cost.detach().clone(), x0.grad.clone(), {k: p.grad.clone() for k, p in model.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for k in res[0][2]:
        a, b = res[0][2][k], res[1][2][k]
        assert float((a - b).abs().max()) <= 1e-6 * (float(b.abs().max()) + 1e-12), k


@contextlib.contextmanager
def _folded_on(cuda):
    from stgraph_amd import kernels
    was = kernels.STEP_FOLDED, kernels.STEP_WGRAD_FROM_P
    kernels.set_step_folded(True), kernels.set_step_wgrad_from_p(True)
    kernels.step_fold_status_word(cuda).zero_()
    try:
        yield
    finally:
        kernels.set_step_folded(was[0]), kernels.set_step_wgrad_from_p(was[1])
        kernels.step_fold_status_word(cuda).zero_()


def _trip(model):
    """A conv bias of 2e6: the conv output of every row leaves [-1e6, 1e6] (the reference clamps it; the folded formulation
    refuses it through its bound)."""
    with torch.no_grad():
        model.temporal.conv_z.bias.fill_(2e6)


@pytest.mark.parametrize("captured", [False, True])
def test_static_epoch_on_data_the_fold_refuses_is_rerun_in_the_reference_formulation(cuda, captured):
    """The epoch functions never read the status word per window (no host sync) and never raise: they keep a snapshot of
    parameters + Adam state, read the word once at the end of the epoch, restore, switch the folded formulation off, capture
    the window graph again and rerun.  Result == the same epochs with the folded formulation off from the start: costs, every
    parameter, the Adam state -- bit for bit (the same launches on the same values)."""
    import warnings
    from stgraph_amd import kernels, temporal
    n, e, T, B = 3000, 24000, 12, 4
    g, ew, targets, _ = _setup(cuda, n, e, T, 21)
    runs = []
    for folded_first in (True, False):
        torch.manual_seed(3)
        model = temporal.STGraphTGCN(32, 64, 1).to(cuda)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=True, fused=True)
        bucket = temporal.GradBucket(model.parameters())
        with _folded_on(cuda):
            kernels.set_step_folded(False), kernels.set_step_wgrad_from_p(False)
            temporal.train_epoch_static(model, g, ew, targets, B, opt, bucket, 32, epoch=0)      # a healthy epoch, the same in both runs
            kernels.set_step_folded(folded_first), kernels.set_step_wgrad_from_p(folded_first)
            cw = temporal.CapturedStaticWindow(model, g, ew, targets, B, opt, bucket, 32) if captured else None
            _trip(model)
            with warnings.catch_warnings(record=True) as rec:
                warnings.simplefilter("always")
                costs = []
                for ep in (1, 2):
                    if captured:
                        costs += temporal.train_epoch_static_captured(cw, model, g, ew, targets, opt, bucket, 32, epoch=ep)
                    else:
                        costs += temporal.train_epoch_static(model, g, ew, targets, B, opt, bucket, 32, epoch=ep)
            warned = [w for w in rec if "folded step formulation is switched off" in str(w.message)]
            assert len(warned) == (1 if folded_first else 0)            # once: the second epoch already runs the reference formulation
            assert not kernels.STEP_FOLDED and not kernels.STEP_WGRAD_FROM_P
            assert int(kernels.step_fold_status_word(cuda).item()) == 0
        state = [v.clone() for p in model.parameters() for v in opt.state[p].values() if torch.is_tensor(v)]
        runs.append((torch.stack([c.reshape(()) for c in costs]).clone(), [p.detach().clone() for p in model.parameters()], state))
    assert torch.isfinite(runs[0][0]).all()
    assert torch.equal(runs[0][0], runs[1][0])
    for k in (1, 2):
        for a, b in zip(runs[0][k], runs[1][k]):
            assert torch.equal(a, b)


def test_captured_dynamic_epoch_on_data_the_fold_refuses_is_rerun(cuda):
    """CapturedDynamicWindows: same contract; the tripped epoch is rerun eagerly, the window graphs are dropped and captured again
    (in the reference formulation) by the next captured epoch."""
    import warnings
    import numpy as np
    from stgraph_amd import kernels, temporal
    from stgraph_amd.graph import NaiveGraph
    n, e0, churn, T, B, feat, hid, m = 3000, 20000, 500, 9, 4, 32, 64, 1000
    rng = np.random.default_rng(5)
    stream = rng.choice(n * n, size=e0 + churn * T, replace=False)
    snaps, pn_edges, pn_targets = [], [], []
    gen = torch.Generator(device=cuda).manual_seed(5)
    for t in range(T):
        keys = stream[t * churn: t * churn + e0]
        s, d = (keys // n).astype(np.int32), (keys % n).astype(np.int32)
        snaps.append((torch.from_numpy(s).to(cuda), torch.from_numpy(d).to(cuda)))
        pos = torch.from_numpy(np.stack([s[:m], d[:m]]).astype(np.int64)).to(cuda)
        pn_edges.append(torch.cat([pos, torch.randint(0, n, (2, m), device=cuda, generator=gen)], 1))
        pn_targets.append(torch.cat([torch.ones(m, device=cuda), torch.zeros(m, device=cuda)]))
    runs = []
    for folded_first in (True, False):
        G = NaiveGraph(snaps, n, device=cuda, sort_inplace=False, resident=True)
        torch.manual_seed(4)
        model = temporal.DynamicSTGraphTGCN(feat, hid).to(cuda)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=True, fused=True)
        bucket = temporal.GradBucket(model.parameters())
        with _folded_on(cuda):
            kernels.set_step_folded(False), kernels.set_step_wgrad_from_p(False)
            temporal.train_epoch_dynamic(model, G, pn_edges, pn_targets, B, opt, bucket, feat, epoch=0)     # healthy, the same in both runs
            kernels.set_step_folded(folded_first), kernels.set_step_wgrad_from_p(folded_first)
            cd = temporal.CapturedDynamicWindows(model, G, pn_edges, pn_targets, B, opt, bucket, feat)
            _trip(model)
            # the FIRST captured epoch both captures (in the folded formulation, if on) and trips
            costs = []
            with warnings.catch_warnings(record=True) as rec:
                warnings.simplefilter("always")
                for ep in (1, 2):
                    costs += [c.clone() for c in temporal.train_epoch_dynamic_captured(cd, epoch=ep)]
            assert sum("folded step formulation is switched off" in str(w.message) for w in rec) == (1 if folded_first else 0)
            assert not kernels.STEP_FOLDED and len(cd.graphs) == 2                # captured again by epoch 2
        runs.append((torch.stack([c.reshape(()) for c in costs]), [p.detach().clone() for p in model.parameters()]))
    assert torch.isfinite(runs[0][0]).all()
    torch.testing.assert_close(runs[0][0], runs[1][0], rtol=1e-6, atol=0)
    for a, b in zip(runs[0][1], runs[1][1]):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)
