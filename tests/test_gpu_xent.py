"""Fused softmax cross-entropy (csrc/xent.hip, nn.functional.cross_entropy) against torch's F.cross_entropy in fp32
and an fp64 restatement: value, gradient, a loss that enters the cost with a factor, slices of a larger logits
matrix (the train-mask prefix of the GCN scripts), out-of-range labels reported through the status word."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 5, 1624, 70_001, 600_000])
@pytest.mark.parametrize("K", [2, 7, 16, 32, 33, 128, 1000])
def test_matches_torch(cuda, n, K):
    from stgraph_amd.nn import functional as SF
    if n * K > 100_000_000:
        pytest.skip("too large")
    g = torch.Generator(device=cuda).manual_seed(n + K)
    logits = torch.randn(n, K, device=cuda, generator=g) * 3
    if n > 2:
        logits[1] = 50.0                       # a saturated row
        logits[2, 0] = -80.0
    labels = torch.randint(0, K, (n,), device=cuda, generator=g)
    res = []
    for fn, dt in ((SF.cross_entropy, torch.float32), (F.cross_entropy, torch.float32), (F.cross_entropy, torch.float64)):
        x = logits.to(dt).clone().requires_grad_(True)
        loss = fn(x, labels)
        (loss * 0.37).backward()
        res.append((loss.detach(), x.grad))
    (l0, g0), (l1, g1), (l2, g2) = res
    for a, b, c, name in ((l0, l1, l2, "loss"), (g0, g1, g2, "grad")):
        scale = float(c.abs().max()) + 1e-30
        err_ours = float((a.double() - c).abs().max()) / scale
        err_torch = float((b.double() - c).abs().max()) / scale
        assert err_ours <= max(4 * err_torch, 2e-6), (name, err_ours, err_torch)


def test_prefix_slice_and_fallbacks(cuda):
    from stgraph_amd.nn import functional as SF
    g = torch.Generator(device=cuda).manual_seed(0)
    full = torch.randn(2708, 7, device=cuda, generator=g).requires_grad_(True)
    labels = torch.randint(0, 7, (2708,), device=cuda, generator=g)
    loss = SF.cross_entropy(full[:1624], labels[:1624])
    loss.backward()
    ref = full.detach().clone().requires_grad_(True)
    F.cross_entropy(ref[:1624], labels[:1624]).backward()
    torch.testing.assert_close(full.grad, ref.grad, rtol=1e-5, atol=1e-8)
    assert not full.grad[1624:].any()
    # double logits / int32 labels: torch's path
    x64 = torch.randn(10, 3, device=cuda, dtype=torch.float64)
    torch.testing.assert_close(SF.cross_entropy(x64, labels[:10] % 3), F.cross_entropy(x64, labels[:10] % 3))


def test_out_of_range_label_sets_the_status_word(cuda):
    from stgraph_amd import kernels
    logits = torch.randn(100, 5, device=cuda)
    labels = torch.randint(0, 5, (100,), device=cuda)
    *_, status = kernels.xent_fwd(logits, labels)
    assert int(status.item()) == 0
    labels[17] = 5
    *_, status = kernels.xent_fwd(logits, labels)
    assert int(status.item()) != 0
    with pytest.raises(kernels._C.StgError):
        kernels.check_xent_status(cuda)         # reads AND clears the sticky word
    assert kernels.xent_status(cuda) == 0


def test_ignore_index_and_bad_labels_match_torch(cuda):
    """ignore_index = -100 rows: no loss term, not in the denominator, zero gradient -- torch's semantics; another
    out-of-range label is treated the same way (torch asserts) and reported."""
    from stgraph_amd import kernels
    from stgraph_amd.nn import functional as SF
    g = torch.Generator(device=cuda).manual_seed(5)
    for n, K in ((100, 5), (5000, 128), (3, 7)):
        base = torch.randn(n, K, device=cuda, generator=g)
        labels = torch.randint(0, K, (n,), device=cuda, generator=g)
        labels[::3] = -100
        a = base.clone().requires_grad_(True)
        la = SF.cross_entropy(a, labels)
        (la * 1.7).backward()
        b = base.clone().requires_grad_(True)
        lb = F.cross_entropy(b, labels)
        (lb * 1.7).backward()
        torch.testing.assert_close(la, lb, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-4, atol=1e-9)
        assert not a.grad[::3].any()
        assert kernels.xent_status(cuda) == 0
        # a label that is neither a class nor ignore_index: same as ignored, but reported
        bad = labels.clone()
        bad[1] = K + 3
        ref = labels.clone()
        ref[1] = -100
        c = base.clone().requires_grad_(True)
        SF.cross_entropy(c, bad).backward()
        d = base.clone().requires_grad_(True)
        F.cross_entropy(d, ref).backward()
        torch.testing.assert_close(c.grad, d.grad, rtol=1e-4, atol=1e-9)
        assert kernels.xent_status(cuda) != 0 and kernels.xent_status(cuda) == 0
    # all rows ignored: 0 / 0, as torch
    x = torch.randn(4, 3, device=cuda)
    assert torch.isnan(SF.cross_entropy(x, torch.full((4,), -100, device=cuda, dtype=torch.int64)))


def test_labels_on_another_device_do_not_reach_the_kernel(cuda):
    from stgraph_amd import kernels
    from stgraph_amd.nn import functional as SF
    x = torch.randn(6, 3, device=cuda)
    with pytest.raises(ValueError):
        kernels.xent_fwd(x, torch.zeros(6, dtype=torch.int64))
    with pytest.raises(Exception):              # F.cross_entropy's own device check, not a GPU fault
        SF.cross_entropy(x, torch.zeros(6, dtype=torch.int64))


@pytest.mark.parametrize("n_total,rows,K", [(2708, 1624, 7), (100_000, 60_000, 128), (50, 50, 3), (10, 1, 5)])
def test_row_prefix_without_slicing(cuda, n_total, rows, K):
    """`rows`: the loss on logits[:rows] with the whole gradient matrix (zero tail) from the one backward launch."""
    from stgraph_amd.nn import functional as SF
    g = torch.Generator(device=cuda).manual_seed(rows)
    base = torch.randn(n_total, K, device=cuda, generator=g)
    labels = torch.randint(0, K, (n_total,), device=cuda, generator=g)
    a = base.clone().requires_grad_(True)
    la = SF.cross_entropy(a, labels, rows)
    (la * 0.5).backward()
    b = base.clone().requires_grad_(True)
    lb = F.cross_entropy(b[:rows], labels[:rows])
    (lb * 0.5).backward()
    torch.testing.assert_close(la, lb, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(a.grad, b.grad, rtol=1e-4, atol=1e-9)
    assert not a.grad[rows:].any()


@pytest.mark.parametrize("n_total,n,K", [(1000, 600, 128), (70_001, 70_001, 128), (5000, 4999, 8), (3000, 100, 256), (2708, 1624, 16)])
def test_backward_with_column_sums(cuda, n_total, n, K):
    """stg_xent_bwd_colsum: the same gradient as stg_xent_bwd bit for bit, and its column sums (the bias gradient of a layer
    right below the loss) to fp32 rounding of a 64-bit sum; twice the same result (fixed summation order)."""
    from stgraph_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(n_total + K)
    logits = torch.randn(n_total, K, device=cuda, generator=gen)
    labels = torch.randint(0, K, (n_total,), device=cuda, generator=gen)
    labels[::7] = -100
    loss, lse, n_counted, _ = kernels.xent_fwd(logits, labels, n)
    g = torch.tensor([0.7], device=cuda)
    plain = kernels.xent_bwd(g, logits, labels, lse, n_counted)
    d, cs = kernels.xent_bwd(g, logits, labels, lse, n_counted, want_colsum=True)
    assert cs is not None and torch.equal(d, plain)
    want = plain.double().sum(0)
    assert ((cs.double() - want).abs() <= 1e-5 * plain.double().abs().sum(0) + 1e-12).all()
    d2, cs2 = kernels.xent_bwd(g, logits, labels, lse, n_counted, want_colsum=True)
    assert torch.equal(cs, cs2)


def test_bias_gradient_of_the_layer_below_the_loss_comes_from_the_loss_backward(cuda):
    """A GCNConv (bias, no activation) right below SF.cross_entropy: its bias gradient is the column sums the loss's backward
    left on the gradient tensor, and equals the separate pass over the matrix."""
    import numpy as np
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn import functional as SF
    from stgraph_amd.nn.pytorch.static.gcn_conv import GCNConv
    from tests.util import gcn_norm, random_graph
    n, e, fin, K = 3000, 30000, 32, 16
    src, dst = random_graph(5, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    g.set_ndata("norm", torch.from_numpy(gcn_norm(np.bincount(dst, minlength=n))).to(cuda))
    torch.manual_seed(0)
    conv = GCNConv(fin, K).to(cuda)
    x = torch.randn(n, fin, device=cuda)
    labels = torch.randint(0, K, (n,), device=cuda)
    calls = []
    real = kernels.bias_act_bwd

    def counting(*a, **k):
        calls.append(1)
        return real(*a, **k)
    kernels.bias_act_bwd = counting
    try:
        SF.cross_entropy(conv(g, x), labels, 1800).backward()
        fused = conv.bias.grad.clone()
        assert not calls                                       # no separate pass for the bias gradient
        conv.zero_grad()
        logits = conv(g, x)
        torch.nn.functional.cross_entropy(logits[:1800], labels[:1800]).backward()
        assert calls
    finally:
        kernels.bias_act_bwd = real
    torch.testing.assert_close(fused, conv.bias.grad, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("n_total,n,K", [(1000, 600, 128), (70_001, 70_001, 128), (5000, 4999, 8), (3000, 100, 256), (2708, 1624, 16),
                                         (1, 1, 4), (600_000, 360_000, 128)])
def test_forward_and_gradient_in_one_pass(cuda, n_total, n, K):
    """stg_xent_fwd_grad: loss, lse, count, gradient and its column sums from ONE pass over the logits (with labels that are
    ignored) against the forward launch followed by the backward launch: lse and the count bit-equal, the loss up to the order of
    its partial sums, the gradient up to the rounding of exp(x - m) / sum against exp(x - lse); stg_xent_scale_grad for g != 1."""
    from stgraph_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(n_total + K)
    logits = torch.randn(n_total, K, device=cuda, generator=gen) * 3
    labels = torch.randint(0, K, (n_total,), device=cuda, generator=gen)
    if n > 10:
        labels[3] = -100
        labels[n - 2] = -100
    assert kernels.xent_fwd_grad_usable(logits)
    loss2, lse2, cnt2, _ = kernels.xent_fwd(logits, labels, n)
    one = torch.ones(1, device=cuda)
    d2, cs2 = kernels.xent_bwd(one, logits, labels, lse2, cnt2, want_colsum=True)
    loss1, lse1, cnt1, _, d1, cs1 = kernels.xent_fwd_grad(logits, labels, n)
    assert torch.equal(lse1, lse2) and torch.equal(cnt1, cnt2)
    torch.testing.assert_close(loss1, loss2, rtol=2e-6, atol=0)
    # exp(x - m) / sum here, exp(x - lse) there: fp32 rounding of lse apart
    assert float((d1 - d2).abs().max()) <= 2e-6 * float(d2.abs().max())
    torch.testing.assert_close(cs1, cs2, rtol=2e-5, atol=1e-6 * float(d2.abs().max()) * (n ** 0.5))
    assert not d1[n:].any()
    # g = 1: the scale launch leaves every bit alone; g = 0.37: the gradient of 0.37 * loss
    keep = d1.clone()
    kernels.xent_scale_grad(d1, cs1, one)
    assert torch.equal(d1, keep)
    g = torch.full((1,), 0.37, device=cuda)
    kernels.xent_scale_grad(d1, cs1, g)
    d3, cs3 = kernels.xent_bwd(g, logits, labels, lse2, cnt2, want_colsum=True)
    assert float((d1 - d3).abs().max()) <= 2e-6 * float(d3.abs().max())
    torch.testing.assert_close(cs1, cs3, rtol=2e-5, atol=1e-6 * float(d3.abs().max()) * (n ** 0.5))


def test_one_pass_loss_backward_twice_and_switched_off(cuda):
    """SF.cross_entropy takes the one-pass form when the logits need a gradient; a second backward through the same graph
    (retain_graph) falls back to the backward launch; kernels.set_xent_one_pass(False) gives the same gradient."""
    from stgraph_amd import kernels
    from stgraph_amd.nn import functional as SF
    gen = torch.Generator(device=cuda).manual_seed(5)
    logits = torch.randn(5000, 64, device=cuda, generator=gen)
    labels = torch.randint(0, 64, (5000,), device=cuda, generator=gen)
    grads = []
    for on in (True, False):
        kernels.set_xent_one_pass(on)
        try:
            x = logits.clone().requires_grad_(True)
            rec = []
            kernels.enable_launch_timing(rec)
            loss = SF.cross_entropy(x, labels, 3000)
            loss.backward(retain_graph=True)
            first = x.grad.clone()
            loss.backward()
            kernels.enable_launch_timing(None)
            assert float((x.grad - 2 * first).abs().max()) <= 4e-6 * float(first.abs().max())
            assert ("xent_fwd_grad" in {r[0] for r in rec}) == on
            grads.append(first)
            # no gradient wanted: the plain forward
            rec = []
            kernels.enable_launch_timing(rec)
            SF.cross_entropy(logits, labels, 3000)
            kernels.enable_launch_timing(None)
            assert {r[0] for r in rec} == {"xent_fwd"}
        finally:
            kernels.enable_launch_timing(None)
            kernels.set_xent_one_pass(True)
    assert float((grads[0] - grads[1]).abs().max()) <= 2e-6 * float(grads[1].abs().max())


@pytest.mark.parametrize("n_total,n,K", [(2708, 1624, 7), (1, 1, 1), (500, 500, 64), (8192, 5000, 4), (5, 3, 33), (2000, 1, 13)])
def test_small_matrix_one_launch_each_way(cuda, n_total, n, K):
    """stg_xent_small_fwd / _bwd (a matrix one workgroup covers: Cora's 2708 x 7) against the general launches on the same inputs:
    lse, loss, gradient and column sums to fp32 rounding (another order of the same
    additions), the tail rows zero, ignore_index rows out of everything, a bad label in the status word; run twice: identical."""
    from stgraph_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(n_total + K)
    logits = torch.randn(n_total, K, device=cuda, generator=gen) * 3
    labels = torch.randint(0, K, (n_total,), device=cuda, generator=gen)
    if n > 4:
        labels[::5] = -100
    assert kernels.xent_small_usable(logits)
    loss, lse, cnt, status = kernels.xent_small_fwd(logits, labels, n)
    loss_g, lse_g, cnt_g, _ = kernels.xent_fwd(logits, labels, n)
    assert int(status.item()) == 0 and float(cnt) == float(cnt_g)
    torch.testing.assert_close(lse, lse_g, rtol=1e-6, atol=2e-6)
    torch.testing.assert_close(loss, loss_g, rtol=1e-5, atol=1e-7)
    g = torch.tensor([0.7], device=cuda)
    d, cs = kernels.xent_small_bwd(g, logits, labels, lse, cnt)
    want = kernels.xent_bwd(g, logits, labels, lse, cnt)
    torch.testing.assert_close(d, want, rtol=1e-6, atol=1e-9)
    assert not d[n:].any()
    ref = want.double().sum(0)
    assert ((cs.double() - ref).abs() <= 1e-5 * want.double().abs().sum(0) + 1e-12).all()
    d2, cs2 = kernels.xent_small_bwd(g, logits, labels, lse, cnt)
    loss2, *_ = kernels.xent_small_fwd(logits, labels, n)
    assert torch.equal(d, d2) and torch.equal(cs, cs2) and torch.equal(loss, loss2)
    if K > 1:
        labels[0] = K                                   # neither a class nor ignore_index
        *_, status = kernels.xent_small_fwd(logits, labels, n)
        assert int(status.item()) != 0
        with pytest.raises(kernels._C.StgError):
            kernels.check_xent_status(cuda)


def test_small_matrix_path_is_taken_by_the_loss_and_can_be_switched_off(cuda):
    from stgraph_amd import kernels
    from stgraph_amd.nn import functional as SF
    gen = torch.Generator(device=cuda).manual_seed(3)
    base = torch.randn(2708, 7, device=cuda, generator=gen)
    labels = torch.randint(0, 7, (2708,), device=cuda, generator=gen)
    out = []
    for on in (True, False):
        kernels.set_xent_small(on)
        try:
            recs = []
            kernels.enable_launch_timing(recs)
            x = base.clone().requires_grad_(True)
            (SF.cross_entropy(x, labels, 1624) * 1.3).backward()
            kernels.enable_launch_timing(None)
            names = [r[0] for r in recs]
            assert (names == ["xent_small_fwd", "xent_small_bwd"]) == on, names
            out.append(x.grad)
        finally:
            kernels.enable_launch_timing(None)
            kernels.set_xent_small(True)
    torch.testing.assert_close(out[0], out[1], rtol=1e-5, atol=1e-9)
    big = torch.randn(70_000, 7, device=cuda)
    assert not kernels.xent_small_usable(big) and not kernels.xent_small_usable(torch.randn(10, 65, device=cuda))
    with pytest.raises(kernels._C.StgError):
        kernels.xent_small_fwd(big, torch.zeros(70_000, dtype=torch.int64, device=cuda))


@pytest.mark.parametrize("N,F,mask", [(2708, 16, True), (2708, 7, False), (100, 33, True), (4096, 16, True), (1, 4, False)])
def test_bias_act_bwd_of_a_small_matrix_finishes_its_column_sums_itself(cuda, N, F, mask):
    """stg_bias_act_bwd with N F <= 65536: one workgroup, the column sums written by the same launch (no finish launch): masked
    gradient bit for bit, column sums against a 64-bit sum."""
    from stgraph_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(N + F)
    g = torch.randn(N, F, device=cuda, generator=gen)
    out = torch.randn(N, F, device=cuda, generator=gen) if mask else None
    g_act, cs = kernels.bias_act_bwd(g, out, want_colsum=True)
    want = g * (out > 0) if mask else g
    assert torch.equal(g_act, want)
    ref = want.double().sum(0)
    assert ((cs.double() - ref).abs() <= 1e-5 * want.double().abs().sum(0) + 1e-12).all()
    _, cs2 = kernels.bias_act_bwd(g, out, want_colsum=True)
    assert torch.equal(cs, cs2)
