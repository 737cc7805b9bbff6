"""The link-prediction head of the dynamic-temporal harness as fused launches (csrc/tgcn_head.hip:
stg_link_head_fwd / _bwd, nn.functional.link_head) against the torch composition the reference's script spells out
(benchmarking/dynamic-temporal-tgcn/seastar/model.py:5-21: relu -> Linear; decode = (z[src] * z[dst]).sum(-1);
train loop: BCEWithLogitsLoss), in fp32 and against an fp64 restatement; the node-sorted incidence list against numpy."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _operands(cuda, N, C, M, seed, dtype=torch.float32):
    g = torch.Generator(device=cuda).manual_seed(seed)
    r = lambda *s: torch.randn(*s, device=cuda, generator=g)  # noqa: E731
    h = r(N, C)
    if N > 2:
        h[1, :3] = 0.0
    W1, b1 = r(32, C) * 0.3, r(32)
    ei = torch.randint(0, N, (2, M), device=cuda, generator=g)
    if M > 4:
        ei[:, 0] = 0                     # a self pair
        ei[0, 1:4] = N - 1               # a node with several incident label edges, both roles
        ei[1, 2:5] = N - 1
    target = (torch.rand(M, device=cuda, generator=g) > 0.5).float()
    gy = r(N, 32) * 0.1
    return [t.to(dtype) if t.is_floating_point() else t for t in (h, W1, b1, ei, target, gy)]


def _composition(h, W1, b1, ei, target):
    y = F.linear(F.relu(h), W1, b1)
    out = (y[ei[0]] * y[ei[1]]).sum(dim=-1).view(-1)
    return y, F.binary_cross_entropy_with_logits(out, target)


def _run(fn, ops, gscale):
    h, W1, b1, ei, target, gy = ops
    leaves = [t.clone().requires_grad_(True) for t in (h, W1, b1)]
    y, loss = fn(*leaves, ei, target)
    (loss * gscale + (y * gy).sum()).backward()
    return [y.detach(), loss.detach()] + [t.grad for t in leaves]


NAMES = ("y", "loss", "dh", "dW1", "db1")


@pytest.mark.parametrize("N,M", [(1, 1), (33, 7), (1000, 5000), (25_000, 20_000)])
@pytest.mark.parametrize("C", [32, 64, 128])
@pytest.mark.parametrize("defer,rows", [(True, 32), (False, 32), (True, 16)])
def test_fused_link_head_matches_torch_composition(cuda, N, M, C, defer, rows):
    from stgraph_amd import _C
    from stgraph_amd.nn import functional as SF
    ops = _operands(cuda, N, C, M, 3 * N + C + M)
    assert SF.link_head_usable(*ops[:5])
    SF.set_deferred_weight_grads(defer)
    _C.set_tuning("cell_rows", rows)
    try:
        got = _run(SF.link_head, ops, 1.0 / 21)
        again = _run(SF.link_head, ops, 1.0 / 21)
    finally:
        SF.set_deferred_weight_grads(True)
        _C.set_tuning("cell_rows", 0)
    want = _run(_composition, ops, 1.0 / 21)
    want64 = _run(_composition, [t.double() if t.is_floating_point() else t for t in ops], 1.0 / 21)
    for name, a, a2, b, c in zip(NAMES, got, again, want, want64):
        assert a.shape == b.shape, name
        assert torch.equal(a, a2), f"{name}: not reproducible"          # no atomics anywhere
        scale = float(c.abs().max()) + 1e-30
        err_ours = float((a.double() - c).abs().max()) / scale
        err_torch = float((b.double() - c).abs().max()) / scale
        assert err_ours <= max(4 * err_torch, 3e-6), (name, err_ours, err_torch)
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * scale, msg=lambda m, n=name: f"{n}: {m}")


def test_incidence_list_against_numpy(cuda):
    from stgraph_amd import kernels
    N, M = 500, 3000
    rng = np.random.default_rng(0)
    ei_np = rng.integers(0, N, (2, M))
    row_ptr, other, eid = kernels.link_incidence(torch.from_numpy(ei_np).to(cuda), N)
    nodes = np.concatenate([ei_np[0], ei_np[1]])
    order = np.argsort(nodes, kind="stable")
    assert np.array_equal(other.cpu().numpy(), np.concatenate([ei_np[1], ei_np[0]])[order])
    assert np.array_equal(eid.cpu().numpy(), order % M)
    assert np.array_equal(row_ptr.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(nodes, minlength=N))]))


def test_only_the_loss_is_used_and_fallback(cuda):
    from stgraph_amd.nn import functional as SF
    h, W1, b1, ei, target, _ = _operands(cuda, 777, 64, 900, 3)
    res = []
    for fn in (SF.link_head, _composition):
        leaves = [t.clone().requires_grad_(True) for t in (h, W1, b1)]
        _, loss = fn(*leaves, ei, target)
        loss.backward()
        res.append([t.grad for t in leaves])
    for name, a, b in zip(NAMES[2:], *res):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6, msg=lambda m, n=name: f"{n}: {m}")
    # a width the kernels do not cover -> the composition itself
    g = torch.Generator(device=cuda).manual_seed(0)
    h2, W2, b2 = torch.randn(50, 48, device=cuda, generator=g), torch.randn(16, 48, device=cuda, generator=g), torch.zeros(16, device=cuda)
    ei2 = torch.randint(0, 50, (2, 30), device=cuda, generator=g)
    t2 = torch.ones(30, device=cuda)
    assert not SF.link_head_usable(h2, W2, b2, ei2, t2)
    y, loss = SF.link_head(h2, W2, b2, ei2, t2)
    y_ref, loss_ref = _composition(h2, W2, b2, ei2, t2)
    torch.testing.assert_close(loss, loss_ref)


def test_dynamic_training_epochs_same_with_and_without_the_fused_head(cuda):
    from stgraph_amd import temporal
    from stgraph_amd.graph import NaiveGraph
    n, e0, churn, T, B, feat, hid, m = 4000, 30000, 800, 12, 4, 32, 64, 1500
    rng = np.random.default_rng(4)
    stream = rng.choice(n * n, size=e0 + churn * T, replace=False)
    out = []
    for fused in (True, False):
        temporal.set_fused_head(fused)
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
            G = NaiveGraph(snaps, n, device=cuda, sort_inplace=False)
            torch.manual_seed(4)
            model = temporal.DynamicSTGraphTGCN(feat, hid).to(cuda)
            # plain SGD: parameter differences then scale with gradient differences (Adam's normalised step turns
            # rounding noise on near-zero gradients into lr-sized differences)
            opt = torch.optim.SGD(model.parameters(), lr=1e-2)
            bucket = temporal.GradBucket(model.parameters())
            losses = []
            for ep in range(2):
                losses += temporal.train_epoch_dynamic(model, G, pn_edges, pn_targets, B, opt, bucket, feat, epoch=ep)
            out.append((torch.stack(losses), [p.detach().clone() for p in model.parameters()]))
        finally:
            temporal.set_fused_head(True)
    torch.testing.assert_close(out[0][0], out[1][0], rtol=2e-4, atol=1e-6)
    for a, b in zip(out[0][1], out[1][1]):
        torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-4)


def test_decoder_of_a_whole_window_in_one_launch(cuda):
    """stg_link_decode_fwd_multi: logits and loss partials of every snapshot of a window, bit for bit what stg_link_decode_fwd
    gives snapshot by snapshot (36 snapshots: two launches of 32 and 4)."""
    from stgraph_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(11)
    B, N, M = 36, 5000, 3001
    ys = [torch.randn(N, 32, device=cuda, generator=gen) for _ in range(B)]
    edges = [torch.randint(0, N, (2, M), device=cuda, generator=gen) for _ in range(B)]
    targets = [(torch.rand(M, device=cuda, generator=gen) < 0.5).float() for _ in range(B)]
    parts = (M + 31) // 32
    lo = [torch.empty(M, device=cuda) for _ in range(B)]
    pa = [torch.empty(parts, device=cuda) for _ in range(B)]
    kernels.link_decode_fwd_window(ys, edges, targets, lo, pa)
    for t in range(B):
        l1, p1 = torch.empty(M, device=cuda), torch.empty(parts, device=cuda)
        kernels.link_decode_fwd(ys[t], edges[t], targets[t], l1, p1)
        assert torch.equal(lo[t], l1) and torch.equal(pa[t], p1), t
