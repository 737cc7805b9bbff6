"""The model head + per-timestep loss of the static-temporal TGCN step as one launch each way
(csrc/tgcn_head.hip, nn.functional.tgcn_head) against the torch composition the reference's script spells out
(benchmarking/static-temporal-tgcn/seastar/model.py:6-18: relu -> Linear -> Linear; train loop: mean squared error),
in fp32 and against an fp64 restatement."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _operands(cuda, N, C, seed, dtype=torch.float32):
    g = torch.Generator(device=cuda).manual_seed(seed)
    r = lambda *s: torch.randn(*s, device=cuda, generator=g)  # noqa: E731
    h = r(N, C)
    if N > 2:
        h[1, :3] = 0.0                              # relu's kink: gradient 0 at exactly 0
    W1, b1, W2, b2 = r(32, C) * 0.3, r(32), r(1, 32) * 0.3, r(1)
    target, gy = r(N, 1), r(N, 32) * 0.1
    return [t.to(dtype) for t in (h, W1, b1, W2, b2, target, gy)]


def _composition(h, W1, b1, W2, b2, target):
    y = F.linear(F.relu(h), W1, b1)
    y_out = F.linear(y, W2, b2)
    return y, y_out, torch.mean((y_out - target) ** 2)


def _run(fn, ops, gscale):
    h, W1, b1, W2, b2, target, gy = ops
    leaves = [t.clone().requires_grad_(True) for t in (h, W1, b1, W2, b2)]
    y, y_out, loss = fn(*leaves, target)
    # y feeds the next step (gradient gy) and the loss enters the cost with a factor
    (loss * gscale + (y * gy).sum()).backward()
    return [y.detach(), y_out.detach(), loss.detach()] + [t.grad for t in leaves]


NAMES = ("y", "y_out", "loss", "dh", "dW1", "db1", "dW2", "db2")


@pytest.mark.parametrize("N", [1, 15, 17, 31, 32, 33, 1000, 4097, 50_000])
@pytest.mark.parametrize("C", [32, 64, 128])
@pytest.mark.parametrize("defer,rows", [(True, 32), (False, 32), (True, 16)])
def test_fused_head_matches_torch_composition(cuda, N, C, defer, rows):
    from stgraph_amd import _C
    from stgraph_amd.nn import functional as SF
    ops = _operands(cuda, N, C, 7 * N + C)
    assert SF.tgcn_head_usable(*ops[:6])
    SF.set_deferred_weight_grads(defer)
    _C.set_tuning("cell_rows", rows)             # 32-row (v_mfma_f32_32x32x2_f32) / 16-row (16x16x4) tiles
    try:
        got = _run(SF.tgcn_head, ops, 1.0 / 26)
    finally:
        SF.set_deferred_weight_grads(True)
        _C.set_tuning("cell_rows", 0)
    want = _run(_composition, ops, 1.0 / 26)
    want64 = _run(_composition, [t.double() for t in ops], 1.0 / 26)
    for name, a, b, c in zip(NAMES, got, want, want64):
        assert a.shape == b.shape, name
        # fp32 vs fp32: both carry rounding of a K = C (resp. N) long sum; judge both against fp64
        scale = float(c.abs().max()) + 1e-30
        err_ours = float((a.double() - c).abs().max()) / scale
        err_torch = float((b.double() - c).abs().max()) / scale
        # (db2 is a sum of N signed terms: cancellation makes its own magnitude a small scale)
        assert err_ours <= max(4 * err_torch, 2e-5 if name == "db2" else 2e-6), (name, err_ours, err_torch)
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * scale, msg=lambda m, n=name: f"{n}: {m}")


def test_unused_outputs_and_missing_gradients(cuda):
    """Only the loss is used (y has no other consumer): g_y arrives as None."""
    from stgraph_amd.nn import functional as SF
    h, W1, b1, W2, b2, target, _ = _operands(cuda, 777, 64, 3)
    res = []
    for fn in (SF.tgcn_head, _composition):
        leaves = [t.clone().requires_grad_(True) for t in (h, W1, b1, W2, b2)]
        _, _, loss = fn(*leaves, target)
        loss.backward()
        res.append([t.grad for t in leaves])
    for name, a, b in zip(NAMES[3:], *res):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6, msg=lambda m, n=name: f"{n}: {m}")


def test_y_out_consumer_gets_its_gradient(cuda):
    from stgraph_amd.nn import functional as SF
    h, W1, b1, W2, b2, target, _ = _operands(cuda, 1234, 32, 5)
    res = []
    for fn in (SF.tgcn_head, _composition):
        leaves = [t.clone().requires_grad_(True) for t in (h, W1, b1, W2, b2)]
        _, y_out, loss = fn(*leaves, target)
        (loss + (y_out * target).sum() * 1e-3).backward()
        res.append([t.grad for t in leaves])
    for name, a, b in zip(NAMES[3:], *res):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6, msg=lambda m, n=name: f"{n}: {m}")


def test_unsupported_shapes_fall_back_to_the_composition(cuda):
    from stgraph_amd.nn import functional as SF
    g = torch.Generator(device=cuda).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=cuda, generator=g)  # noqa: E731
    h, W1, b1, W2, b2, target = r(100, 48), r(16, 48), r(16), r(2, 16), r(2), r(100, 2)
    assert not SF.tgcn_head_usable(h, W1, b1, W2, b2, target)
    y, y_out, loss = SF.tgcn_head(h, W1, b1, W2, b2, target)
    y2, y_out2, loss2 = _composition(h, W1, b1, W2, b2, target)
    torch.testing.assert_close(y_out, y_out2, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(loss, loss2, rtol=1e-4, atol=1e-6)


def test_training_epoch_same_with_and_without_the_fused_head(cuda):
    """Whole static-temporal epochs (eager): losses and parameters agree with the unfused head."""
    from stgraph_amd import temporal
    from stgraph_amd.graph import StaticGraph
    from tests.util import random_graph
    n, e, feat, hid, T, B = 5000, 40000, 32, 64, 12, 4
    src, dst = random_graph(11, n, e)
    e = len(src)
    out = []
    for fused in (True, False):
        temporal.set_fused_head(fused)
        try:
            g = StaticGraph((src.copy(), dst.copy()), None, n, device=cuda, sort_inplace=False)
            g.set_ndata("norm", temporal.in_degree_norm(g))
            gen = torch.Generator(device=cuda).manual_seed(1)
            ew = torch.rand(e, 1, device=cuda, generator=gen) + 0.5
            targets = torch.randn(T, n, 1, device=cuda, generator=gen)
            torch.manual_seed(5)
            model = temporal.STGraphTGCN(feat, hid, 1).to(cuda)
            # plain SGD: parameter differences then scale with gradient differences (Adam's normalised step turns
            # rounding noise on near-zero gradients into lr-sized differences)
            opt = torch.optim.SGD(model.parameters(), lr=1e-2)
            bucket = temporal.GradBucket(model.parameters())
            losses = []
            for ep in range(2):
                losses += temporal.train_epoch_static(model, g, ew, targets, B, opt, bucket, feat, epoch=ep)
            out.append((torch.stack(losses), [p.detach().clone() for p in model.parameters()]))
        finally:
            temporal.set_fused_head(True)
    torch.testing.assert_close(out[0][0], out[1][0], rtol=2e-4, atol=1e-6)
    for a, b in zip(out[0][1], out[1][1]):
        torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-4)


def test_running_cost_is_folded_into_the_loss_kernel(cuda):
    """`cost = cost + loss` of the training loop inside the launch: two chained heads, value and every gradient equal
    to the spelled-out additions."""
    from stgraph_amd.nn import functional as SF
    ops_a, ops_b = _operands(cuda, 3000, 64, 1), _operands(cuda, 3000, 64, 2)
    res = []
    for fused in (True, False):
        la = [t.clone().requires_grad_(True) for t in ops_a[:5]]
        lb = [t.clone().requires_grad_(True) for t in ops_b[:5]]
        if fused:
            _, _, cost = SF.tgcn_head(*la, ops_a[5])
            _, _, cost = SF.tgcn_head(*lb, ops_b[5], cost=cost)
        else:
            cost = _composition(*la, ops_a[5])[2] + _composition(*lb, ops_b[5])[2]
        (cost / 3).backward()
        res.append([cost.detach()] + [t.grad for t in la + lb])
    for a, b in zip(*res):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)
