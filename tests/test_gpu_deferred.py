"""nn.deferred (one weight-gradient launch per parameter per backward pass): a backward() that raises must not leave
its accumulator behind for the next pass; compiler.executor: a forward that no backward can follow keeps nothing."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


class _Boom(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        raise RuntimeError("boom")


def test_failed_backward_does_not_poison_the_next_pass(cuda):
    from stgraph_amd.nn import deferred
    from stgraph_amd.nn import functional as SF
    torch.manual_seed(0)
    x = torch.randn(8192, 48, device=cuda)
    R = torch.randn(8192, 24, device=cuda)
    w = torch.randn(24, 48, device=cuda, requires_grad=True)
    b = torch.randn(24, device=cuda, requires_grad=True)

    def loss(through_boom):
        h = x.clone().requires_grad_(True)
        y = SF.linear(_Boom.apply(h) if through_boom else h, w, b)       # _Linear registers (w, b) with the accumulator
        return (y * R).sum()

    with pytest.raises(RuntimeError, match="boom"):
        loss(True).backward()                   # the Linear's backward has run and registered; the pass then dies
    assert w.grad is None and b.grad is None    # the engine skipped the queued flush
    loss(False).backward()                      # must get a fresh accumulator AND a fresh flush callback
    ref_w = torch.nn.Parameter(w.detach().clone())
    ref_b = torch.nn.Parameter(b.detach().clone())
    (torch.nn.functional.linear(x, ref_w, ref_b) * R).sum().backward()
    torch.testing.assert_close(w.grad, ref_w.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(b.grad, ref_b.grad, rtol=1e-4, atol=1e-3)
    assert deferred._active is None
    # and once more: nothing doubled, nothing withheld
    w.grad = b.grad = None
    loss(False).backward()
    torch.testing.assert_close(w.grad, ref_w.grad, rtol=1e-4, atol=1e-3)


def test_forward_without_backward_keeps_no_executor_state(cuda):
    """Evaluation under torch.no_grad() (the eval pass of the GCN / GAT scripts) through the @compile path: the
    reference pushes one entry per call and only backward pops (executor.py:236-259); here such a call keeps none."""
    import numpy as np
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn.pytorch.static.gcn_conv import GCNConv
    rng = np.random.default_rng(0)
    n = 200
    keys = rng.choice(n * n, size=1500, replace=False)
    g = StaticGraph(np.stack([keys // n, keys % n], 1).astype(np.int32), None, n, device=cuda)
    deg = g.in_degrees().astype(np.float32)
    g.set_ndata("norm", torch.from_numpy(np.where(deg > 0, deg, 1) ** -0.5).float().view(-1, 1).to(cuda))
    conv = GCNConv(12, 8, activation=None, bias=False).to(cuda)       # no bias, no activation: the @compile path
    x = torch.randn(n, 12, device=cuda)
    ref = conv(g, x.clone().requires_grad_(True))
    stacks = _stacks_of(conv)
    depth = [len(s) for s in stacks]
    with torch.no_grad():
        for _ in range(5):
            out = conv(g, x)
    assert torch.equal(out, ref.detach())
    assert [len(s) for s in stacks] == depth
    ref.sum().backward()                        # the tracked call still finds its entry
    assert all(len(s) == 0 for s in stacks)
    for p in conv.parameters():
        p.requires_grad_(False)
    for _ in range(3):
        conv(g, x)                              # grad mode on, but no differentiable input
    assert all(len(s) == 0 for s in stacks)


def _stacks_of(module):
    """tensor_map_stack / graph_timestamp_stack of every executor reachable from the layer's compiled functions."""
    from stgraph_amd.compiler.executor import Executor
    found, seen = [], set()

    def walk(o, depth):
        if id(o) in seen or depth > 8 or isinstance(o, types.ModuleType):
            return
        seen.add(id(o))
        if isinstance(o, Executor):
            found.extend([o.ts.tensor_map_stack, o.ts.graph_timestamp_stack])
            return
        if isinstance(o, dict):
            for v in o.values():
                walk(v, depth + 1)
        elif isinstance(o, (list, tuple)):
            for v in o:
                walk(v, depth + 1)
        elif hasattr(o, "__dict__") and not isinstance(o, torch.Tensor):
            for v in vars(o).values():
                walk(v, depth + 1)
    walk(module, 0)
    assert found, "no executor found behind the layer"
    return found
