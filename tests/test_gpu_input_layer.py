"""GCNConv on an input that carries no gradient: aggregate first, then the weight product (functional._InputLayer) --
the layer's backward then has no aggregation.  Against the reference order (x W first; reference
nn/pytorch/static/gcn_conv.py:158-188) on the same inputs: outputs and every parameter gradient to fp32 rounding, and the
launch counts that make it worthwhile."""
import numpy as np
import pytest
import torch

from tests.util import gcn_norm, random_graph

pytestmark = pytest.mark.gpu


def _model(fin, hid, out, seed, cuda):
    from stgraph_amd.nn.pytorch.static.gcn_conv import GCNConv
    torch.manual_seed(seed)
    return torch.nn.ModuleList([GCNConv(fin, hid, torch.relu), GCNConv(hid, out, None)]).to(cuda)


@pytest.mark.parametrize("bias0", [30.0, 0.0])
@pytest.mark.parametrize("fin,hid,out,use_ew", [(128, 128, 128, False), (24, 64, 7, True), (16, 16, 5, False)])
def test_reordered_input_layer_matches_the_reference_order(cuda, fin, hid, out, use_ew, bias0):
    """``bias0`` = 30: every pre-activation of the first layer is positive, so the two orders must agree to fp32 rounding
    everywhere.  ``bias0`` = 0: a pre-activation within rounding of zero may land on either side of the ReLU (as it
    may between any two fp32 evaluations, the reference's FMA build and its no-FMA emulation included), which moves a
    whole column of the weight gradient by one row's contribution: outputs strictly, gradients in norm."""
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn import functional as SF
    n = 5000
    src, dst = random_graph(fin + out, n, 60000)
    e = len(src)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    deg = np.bincount(dst, minlength=n)
    g.set_ndata("norm", torch.from_numpy(gcn_norm(deg)).to(cuda))
    rng = np.random.default_rng(fin)
    x = torch.from_numpy(rng.standard_normal((n, fin)).astype(np.float32)).to(cuda)
    ew = torch.from_numpy((rng.random((e, 1)) + 0.5).astype(np.float32)).to(cuda) if use_ew else None
    R = torch.from_numpy(rng.standard_normal((n, out)).astype(np.float32)).to(cuda)
    res, aggs = [], []
    for reorder in (True, False):
        SF.set_input_layer_reorder(reorder)
        try:
            layers = _model(fin, hid, out, 5, cuda)
            with torch.no_grad():
                layers[0].bias.fill_(bias0)
            assert SF.input_layer_usable(g, x, layers[0].weight, layers[0].activation) == reorder
            rec = []
            kernels.enable_launch_timing(rec)
            h = x
            for layer in layers:
                h = layer(g, h, ew)
            h.backward(R)
            torch.cuda.synchronize()
            kernels.enable_launch_timing(None)
        finally:
            SF.set_input_layer_reorder(True)
        aggs.append(sum(1 for r in rec if r[0].startswith("gcn_agg") or r[0].startswith("gcn_layer")))
        res.append([h.detach().clone()] + [p.grad.clone() for p in layers.parameters()])
    assert aggs == [3, 4], (aggs, [r[0] for r in rec])    # one aggregation fewer per training step
    names = ["out"] + [n for n, _ in layers.named_parameters()]
    for name, a, b in zip(names, res[0], res[1]):
        if bias0 or name == "out":
            torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5 * float(b.abs().max() + 1),
                                       msg=lambda m, name=name: f"{name}: {m}")
        else:
            assert float((a - b).norm() / (b.norm() + 1e-30)) < 2e-3, name


def test_an_input_that_needs_a_gradient_keeps_the_reference_order(cuda):
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn import functional as SF
    n, e = 300, 2000
    src, dst = random_graph(3, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    g.set_ndata("norm", torch.from_numpy(gcn_norm(np.bincount(dst, minlength=n))).to(cuda))
    layers = _model(8, 8, 4, 1, cuda)
    x = torch.randn(n, 8, device=cuda, requires_grad=True)
    assert not SF.input_layer_usable(g, x, layers[0].weight, layers[0].activation)
    layers[1](g, layers[0](g, x)).sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    # a layer that narrows (in > out) transforms first whatever its input
    assert not SF.input_layer_usable(g, x.detach(), torch.empty(8, 4, device=cuda), None)


def test_bias_act_fwd_kernel(cuda):
    from stgraph_amd import kernels
    for n, f in ((1000, 128), (777, 7), (5, 16)):
        y = torch.randn(n, f, device=cuda)
        b = torch.randn(f, device=cuda)
        want = torch.relu(y + b)
        assert torch.equal(kernels.bias_act_fwd_(y.clone(), b, kernels.ACT_RELU), want)
        assert torch.equal(kernels.bias_act_fwd_(y.clone(), b, kernels.ACT_NONE), y + b)
        assert torch.equal(kernels.bias_act_fwd_(y.clone(), None, kernels.ACT_RELU), torch.relu(y))
