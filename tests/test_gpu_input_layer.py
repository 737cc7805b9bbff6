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


@pytest.mark.parametrize("bias0", ["positive", "kink_free"])
@pytest.mark.parametrize("fin,hid,out,use_ew", [(128, 128, 128, False), (24, 64, 7, True), (16, 16, 5, False)])
def test_reordered_input_layer_matches_the_reference_order(cuda, fin, hid, out, use_ew, bias0):
    """Both orders on the same GPU, outputs and every gradient to 1e-4 of the tensor's largest entry.  ``positive``: a first-layer
    bias of 30 keeps every pre-activation positive (no ReLU decision at all).  ``kink_free``: a per-column bias that leaves
    no pre-activation within 2e-5 of zero (``_kink_free_bias``), so the ReLU mask is a real mix of zeros and ones but cannot
    depend on the order of an fp32 sum.  (With pre-activations ON the kink two correct fp32 evaluations flip individual
    ReLUs -- measured at the full cfg2 shape: 4.4e-4 of a gradient's size, 8.2e-4 for the reference order itself,
    profiles/r03_input_layer_error.json; that is a property of the inputs, not a tolerance of this test.)"""
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
    if bias0 == "kink_free":
        probe = _model(fin, hid, out, 5, cuda)
        with torch.no_grad():
            pre = kernels.gcn_agg(x @ probe[0].weight, g.get_ndata("norm"), g.get_ndata("norm"), g.csr("fwd"), ew=ew)
        bias_np = _kink_free_bias(pre.cpu().numpy())
    for reorder in (True, False):
        SF.set_input_layer_reorder(reorder)
        try:
            layers = _model(fin, hid, out, 5, cuda)
            with torch.no_grad():
                if bias0 == "positive":
                    layers[0].bias.fill_(30.0)
                else:
                    layers[0].bias.copy_(torch.from_numpy(bias_np).to(cuda))
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
        err = float((a - b).abs().max() / (b.abs().max() + 1e-30))
        assert err <= 1e-4, (name, err)


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


def _kink_free_bias(pre: np.ndarray, margin: float = 2e-5) -> np.ndarray:
    """Per column, a bias that puts zero in the middle of a gap >= 2 * margin of that column's pre-activations: after the
    shift no ReLU input lies within ``margin`` of the kink, so the ReLU mask cannot depend on fp32 summation order."""
    bias = np.zeros(pre.shape[1], np.float32)
    for c in range(pre.shape[1]):
        v = np.sort(pre[:, c].astype(np.float64))
        gaps = v[1:] - v[:-1]
        ok = np.nonzero(gaps >= 2.5 * margin)[0]
        mids = (v[ok] + v[ok + 1]) / 2
        bias[c] = -mids[np.argmin(np.abs(mids))]
    return bias


@pytest.mark.parametrize("use_ew", [False, True])
def test_aggregate_first_layer_against_the_reference_order_oracle(cuda, use_ew):
    """The bench-default path (aggregate first, bias + ReLU in the GEMM epilogue, ReLU-masked split-K weight gradient, no
    aggregation in the layer's backward) against the ORACLE evaluated in the reference's order (x W, emitted aggregation,
    bias, ReLU: nn/pytorch/static/gcn_conv.py:158-188) with torch-CPU autograd around it -- strictly: outputs to 1e-5,
    every gradient to 1e-4 of its largest entry.  The first layer's bias is chosen so that no pre-activation lies within
    2e-5 of zero (``_kink_free_bias``): with ties present the two ORDERS OF SUMMATION flip individual ReLUs -- measured at
    the full cfg2 shape in profiles/r03_input_layer_error.json: 4.4e-4 of the gradient's size this way round, 8.2e-4 with
    the GPU itself in reference order."""
    from oracle import stg_oracle as orc
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn import functional as SF
    from tests.oracle_layers import OracleGCNConv, OracleGraphView
    n, e, fin, hid, out = 20000, 320000, 128, 128, 128
    src, dst = random_graph(77, n, e)
    e = len(src)
    og = OracleGraphView(src, dst, n)
    norm_np = gcn_norm(og.in_degrees())
    og.set_ndata("norm", torch.from_numpy(norm_np))
    g = StaticGraph((src.copy(), dst.copy()), None, n, device=cuda, sort_inplace=False)
    g.set_ndata("norm", torch.from_numpy(norm_np).to(cuda))
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((n, fin)).astype(np.float32))
    ew = torch.from_numpy((rng.random((e, 1)) + 0.5).astype(np.float32)) if use_ew else None
    R = torch.from_numpy(rng.standard_normal((n, out)).astype(np.float32)) / n
    layers = _model(fin, hid, out, 9, cuda)
    cpu = torch.nn.ModuleList([OracleGCNConv(fin, hid, torch.relu), OracleGCNConv(hid, out, None)])
    with torch.no_grad():
        pre = orc.gcn_agg((x @ layers[0].weight.cpu()).numpy(), norm_np, norm_np, og.g.fwd,
                          ew=None if ew is None else ew.numpy())
        layers[0].bias.copy_(torch.from_numpy(_kink_free_bias(pre)).to(cuda))
        assert np.abs(pre + layers[0].bias.cpu().numpy()).min() >= 2e-5
        for a, b in zip(cpu.parameters(), layers.parameters()):
            a.copy_(b.cpu())
    xg, ewg = x.to(cuda), None if ew is None else ew.to(cuda)
    assert SF.input_layer_usable(g, xg, layers[0].weight, layers[0].activation)
    got = layers[1](g, layers[0](g, xg, ewg), ewg)
    got.backward(R.to(cuda))
    want = cpu[1](og, cpu[0](og, x, ew), ew)
    want.backward(R)
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-5, atol=1e-5)
    for (k, a), b in zip(cpu.named_parameters(), layers.parameters()):
        w = a.grad.numpy()
        err = np.abs(b.grad.cpu().numpy() - w).max() / np.abs(w).max()
        assert err <= 1e-4, (k, float(err))


@pytest.mark.parametrize("observe", ["retain_grad", "hook"])
def test_hidden_gradient_observed_by_the_caller_is_torch_s(cuda, observe):
    """The layer above an _InputLayer may multiply its input gradient by the ReLU's sign bits in the launch that forms it -- the
    gradient of the PRE-activation.  That is only invisible while nobody looks at the hidden tensor's own gradient: with
    ``hidden.retain_grad()`` or a hook on it, ``hidden.grad`` must be ``g @ W^T`` unmasked, as torch gives it (nonzero where the
    hidden value is 0), and the parameter gradients must not change."""
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    n, f = 70_000, 128
    src, dst = random_graph(3, n, 400_000)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    g.set_ndata("norm", torch.from_numpy(gcn_norm(np.bincount(dst, minlength=n))).to(cuda))
    gen = torch.Generator(device=cuda).manual_seed(1)
    x = torch.randn(n, f, device=cuda, generator=gen)
    R = torch.randn(n, f, device=cuda, generator=gen)
    layers = _model(f, f, f, 2, cuda)
    seen = {}

    def run(watch):
        layers.zero_grad()
        hidden = layers[0](g, x)
        if watch == "retain_grad":
            hidden.retain_grad()
        elif watch == "hook":
            hidden.register_hook(lambda gr: seen.__setitem__("g", gr.clone()))
        rec = []
        kernels.enable_launch_timing(rec)
        try:
            layers[1](g, hidden).backward(R)
        finally:
            kernels.enable_launch_timing(None)
        return hidden, [p.grad.clone() for p in layers.parameters()]

    assert kernels.rowgemm_bits_usable(R, f, f)
    hidden0, grads0 = run(None)                                    # nobody looks: the masked launch may run
    hidden, grads = run(observe)
    got = hidden.grad if observe == "retain_grad" else seen["g"]
    with torch.no_grad():
        agg = kernels.gcn_agg(R, g.get_ndata("norm"), g.get_ndata("norm"), g.csr("bwd"))
        want = agg @ layers[1].weight.t()                          # d loss / d hidden of layers[1] = A_hat^T R W^T
    dead = hidden.detach() == 0
    assert dead.float().mean() > 0.2
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    assert float(got[dead].abs().max()) > 0                        # NOT masked where the ReLU is off
    for a, b in zip(grads0, grads):
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-9
