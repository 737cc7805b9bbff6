"""GCNConv's tail (``+ bias``, activation; reference nn/pytorch/static/gcn_conv.py:185-188) fused into the
aggregation kernel's store, and its one-pass backward (ReLU mask + bias gradient): bit-identical to the
unfused sequence of the same layer, and equal to the oracle's aggregation followed by numpy's add / maximum."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import stg_oracle as orc
from tests.util import gcn_norm, random_graph

pytestmark = pytest.mark.gpu


def _graph(cuda, n, e, seed):
    from stgraph_amd.graph import StaticGraph
    src, dst = random_graph(seed, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    og = orc.build_graph(src, dst, n)
    norm = gcn_norm(og.in_degrees())
    g.set_ndata("norm", torch.from_numpy(norm).to(cuda))
    return g, og, norm


@pytest.mark.parametrize("Fo", [7, 16, 64, 128, 300])
@pytest.mark.parametrize("act", [None, "relu"])
@pytest.mark.parametrize("use_ew", [False, True])
def test_fused_tail_equals_unfused_layer_and_oracle(cuda, Fo, act, use_ew):
    from stgraph_amd.nn import functional as SF
    from stgraph_amd.nn.pytorch.static.gcn_conv import GCNConv
    n, e, Fi = 1500, 20000, 24
    g, og, norm = _graph(cuda, n, e, 5)
    torch.manual_seed(Fo)
    conv = GCNConv(Fi, Fo, activation=F.relu if act else None).to(cuda)
    with torch.no_grad():
        conv.bias.copy_(torch.randn(Fo))
    w = (torch.rand(e, 1, device=cuda) + 0.5) if use_ew else None
    x0 = torch.randn(n, Fi, device=cuda)
    R = torch.randn(n, Fo, device=cuda)
    res = []
    for fused in (True, False):
        x = x0.clone().requires_grad_(True)
        conv.zero_grad()
        if fused:
            assert SF.gcn_layer_tail_usable(g, x @ conv.weight, conv.activation)
            out = conv(g, x, w)
        else:                                     # the layer as the reference writes it
            h = conv.aggregate(g, SF.mm(x, conv.weight), w) + conv.bias
            out = conv.activation(h) if conv.activation else h
        (out * R).sum().backward()
        torch.cuda.synchronize()
        res.append((out.detach().clone(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()))
    (o1, gx1, gw1, gb1), (o2, gx2, gw2, gb2) = res
    assert torch.equal(o1, o2) and torch.equal(gx1, gx2) and torch.equal(gw1, gw2)
    torch.testing.assert_close(gb1, gb2, rtol=1e-4, atol=1e-4 * float(gb2.abs().max() + 1))
    # oracle: sequential aggregation, then numpy's add and maximum
    h = (x0 @ conv.weight).detach().cpu().numpy()
    want = orc.gcn_agg(h, norm, norm, og.fwd, ew=None if w is None else w.cpu().numpy()) + conv.bias.detach().cpu().numpy()
    if act:
        want = np.maximum(want, 0)
    assert np.array_equal(o1.cpu().numpy(), want.astype(np.float32))


@pytest.mark.parametrize("N,Fo", [(1, 1), (5, 7), (1000, 16), (100003, 100), (40000, 128), (3000, 300), (2000, 1024),
                                  (500, 1500)])
@pytest.mark.parametrize("masked", [False, True])
def test_bias_act_bwd_kernel(cuda, N, Fo, masked):
    from stgraph_amd import kernels
    torch.manual_seed(N + Fo)
    g = torch.randn(N, Fo, device=cuda)
    out = torch.relu(torch.randn(N, Fo, device=cuda)) if masked else None
    ga, cs = kernels.bias_act_bwd(g, out, want_colsum=True)
    want = g * (out > 0) if masked else g
    assert torch.equal(ga, want)
    ref = want.double().sum(0)
    torch.testing.assert_close(cs.double(), ref, rtol=1e-5, atol=2e-6 * float(want.abs().sum(0).max() + 1))
    ga2, cs2 = kernels.bias_act_bwd(g, out, want_colsum=True)
    assert torch.equal(cs, cs2)                                 # fixed reduction order
    if masked:
        ga3, none = kernels.bias_act_bwd(g, out, want_colsum=False)
        assert none is None and torch.equal(ga3, want)


def test_bias_act_bwd_unaligned_rows_and_wide_limit(cuda):
    from stgraph_amd import _C, kernels
    base = torch.randn(1000 * 12 + 1, device=cuda)
    g = base[1:].view(1000, 12)                                # 4-byte aligned only
    out = torch.relu(torch.randn(1000, 12, device=cuda))
    ga, cs = kernels.bias_act_bwd(g, out)
    assert torch.equal(ga, g * (out > 0))
    torch.testing.assert_close(cs, (g * (out > 0)).sum(0), rtol=1e-4, atol=1e-4)
    with pytest.raises(_C.StgError):
        kernels.bias_act_bwd(torch.randn(4, 5001, device=cuda), None)


def test_tail_at_bench_scale_properties(cuda):
    """|V| = 1M, |E| = 16M, F = 128: epilogue == separate ops on the un-fused aggregation (bit-exact), masked
    gradient and bias gradient against torch."""
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    n, e, Fo = 1_000_000, 16_000_000, 128
    src, dst = random_graph(1, n, e, hub=False)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    f = g.csr("fwd")
    deg = (f.row_offset[1:] - f.row_offset[:-1]).float()
    norm = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg)).unsqueeze(1)
    x = torch.randn(n, Fo, device=cuda)
    b = torch.randn(Fo, device=cuda)
    plain = kernels.gcn_agg(x, norm, norm, f)
    fused = kernels.gcn_agg(x, norm, norm, f, bias=b, act=kernels.ACT_RELU)
    assert torch.equal(fused, torch.relu(plain + b))
    gr = torch.randn(n, Fo, device=cuda)
    ga, cs = kernels.bias_act_bwd(gr, fused)
    assert torch.equal(ga, gr * (fused > 0))
    torch.testing.assert_close(cs.double(), ga.double().sum(0), rtol=1e-5, atol=1e-2)


@pytest.mark.parametrize("second_first", [False, True])
def test_bias_gradient_with_a_second_consumer_of_the_logits(cuda, second_first):
    """The cross-entropy backward leaves the gradient's column sums on the gradient tensor for the bias layer below it
    (nn/functional.py: `_stg_colsum`).  When the logits have a SECOND consumer (a regulariser here) autograd adds the two
    gradients -- in place or into a new tensor -- and the sums no longer describe what the layer receives: the layer must
    notice (version counter / storage address) and re-read the matrix.  Checked against the torch composition of the same
    loss, with the regulariser's node created before and after the loss's (both accumulation orders)."""
    from stgraph_amd.nn import functional as SF
    from stgraph_amd.nn.pytorch.static.gcn_conv import GCNConv
    n, e, Fi, K = 1500, 20000, 24, 7
    g, og, norm = _graph(cuda, n, e, 9)
    torch.manual_seed(3)
    conv = GCNConv(Fi, K, activation=None).to(cuda)
    x = torch.randn(n, Fi, device=cuda)
    labels = torch.randint(0, K, (n,), device=cuda)
    res = []
    for fused in (True, False):
        conv.zero_grad()
        logits = conv(g, x)
        if second_first:
            reg = 0.3 * logits.pow(2).mean()
            loss = (SF.cross_entropy(logits, labels) if fused else F.cross_entropy(logits, labels)) + reg
        else:
            loss = (SF.cross_entropy(logits, labels) if fused else F.cross_entropy(logits, labels)) + 0.3 * logits.pow(2).mean()
        loss.backward()
        res.append((conv.bias.grad.clone(), conv.weight.grad.clone()))
    (gb1, gw1), (gb2, gw2) = res
    torch.testing.assert_close(gb1, gb2, rtol=1e-4, atol=1e-5 * float(gb2.abs().max()))
    torch.testing.assert_close(gw1, gw2, rtol=1e-4, atol=1e-5 * float(gw2.abs().max()))
    # and alone (the fast path still taken): the loss's own column sums are the bias gradient
    conv.zero_grad()
    SF.cross_entropy(conv(g, x), labels).backward()
    gb_alone = conv.bias.grad.clone()
    conv.zero_grad()
    F.cross_entropy(conv(g, x), labels).backward()
    torch.testing.assert_close(gb_alone, conv.bias.grad, rtol=1e-4, atol=1e-5 * float(conv.bias.grad.abs().max()))
