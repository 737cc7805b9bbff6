"""stg_gcn_agg_transform: (A_hat x) W in one kernel == A_hat (x W) within fp32 rounding; P = A_hat x
bit-identical to gcn_agg; backward through the autograd wrapper == the two-kernel formulation."""
import numpy as np
import pytest
import torch

from oracle import stg_oracle as orc
from tests.util import gcn_norm, random_graph

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fin,fout", [(16, 32), (32, 192), (32, 64), (48, 96), (64, 128), (20, 160)])
@pytest.mark.parametrize("use_ew,nid", [(False, False), (True, False), (True, True)])
def test_matches_two_kernel_form_and_oracle(cuda, fin, fout, use_ew, nid):
    from stgraph_amd import kernels
    n, e = 3001, 41000
    src, dst = random_graph(fin * 7 + fout, n, e)
    g = kernels.build_graph_csr(src, dst, n, cuda)
    og = orc.build_graph(src, dst, n)
    rng = np.random.default_rng(fin)
    x = rng.standard_normal((n, fin)).astype(np.float32)
    W = (rng.standard_normal((fin, fout)) / np.sqrt(fin)).astype(np.float32)
    norm = gcn_norm(og.in_degrees())
    ew = rng.uniform(0.5, 1.5, (len(src), 1)).astype(np.float32) if use_ew else None
    t = lambda a: None if a is None else torch.from_numpy(a).to(cuda)  # noqa: E731
    out, P = kernels.gcn_agg_transform(t(x), t(W), t(norm), t(norm), g.fwd, ew=t(ew), use_node_ids=nid)
    P0 = orc.gcn_agg(x, norm, norm, og.fwd, ew=ew, use_node_ids=nid)
    assert np.array_equal(P.cpu().numpy(), P0)                           # the aggregation part is bit-exact
    want = P0.astype(np.float64) @ W.astype(np.float64)
    scale = np.abs(P0).astype(np.float64) @ np.abs(W).astype(np.float64)
    assert (np.abs(out.cpu().numpy() - want) <= 1e-6 * scale + 1e-6).all()
    ref = kernels.gcn_agg(t(x) @ t(W), t(norm), t(norm), g.fwd, ew=t(ew), use_node_ids=nid)   # the layer's order
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)
    # launch shapes (rows / waves per workgroup) only move work around: same bits
    from stgraph_amd import _C
    try:
        for rows, waves in ((32, 4), (32, 8), (64, 8)):
            _C.set_tuning("xw_rows", rows)
            _C.set_tuning("xw_waves", waves)
            out2, P2 = kernels.gcn_agg_transform(t(x), t(W), t(norm), t(norm), g.fwd, ew=t(ew), use_node_ids=nid)
            assert torch.equal(out2, out) and torch.equal(P2, P), (rows, waves)
    finally:
        _C.set_tuning("xw_rows", 0)
        _C.set_tuning("xw_waves", 0)


def test_unsupported_shapes_are_reported(cuda):
    from stgraph_amd import kernels
    assert not kernels.agg_transform_supported(7, 32) and not kernels.agg_transform_supported(32, 48)
    assert not kernels.agg_transform_supported(128, 128)                  # W would not fit the LDS budget
    assert kernels.agg_transform_supported(32, 192)


def test_autograd_wrapper_matches_unfused_layer(cuda):
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn import functional as SF
    from stgraph_amd.nn.pytorch.static.gcn_conv import GCNConv
    n, e, fin, fout = 4000, 52000, 32, 96
    src, dst = random_graph(91, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    f = g.csr("fwd")
    deg = (f.row_offset[1:] - f.row_offset[:-1]).float()
    g.set_ndata("norm", torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg)).unsqueeze(1))
    torch.manual_seed(0)
    conv = GCNConv(fin, fout, bias=False).to(cuda)
    x = torch.randn(n, fin, device=cuda, requires_grad=True)
    ew = torch.rand(len(src), 1, device=cuda) + 0.5
    R = torch.randn(n, fout, device=cuda)
    assert SF.agg_transform_usable(g, x, conv.weight)
    (SF.agg_transform(g, x, conv.weight, ew) * R).sum().backward()
    gx, gw = x.grad.clone(), conv.weight.grad.clone()
    x.grad = None
    conv.weight.grad = None
    out = conv(g, x, ew)
    (out * R).sum().backward()
    torch.testing.assert_close(gx, x.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gw, conv.weight.grad, rtol=1e-4, atol=1e-3)
