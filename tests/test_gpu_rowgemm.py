"""stg_rowgemm_f32 (one wave per 32-row tile, A operands from registers, W in LDS, fp32 MFMA) against an fp64
reference."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,K,M", [(1, 4, 32), (63, 8, 32), (64, 64, 128), (65, 128, 64), (5000, 192, 32),
                                   (50_000, 128, 64), (50_000, 64, 128), (20_001, 32, 192), (4097, 100, 96)])
@pytest.mark.parametrize("trans_w,use_bias", [(False, False), (True, True), (True, False)])
def test_matches_reference(cuda, N, K, M, trans_w, use_bias):
    from stgraph_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(N + K + M)
    x = torch.randn(N, K, device=cuda, generator=gen)
    w = torch.randn((M, K) if trans_w else (K, M), device=cuda, generator=gen)
    b = torch.randn(M, device=cuda, generator=gen) if use_bias else None
    got = kernels.rowgemm(x, w, b, trans_w=trans_w)
    wd = w.double().t() if trans_w else w.double()
    want = x.double() @ wd + (b.double() if use_bias else 0)
    scale = x.double().abs() @ wd.abs() + 1
    assert ((got.double() - want).abs() <= 2e-6 * scale).all()


def test_asymmetric_integer_data_exact(cuda):
    from stgraph_amd import kernels
    N, K, M = 300, 36, 64
    x = (torch.arange(N * K, device=cuda) % 7 - 3).float().view(N, K)
    w = (torch.arange(K * M, device=cuda) % 5 - 2).float().view(K, M)
    assert torch.equal(kernels.rowgemm(x, w), (x.double() @ w.double()).float())
    wt = (torch.arange(K * M, device=cuda) % 11 - 5).float().view(M, K)
    assert torch.equal(kernels.rowgemm(x, wt, trans_w=True), (x.double() @ wt.double().t()).float())


def test_unsupported_shapes(cuda):
    from stgraph_amd import _C, kernels
    assert not _C.lib.stg_rowgemm_supported(30, 64) and not _C.lib.stg_rowgemm_supported(64, 48)
    assert not _C.lib.stg_rowgemm_supported(256, 256) and not _C.lib.stg_rowgemm_supported(192, 192)
    assert _C.lib.stg_rowgemm_supported(192, 128) and _C.lib.stg_rowgemm_supported(128, 192)
    x = torch.zeros(10, 30, device=cuda)
    with pytest.raises(_C.StgError):
        kernels.rowgemm(x, torch.zeros(30, 64, device=cuda))


def test_wide_linear_in_column_slices(cuda):
    """[N, 64] -> 512 (GATConv's fc at cfg3) as four 128-column slices written into one output."""
    from stgraph_amd import kernels
    x = torch.randn(70_000, 64, device=cuda)
    w = torch.randn(512, 64, device=cuda)
    b = torch.randn(512, device=cuda)
    assert kernels.wide_linear_usable(x, w)
    for bias in (b, None):
        got = kernels.linear_fwd(x, w, bias)
        want = x.double() @ w.double().t() + (bias.double() if bias is not None else 0)
        scale = x.double().abs() @ w.double().abs().t() + 1
        assert ((got.double() - want).abs() <= 2e-6 * scale).all()
    assert not kernels.wide_linear_usable(x[:1000], w) and not kernels.wide_linear_usable(x, w[:200])


@pytest.mark.parametrize("N", [1, 17, 4096, 70_001])
@pytest.mark.parametrize("K,M", [(128, 128), (64, 128), (128, 64), (64, 64)])
@pytest.mark.parametrize("trans_w,use_bias,relu", [(False, True, True), (True, False, False), (False, False, True), (True, True, False)])
def test_row_piece_kernel_with_its_epilogue(cuda, N, K, M, trans_w, use_bias, relu):
    """stg_rowgemm_act_f32 (16-row tiles in the step kernels' row-piece layout; bias and ReLU in the epilogue): the dense
    layer of the GCN / GAT configs.  The last tile is partial at every N here but 4096; integer data is exact."""
    from stgraph_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(N + K + 2 * M)
    x = torch.randn(N, K, device=cuda, generator=gen)
    w = torch.randn((M, K) if trans_w else (K, M), device=cuda, generator=gen)
    b = torch.randn(M, device=cuda, generator=gen) if use_bias else None
    got = kernels.rowgemm_act(x, w, b, trans_w, kernels.ACT_RELU if relu else kernels.ACT_NONE)
    wd = w.double().t() if trans_w else w.double()
    want = x.double() @ wd + (b.double() if use_bias else 0)
    scale = x.double().abs() @ wd.abs() + 1
    if relu:
        want = want.relu()
    assert ((got.double() - want).abs() <= 2e-6 * scale).all()
    xi = (torch.arange(N * K, device=cuda) % 7 - 3).float().view(N, K)
    wi = (torch.arange(K * M, device=cuda) % 5 - 2).float().view(w.shape)
    wid = wi.double().t() if trans_w else wi.double()
    assert torch.equal(kernels.rowgemm_act(xi, wi, None, trans_w), (xi.double() @ wid).float())


@pytest.mark.parametrize("N", [1, 31, 4096, 70_001])
@pytest.mark.parametrize("K,M", [(128, 128), (64, 128), (128, 64), (64, 64)])
@pytest.mark.parametrize("trans_w,use_bias,relu", [(False, True, True), (True, False, False), (True, True, False)])
def test_split_form_on_the_matrix_cores(cuda, N, K, M, trans_w, use_bias, relu):
    """The same products as three-term bf16 splits on v_mfma_f32_16x16x32_bf16 (rowgemm_x3.hip; the default from 64 K rows,
    forced here at every N): the fp32 kernel's error bound against fp64, integers exact, and the plain (strided) entry."""
    from stgraph_amd import _C, kernels
    gen = torch.Generator(device=cuda).manual_seed(N + K + 2 * M)
    x = torch.randn(N, K, device=cuda, generator=gen) * torch.exp(3 * torch.randn(N, 1, device=cuda, generator=gen))
    w = torch.randn((M, K) if trans_w else (K, M), device=cuda, generator=gen)
    b = torch.randn(M, device=cuda, generator=gen) if use_bias else None
    wd = w.double().t() if trans_w else w.double()
    want = x.double() @ wd + (b.double() if use_bias else 0)
    scale = x.double().abs() @ wd.abs() + 1
    xi = (torch.arange(N * K, device=cuda) % 7 - 3).float().view(N, K)
    wi = (torch.arange(K * M, device=cuda) % 5 - 2).float().view(w.shape)
    wid = wi.double().t() if trans_w else wi.double()
    _C.set_tuning("rowgemm_x3", 2)
    try:
        got = kernels.rowgemm_act(x, w, b, trans_w, kernels.ACT_RELU if relu else kernels.ACT_NONE)
        goti = kernels.rowgemm_act(xi, wi, None, trans_w)
        y = torch.full((N, M + 8), 7.0, device=cuda)
        if N >= 4096:
            _C.check(_C.lib.stg_rowgemm_strided_f32(x.data_ptr(), w.data_ptr(), b.data_ptr() if use_bias else None, y.data_ptr(),
                                                    N, K, M, M + 8, int(trans_w), None))
            torch.cuda.synchronize()
    finally:
        _C.set_tuning("rowgemm_x3", 0)
    assert ((got.double() - (want.relu() if relu else want)).abs() <= 2e-6 * scale).all()
    assert torch.equal(goti, (xi.double() @ wid).float())
    if N >= 4096:
        assert ((y[:, :M].double() - want).abs() <= 2e-6 * scale).all() and bool((y[:, M:] == 7.0).all())


def test_gcn_training_step_with_the_split_products_matches_the_fp32_products(cuda):
    """A 2-layer GCN (128 -> 128 -> 128, the cfg2 widths) on 70 K vertices -- where the row products take the split form by
    default -- against the same step with them on the fp32 matrix instruction: loss and every gradient.  A first-layer bias
    of 30 keeps every pre-activation positive: no ReLU decision hangs on an fp32 rounding (tests/test_gpu_input_layer.py)."""
    import numpy as np
    from stgraph_amd import _C, kernels
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn import functional as SF
    from stgraph_amd.nn.pytorch.static.gcn_conv import GCNConv
    from tests.util import gcn_norm, random_graph
    n, f = 70_001, 128
    src, dst = random_graph(3, n, 600_000)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    g.set_ndata("norm", torch.from_numpy(gcn_norm(np.bincount(dst, minlength=n))).to(cuda))
    gen = torch.Generator(device=cuda).manual_seed(1)
    x = torch.randn(n, f, device=cuda, generator=gen)
    labels = torch.randint(0, f, (n,), device=cuda, generator=gen)
    res = []
    for knob in (0, 1):
        _C.set_tuning("rowgemm_x3", knob)
        try:
            torch.manual_seed(4)
            layers = torch.nn.ModuleList([GCNConv(f, f, torch.relu), GCNConv(f, f, None)]).to(cuda)
            with torch.no_grad():
                layers[0].bias.fill_(30.0)
            rec = []
            kernels.enable_launch_timing(rec)
            h = x
            for layer in layers:
                h = layer(g, h)
            loss = SF.cross_entropy(h, labels, n)
            loss.backward()
            kernels.enable_launch_timing(None)
        finally:
            kernels.enable_launch_timing(None)
            _C.set_tuning("rowgemm_x3", 0)
        assert any(r[0] == "rowgemm" for r in rec), {r[0] for r in rec}
        res.append((loss.detach().clone(), [p.grad.clone() for p in layers.parameters()]))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-6, atol=0)
    for a, b in zip(res[0][1], res[1][1]):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), (float((a - b).abs().max()), float(b.abs().max()))
