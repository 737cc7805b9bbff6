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


def _decode_bits(bits, N, M):
    """[N, M] bool from the words of stg_rowgemm_act_bits_f32, by the layout include/stgraph_hip.h states."""
    dev = bits.device
    row = torch.arange(N, device=dev).view(N, 1)
    col = torch.arange(M, device=dev).view(1, M)
    word = 64 * (row >> 4) + (row & 7) + 8 * ((col >> 4) & 1) + 16 * ((col >> 2) & 3)
    bit = 8 * (col >> 5) + 4 * ((row >> 3) & 1) + (col & 3)
    return ((bits.long()[word] >> bit) & 1).bool()


@pytest.mark.parametrize("N", [1, 31, 33, 4096, 70_001])
@pytest.mark.parametrize("K,M", [(128, 128), (64, 128), (128, 64), (64, 64)])
def test_relu_sign_pattern_as_bits(cuda, N, K, M):
    """stg_rowgemm_act_bits_f32: the ReLU forward leaves [y > 0] as one bit per element (the documented layout), and the
    transposed product of the layer above multiplies by it -- both bit-equal to the launches without the pattern."""
    from stgraph_amd import _C, kernels
    gen = torch.Generator(device=cuda).manual_seed(N + K + 3 * M)
    x = torch.randn(N, K, device=cuda, generator=gen)
    w = torch.randn(K, M, device=cuda, generator=gen)
    b = torch.randn(M, device=cuda, generator=gen)
    g = torch.randn(N, K, device=cuda, generator=gen)           # gradient of the layer above: [N, K2] with K2 = K here
    w2 = torch.randn(M, K, device=cuda, generator=gen)          # that layer's weight, [in = M][out = K]
    assert _C.lib.stg_rowgemm_bits_supported(N, K, M) and _C.lib.stg_rowgemm_bits_words(N) == ((N + 31) // 32) * 128
    _C.set_tuning("rowgemm_x3", 2)
    try:
        y_plain = kernels.rowgemm_act(x, w, b, False, kernels.ACT_RELU)
        gx_plain = kernels.rowgemm_act(g, w2, None, True)
        y, bits = kernels.rowgemm_relu_bits(x, w, b)
        gx = kernels.rowgemm_masked_t(g, w2, bits)
    finally:
        _C.set_tuning("rowgemm_x3", 0)
    assert torch.equal(y, y_plain)
    assert torch.equal(_decode_bits(bits, N, M), y > 0)
    assert torch.equal(gx, gx_plain * (y > 0))
    # argument checks
    with pytest.raises(RuntimeError):
        _C.check(_C.lib.stg_rowgemm_act_bits_f32(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), N, K, M, 0, kernels.ACT_RELU, None, None, None))
    with pytest.raises(RuntimeError):
        _C.check(_C.lib.stg_rowgemm_act_bits_f32(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), N, K, M, 1, kernels.ACT_RELU, None,
                                                 bits.data_ptr(), None))
    assert not _C.lib.stg_rowgemm_bits_supported(N, 96, M)


def test_gcn_training_step_with_the_relu_pattern_as_bits(cuda):
    """The 2-layer GCN of cfg2's widths on 70 K vertices with real ReLU decisions: the input layer leaves its sign pattern as
    bits, the layer above masks its input gradient with them, the input layer's weight gradient is the plain contraction --
    against the same step with the pattern re-read from the layer's output (kernels.set_relu_bits(False)).  The masks are
    the same bits either way; the gradients differ by the rounding of two split-K orders."""
    import numpy as np
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn import functional as SF
    from stgraph_amd.nn.pytorch.static.gcn_conv import GCNConv
    from tests.util import gcn_norm, random_graph
    n, f = 70_001, 128
    src, dst = random_graph(5, n, 600_000)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    g.set_ndata("norm", torch.from_numpy(gcn_norm(np.bincount(dst, minlength=n))).to(cuda))
    gen = torch.Generator(device=cuda).manual_seed(2)
    x = torch.randn(n, f, device=cuda, generator=gen)
    labels = torch.randint(0, f, (n,), device=cuda, generator=gen)
    res = []
    for on in (True, False):
        kernels.set_relu_bits(on)
        try:
            torch.manual_seed(4)
            layers = torch.nn.ModuleList([GCNConv(f, f, torch.relu), GCNConv(f, f, None)]).to(cuda)
            rec = []
            kernels.enable_launch_timing(rec)
            h1 = layers[0](g, x)
            h = layers[1](g, h1)
            loss = SF.cross_entropy(h, labels, n)
            loss.backward()
        finally:
            kernels.enable_launch_timing(None)
            kernels.set_relu_bits(True)
        masked_form = 4 * n * (2 * f + f) + 4 * f * f               # bytes of gemm_tn_relu_mask's record
        assert any(r[0] == "gemm_tn" and r[3] == masked_form for r in rec) == (not on), [(r[0], r[3]) for r in rec]
        assert float((h1 == 0).float().mean()) > 0.2                # the ReLU does decide
        res.append((loss.detach().clone(), h1.detach().clone(), [p.grad.clone() for p in layers.parameters()]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2], res[1][2]):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), (float((a - b).abs().max()), float(b.abs().max()))


def test_relu_pattern_is_dropped_when_the_output_is_written_again(cuda):
    """The pattern rides on the tensor object with its version: an in-place write of the ReLU output between the layers
    (here `h1 += 1`) makes the layer above fall back to the plain input gradient, and the result is still right."""
    from stgraph_amd.nn import functional as SF
    from stgraph_amd import kernels
    n, f = 70_016, 128
    gen = torch.Generator(device=cuda).manual_seed(3)
    x = torch.randn(n, f, device=cuda, generator=gen)
    w1 = torch.randn(f, f, device=cuda, generator=gen).requires_grad_()
    h1, bits = kernels.rowgemm_relu_bits(x, w1.detach(), None)
    SF._tag(h1, "_stg_relu_bits", bits)
    assert SF._tagged(h1, "_stg_relu_bits") is bits
    h1 += 1
    assert SF._tagged(h1, "_stg_relu_bits") is None


@pytest.mark.parametrize("N", [1, 33, 70_001])
@pytest.mark.parametrize("K,M,heads", [(64, 64, 8), (128, 64, 2), (64, 128, 3)])
def test_per_head_products_of_a_wide_matrix(cuda, N, K, M, heads):
    """stg_rowgemm_heads_f32: Y[h] = X[:, h K : (h + 1) K] @ W[h] for the column blocks of X [N, heads K] (rows read with the
    wide matrix's stride) against fp64."""
    from stgraph_amd import _C
    gen = torch.Generator(device=cuda).manual_seed(N + K + M + heads)
    x = torch.randn(N, heads * K, device=cuda, generator=gen)
    w = torch.randn(heads, K, M, device=cuda, generator=gen)
    y = torch.empty(heads, N, M, device=cuda)
    assert _C.lib.stg_rowgemm_heads_supported(N, K, M, heads)
    _C.check(_C.lib.stg_rowgemm_heads_f32(x.data_ptr(), w.data_ptr(), y.data_ptr(), N, K, M, heads, None))
    torch.cuda.synchronize()
    xd = x.double().view(N, heads, K).transpose(0, 1)
    want = torch.bmm(xd, w.double())
    scale = torch.bmm(xd.abs(), w.double().abs()) + 1
    assert ((y.double() - want).abs() <= 2e-6 * scale).all()
