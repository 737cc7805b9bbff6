"""GATConv's attention projections fused (csrc/gat_proj.hip) and the one-node GAT layer built on them, against
torch's own ops for reference nn/pytorch/static/gat_conv.py:43-45 and against the compiled-vertex-function path."""
import pytest
import torch

from tests.util import random_graph

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,H,D", [(1, 1, 4), (7, 2, 4), (1000, 8, 8), (5000, 8, 64), (3000, 4, 256), (2049, 16, 64),
                                   (257, 1, 64), (40_000, 8, 64)])
def test_projection_kernels_match_torch(cuda, N, H, D):
    from stgraph_amd import kernels
    assert kernels.gat_proj_supported(H, D)
    g = torch.Generator(device=cuda).manual_seed(N + H + D)
    r = lambda *s: torch.randn(*s, device=cuda, generator=g)  # noqa: E731
    feat, al, ar = r(N, H, D), r(H, D), r(H, D)
    el, er = kernels.gat_proj_fwd(feat, al, ar)
    torch.testing.assert_close(el, (feat * al).sum(-1, keepdim=True), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(er, (feat * ar).sum(-1, keepdim=True), rtol=1e-5, atol=1e-5)
    d_el, d_er, gin = r(N, H, 1), r(N, H, 1), r(N, H, D)
    want_f = gin + d_el * al + d_er * ar
    want_l = (d_el.double() * feat.double()).sum(0)
    want_r = (d_er.double() * feat.double()).sum(0)
    for inplace in (False, True):
        gsrc = gin.clone()
        df, dl, dr = kernels.gat_proj_bwd(feat, al, ar, d_el, d_er, gsrc, inplace=inplace)
        assert (df.data_ptr() == gsrc.data_ptr()) == inplace
        torch.testing.assert_close(df, want_f, rtol=1e-6, atol=1e-6)
        scale = float((d_el.abs().double() * feat.abs().double()).sum(0).max()) + 1
        torch.testing.assert_close(dl.double(), want_l, rtol=1e-5, atol=2e-6 * scale)
        torch.testing.assert_close(dr.double(), want_r, rtol=1e-5, atol=2e-6 * scale)
    df0, dl0, _ = kernels.gat_proj_bwd(feat, al, ar, d_el, d_er, None)
    torch.testing.assert_close(df0, d_el * al + d_er * ar, rtol=1e-6, atol=1e-6)
    dl1 = kernels.gat_proj_bwd(feat, al, ar, d_el, d_er, None)[1]
    assert torch.equal(dl0, dl1)                                   # fixed reduction order


def test_unsupported_shapes_fall_back(cuda):
    from stgraph_amd import kernels
    assert not kernels.gat_proj_supported(8, 7) and not kernels.gat_proj_supported(8, 48)
    assert not kernels.gat_proj_supported(3, 64) or 3 * 64 // 4 in (48,)      # 48 float4 per row does not tile 256
    assert not kernels.gat_proj_supported(3, 64)


@pytest.mark.parametrize("H,D", [(2, 4), (8, 8), (8, 64)])
def test_gatconv_fused_layer_equals_compiled_path(cuda, H, D):
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn import functional as SF
    from stgraph_amd.nn.pytorch.static.gat_conv import GATConv
    n, e, fin = 3000, 40000, 24
    src, dst = random_graph(H * D, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    torch.manual_seed(1)
    conv = GATConv(fin, D, H).to(cuda)
    x0 = torch.randn(n, fin, device=cuda)
    R = torch.randn(n, H, D, device=cuda)
    res = []
    for fused in (True, False):
        usable = SF.gat_layer_usable
        if not fused:
            SF.gat_layer_usable = lambda *a: False
        try:
            conv.zero_grad()
            x = x0.clone().requires_grad_(True)
            out = conv(g, x)
            (out * R).sum().backward()
        finally:
            SF.gat_layer_usable = usable
        res.append((out.detach().clone(), x.grad.clone(), conv.attn_l.grad.clone(), conv.attn_r.grad.clone(),
                    conv.fc.weight.grad.clone()))
    assert torch.equal(res[0][0], res[1][0])                       # the output does not depend on el / er (D2)
    for a, b, name in zip(res[0][1:], res[1][1:], ("x", "attn_l", "attn_r", "fc.weight")):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max() + 1), msg=lambda m, n=name: f"{n}: {m}")


@pytest.mark.parametrize("fin,H", [(64, 8), (32, 8), (64, 2), (32, 4)])
def test_gat_fc_kernel_against_gemm_and_projection(cuda, fin, H):
    """stg_gat_fc_fwd (fc GEMM on MFMA with el / er from its accumulators) against x @ W.T in fp64 and the
    projections of that; ragged last tile."""
    from stgraph_amd import kernels
    D, n = 64, 16 * 777 + 5
    assert kernels.gat_fc_supported(fin, H, D) and not kernels.gat_fc_supported(fin, H, 32)
    assert not kernels.gat_fc_supported(128, 8, D)
    gen = torch.Generator(device=cuda).manual_seed(fin + H)
    x = torch.randn(n, fin, device=cuda, generator=gen)
    W = torch.randn(H * D, fin, device=cuda, generator=gen) / fin ** 0.5
    al = torch.randn(H, D, device=cuda, generator=gen)
    ar = torch.randn(H, D, device=cuda, generator=gen)
    feat, el, er = kernels.gat_fc_fwd(x, W, al, ar, H, D)
    want = (x.double() @ W.double().t()).view(n, H, D)
    torch.testing.assert_close(feat.double(), want, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(el.double(), (want * al.double()).sum(-1, keepdim=True), rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(er.double(), (want * ar.double()).sum(-1, keepdim=True), rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("fin,H", [(64, 8), (32, 2)])
def test_gatconv_with_fused_input_side_equals_the_unfused_layer(cuda, fin, H):
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn import functional as SF
    from stgraph_amd.nn.pytorch.static.gat_conv import GATConv
    D, n, e = 64, 3001, 40000
    src, dst = random_graph(H * D + fin, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    torch.manual_seed(3)
    conv = GATConv(fin, D, H).to(cuda)
    x0 = torch.randn(n, fin, device=cuda)
    R = torch.randn(n, H, D, device=cuda)
    res = []
    for fused in (True, False):
        SF.set_gat_fc(fused)
        try:
            assert SF.gat_fc_layer_usable(g, x0, conv.fc, H, D) == fused
            conv.zero_grad()
            x = x0.clone().requires_grad_(True)
            out = conv(g, x)
            out.backward(R)
        finally:
            SF.set_gat_fc(True)
        res.append((out.detach().clone(), x.grad.clone(), conv.attn_l.grad.clone(), conv.attn_r.grad.clone(),
                    conv.fc.weight.grad.clone()))
    for a, b, name in zip(res[0], res[1], ("out", "x", "attn_l", "attn_r", "fc.weight")):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max() + 1), msg=lambda m, n=name: f"{n}: {m}")


@pytest.mark.parametrize("fin,H", [(64, 8), (32, 4)])
@pytest.mark.parametrize("act", ["elu", "none", "elu_module", "tanh"])
def test_uniform_attention_form_equals_the_full_width_layer(cuda, fin, H, act):
    """The layer with K1 at the input width + product (+ fused elu, and its backward inside the per-vertex pass) against
    the same layer with K1 at full width and torch's elu around it: forward and every gradient."""
    import torch.nn.functional as F
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn import functional as SF
    from stgraph_amd.nn.pytorch.static.gat_conv import GATConv
    D, n, e = 64, 5003, 60000
    src, dst = random_graph(7 * H + fin, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    torch.manual_seed(5)
    activation = {"elu": F.elu, "none": None, "elu_module": torch.nn.ELU(), "tanh": torch.tanh}[act]   # smooth ones: a kink would flip on rounding
    assert SF.is_elu(activation) == act.startswith("elu") and not SF.is_elu(torch.nn.ELU(alpha=0.5))
    conv = GATConv(fin, D, H, activation=activation).to(cuda)
    x0 = torch.randn(n, fin, device=cuda)
    R = torch.randn(n, H, D, device=cuda)
    res = []
    for uniform in (True, False):
        kernels.set_gat_uniform_form(uniform)
        try:
            assert kernels.gat_uniform_usable(x0, H, D) == uniform
            rec = []
            kernels.enable_launch_timing(rec)
            conv.zero_grad()
            x = x0.clone().requires_grad_(True)
            out = conv(g, x)
            out.backward(R)
        finally:
            kernels.enable_launch_timing(None)
            kernels.set_gat_uniform_form(True)
        names = {r[0] for r in rec}
        assert ("gat_k1_uniform" in names) == uniform and ("gat_k1" in names) == (not uniform), names
        res.append((out.detach().clone(), x.grad.clone(), conv.attn_l.grad.clone(), conv.attn_r.grad.clone(),
                    conv.fc.weight.grad.clone()))
    for a, b, name in zip(res[0], res[1], ("out", "x", "attn_l", "attn_r", "fc.weight")):
        # attn_r's gradient is slope * (g . out - P S) summed over the vertices: exactly 0 in real arithmetic, rounding noise
        # of n terms of size |g . out| in either form -- compared on that scale
        atol = 5e-5 if name == "attn_r" else 1e-5 * float(b.abs().max() + 1)
        torch.testing.assert_close(a, b, rtol=1e-4, atol=atol, msg=lambda m, n=name: f"{n}: {m}")


@pytest.mark.parametrize("elu", [True, False])
def test_uniform_form_with_a_non_finite_score_is_the_emitted_unit_bit_for_bit(cuda, elu):
    """One inf in el: the device flag is set, the narrow pass returns at once and the full-width K1 overwrites out (and
    its elu) -- exactly gat_fwd's result; with finite scores the narrow pass's mean of x is K1's own sum at width fin."""
    import torch.nn.functional as F
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    fin, H, D, n, e = 64, 8, 64, 4099, 50000
    src, dst = random_graph(11, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    csr = g.csr("fwd")
    gen = torch.Generator(device=cuda).manual_seed(9)
    x = torch.randn(n, fin, device=cuda, generator=gen)
    W = torch.randn(H * D, fin, device=cuda, generator=gen) / 8
    al, ar = torch.randn(H, D, device=cuda, generator=gen), torch.randn(H, D, device=cuda, generator=gen)
    feat, el, er = kernels.gat_fc_fwd(x, W, al, ar, H, D)
    for bad in (True, False):
        el2 = el.clone()
        if bad:
            el2[int(src[0]), 3, 0] = float("inf")
        want, A0, S0 = kernels.gat_fwd(el2, er, feat, csr, 0.2, False, ones_shortcut=True)
        out, act, A, S = kernels.gat_fwd_uniform(x, W, el2, er, feat, csr, 0.2, False, elu)
        assert int(A._stg_ones.item()) == int(bad)
        torch.testing.assert_close(S, S0, rtol=0, atol=0, equal_nan=True)
        if bad:
            assert not torch.isfinite(want).all()
            torch.testing.assert_close(out, want, rtol=0, atol=0, equal_nan=True)
            torch.testing.assert_close(A, A0, rtol=0, atol=0, equal_nan=True)
        else:
            torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-5)
            deg = S[:, 0, 0]
            xm = torch.zeros(n, fin, device=cuda, dtype=torch.float64).index_add_(
                0, torch.as_tensor(dst, device=cuda).long(), x.double()[torch.as_tensor(src, device=cuda).long()])
            xm = xm / deg.double().clamp(min=1).unsqueeze(1)
            torch.testing.assert_close(out.double().view(n, -1), xm @ W.double().t(), rtol=1e-5, atol=1e-5)
        if elu:
            torch.testing.assert_close(act, F.elu(out), rtol=1e-6, atol=1e-7, equal_nan=True)
        else:
            assert act is None


def test_factored_backward_with_fused_elu_equals_elu_backward_then_the_unit(cuda):
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    for H, D in ((8, 64), (4, 16)):
        n, e = 3001, 30000
        src, dst = random_graph(13 + H, n, e)
        g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
        gen = torch.Generator(device=cuda).manual_seed(H)
        feat = torch.randn(n, H, D, device=cuda, generator=gen)
        el, er = torch.randn(n, H, 1, device=cuda, generator=gen), torch.randn(n, H, 1, device=cuda, generator=gen)
        out, A, S = kernels.gat_fwd(el, er, feat, g.csr("fwd"), 0.2, False, ones_shortcut=True)
        gy = torch.randn(n, H, D, device=cuda, generator=gen)
        got = kernels.gat_bwd(A, S, out, gy, el, er, feat, g.csr("fwd"), g.csr("bwd"), 0.2, False, elu=True)
        gpre = torch.ops.aten.elu_backward(gy, 1.0, 1.0, 1.0, False, out)
        want = kernels.gat_bwd(A, S, out, gpre, el, er, feat, g.csr("fwd"), g.csr("bwd"), 0.2, False)
        for a, b, name in zip(got, want, ("grad_feat", "grad_el", "grad_er")):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5 * float(b.abs().max() + 1), msg=lambda m, n=name: f"{n}: {m}")


@pytest.mark.parametrize("fin,H,act", [(64, 8, "elu"), (32, 4, "none")])
def test_projection_backward_at_width_H_equals_the_full_width_one(cuda, fin, H, act):
    """_GatFcLayer.backward with the attention projections' terms formed from [gel | ger]^T x (no dfeat, no stg_gat_proj_bwd)
    against the same layer with stg_gat_proj_bwd: every gradient."""
    import torch.nn.functional as F
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn import functional as SF
    from stgraph_amd.nn.pytorch.static.gat_conv import GATConv
    D, n, e = 64, 70_003, 700_000                       # >= 64 K rows: the native weight-gradient path
    src, dst = random_graph(17 + H, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    torch.manual_seed(6)
    conv = GATConv(fin, D, H, activation=F.elu if act == "elu" else None).to(cuda)
    x0 = torch.randn(n, fin, device=cuda)
    R = torch.randn(n, H, D, device=cuda)
    res = []
    for fold in (True, False):
        SF.set_gat_proj_fold(fold)
        try:
            rec = []
            kernels.enable_launch_timing(rec)
            conv.zero_grad()
            x = x0.clone().requires_grad_(True)
            conv(g, x).backward(R)
        finally:
            kernels.enable_launch_timing(None)
            SF.set_gat_proj_fold(True)
        assert ("gat_proj_bwd" in {r[0] for r in rec}) == (not fold)
        res.append((x.grad.clone(), conv.attn_l.grad.clone(), conv.attn_r.grad.clone(), conv.fc.weight.grad.clone()))
    for a, b, name in zip(res[0], res[1], ("x", "attn_l", "attn_r", "fc.weight")):
        atol = 5e-5 if name == "attn_r" else 1e-5 * float(b.abs().max() + 1)      # attn_r: rounding noise around 0 (see above)
        torch.testing.assert_close(a, b, rtol=1e-4, atol=atol, msg=lambda m, nm=name: f"{nm}: {m}")


@pytest.mark.parametrize("act", ["elu", "none"])
@pytest.mark.parametrize("bad", [False, True])
def test_uniform_backward_unit_equals_the_factored_one(cuda, act, bad):
    """The backward of the uniform-attention layer at H = 8, D = 64, fin = 64 without a gather of width H * D per edge
    (stg_gat_bwd_uniform_edges: x gathered against per-vertex products of g with W, grad_feat only through grad_feat W and
    g^T xm) against the factored unit: every gradient.  `bad`: one non-finite input row sets the device flag -- every step
    then takes the general unit's results (grad_feat, T) through the gated launches, NaNs in the same places."""
    import torch.nn.functional as F
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn.pytorch.static.gat_conv import GATConv
    fin, H, D, n, e = 64, 8, 64, 70_003, 700_000
    src, dst = random_graph(23, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    torch.manual_seed(8)
    conv = GATConv(fin, D, H, activation=F.elu if act == "elu" else None).to(cuda)
    x0 = torch.randn(n, fin, device=cuda)
    if bad:
        x0[int(src[5]), 3] = float("inf")
    R = torch.randn(n, H, D, device=cuda)
    res = []
    for on in (True, False):
        kernels.set_gat_uniform_backward(on)
        try:
            rec = []
            kernels.enable_launch_timing(rec)
            conv.zero_grad()
            x = x0.clone().requires_grad_(True)
            out = conv(g, x)
            out.backward(R)
        finally:
            kernels.enable_launch_timing(None)
            kernels.set_gat_uniform_backward(True)
        names = {r[0] for r in rec}
        assert ("gat_bwd_uniform" in names) == on and ("gat_bwd" in names) == (not on), names
        res.append((out.detach().clone(), x.grad.clone(), conv.attn_l.grad.clone(), conv.attn_r.grad.clone(),
                    conv.fc.weight.grad.clone()))
    assert torch.equal(res[0][0].isnan(), res[1][0].isnan()) and bool(res[0][0].isnan().any()) == bad
    for a, b, name in zip(res[0], res[1], ("out", "x", "attn_l", "attn_r", "fc.weight")):
        assert torch.equal(a.isnan(), b.isnan()), name
        fin_b = b[~b.isnan()]
        atol = 5e-5 if name == "attn_r" else 2e-5 * float(fin_b.abs().max() + 1 if fin_b.numel() else 1)
        torch.testing.assert_close(a, b, rtol=2e-4, atol=atol, equal_nan=True, msg=lambda m, n=name: f"{n}: {m}")


def test_gated_contraction_runs_on_the_device_word(cuda):
    """stg_gemm_tn_gated_f32: of two products gated on the same word, one of each kind, exactly one writes C."""
    from stgraph_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(1)
    a1, b1 = torch.randn(70_000, 512, device=cuda, generator=gen), torch.randn(70_000, 64, device=cuda, generator=gen)
    a2, b2 = torch.randn(70_000, 512, device=cuda, generator=gen), torch.randn(70_000, 64, device=cuda, generator=gen)
    for word in (0, 1):
        gate = torch.full((1,), word, dtype=torch.int32, device=cuda)
        c = torch.full((512, 64), 7.0, device=cuda)
        kernels.gemm_tn_gated(a1, b1, c, gate, True)
        kernels.gemm_tn_gated(a2, b2, c, gate, False)
        want = kernels.gemm_tn(a2, b2) if word else kernels.gemm_tn(a1, b1)
        assert torch.equal(c, want)


def test_uniform_launches_at_the_full_cfg3_shape(cuda):
    """BASELINE configs[2] at full size (|V| = 256K, |E| = 8M, GATConv(64, 64, 8 heads, elu)) through the launches bench.py TIMES
    (gat_k1_uniform, gat_fc_out, gat_bwd_prepass, gat_bwd_gw, gat_bwd_uniform): the launch record proves they ran; the results are
    checked through size-independent identities of the degenerate softmax (every A = 1: SURVEY.md D2) --
      out = elu(mean over in-neighbours of (x W^T)): kernels.gcn_agg with 1 / in-degree, over the whole tensor --
    and against the general unit (set_gat_uniform_form / _backward(False): emitted K0 / K1 / K2 at width H x D) on 512 sampled
    rows of every gradient."""
    import torch.nn.functional as F
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn.pytorch.static.gat_conv import GATConv
    n, e, fin, H, D = 256_000, 8_000_000, 64, 8, 64
    gen = torch.Generator(device=cuda).manual_seed(2)
    key = torch.unique(torch.randint(0, n * n, (int(e * 1.02),), generator=gen, device=cuda, dtype=torch.int64))
    key = key[torch.randperm(key.shape[0], generator=gen, device=cuda)][:e]
    g = StaticGraph(((key // n).to(torch.int32), (key % n).to(torch.int32)), None, n, device=cuda, sort_inplace=False)
    torch.manual_seed(2)
    conv = GATConv(fin, D, H, activation=F.elu).to(cuda)
    x0 = torch.randn(n, fin, device=cuda, generator=gen)
    R = torch.randn(n, H, D, device=cuda, generator=gen)
    rows = torch.randint(0, n, (512,), generator=gen, device=cuda)
    res = {}
    for form in ("uniform", "general"):
        kernels.set_gat_uniform_form(form == "uniform")
        kernels.set_gat_uniform_backward(form == "uniform")
        try:
            rec = []
            kernels.enable_launch_timing(rec)
            conv.zero_grad()
            x = x0.clone().requires_grad_(True)
            out = conv(g, x)
            out.backward(R)
        finally:
            kernels.enable_launch_timing(None)
            kernels.set_gat_uniform_form(True)
            kernels.set_gat_uniform_backward(True)
        names = {r[0] for r in rec}
        if form == "uniform":
            assert {"gat_k1_uniform", "gat_fc_out", "gat_bwd_prepass_heads", "gat_bwd_uniform"} <= names, sorted(names)
            assert "gat_k1" not in names and "gat_bwd" not in names, sorted(names)
        else:
            assert {"gat_k1", "gat_bwd"} <= names and "gat_k1_uniform" not in names and "gat_bwd_uniform" not in names, sorted(names)
        res[form] = (out.detach(), x.grad.clone(), conv.fc.weight.grad.clone(), conv.attn_l.grad.clone(), conv.attn_r.grad.clone())
        del x, out
    # identity: elu(mean aggregation of feat) over the WHOLE output
    with torch.no_grad():
        feat = x0 @ conv.fc.weight.t()
        fwd = g.csr("fwd")
        deg = (fwd.row_offset[1:] - fwd.row_offset[:-1]).float()
        inv = torch.where(deg > 0, 1.0 / deg, torch.zeros_like(deg)).unsqueeze(1)
        mean = kernels.gcn_agg(feat, inv, torch.ones(n, 1, device=cuda), fwd)
        want = F.elu(mean).view(n, H, D)
    torch.testing.assert_close(res["uniform"][0], want, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(res["uniform"][0][rows], res["general"][0][rows], rtol=1e-4, atol=2e-5)
    # gradients: the general unit on the sampled rows (x) and in full (parameters), 1e-4 of the tensor's largest entry
    gu, gg = res["uniform"], res["general"]
    assert float((gu[1][rows] - gg[1][rows]).abs().max()) <= 1e-4 * float(gg[1].abs().max())
    assert float((gu[2] - gg[2]).abs().max()) <= 1e-4 * float(gg[2].abs().max())
    assert float((gu[3] - gg[3]).abs().max()) <= 1e-4 * float(gg[3].abs().max())
    # attn_r's gradient is zero in exact arithmetic (rounding noise in both forms): held to attn_l's scale
    assert float((gu[4] - gg[4]).abs().max()) <= 1e-4 * float(gg[3].abs().max())
    for t in gu:
        assert bool(torch.isfinite(t).all())


@pytest.mark.parametrize("N", [1, 31, 4099, 70_001])
def test_heads_products_in_the_split_form(cuda, N):
    """csrc/gat_heads_x3.hip at 8 heads of 64 over 64 inputs: stg_gat_fc_fwd (feat, el, er), stg_gat_fc_out (out, elu(out)) against fp64
    -- error no worse than the fp32-instruction kernels' (knob rowgemm_x3 = 1) -- and against those kernels to fp32 rounding."""
    import torch.nn.functional as F
    from stgraph_amd import _C, kernels
    fin, H, D = 64, 8, 64
    gen = torch.Generator(device=cuda).manual_seed(N)
    x = torch.randn(N, fin, device=cuda, generator=gen)
    W = torch.randn(H * D, fin, device=cuda, generator=gen) / 8
    al, ar = torch.randn(H, D, device=cuda, generator=gen), torch.randn(H, D, device=cuda, generator=gen)
    res = {}
    for knob in (0, 1):
        _C.set_tuning("rowgemm_x3", knob)
        try:
            feat, el, er = kernels.gat_fc_fwd(x, W, al, ar, H, D)
            _, el2, er2 = kernels.gat_fc_fwd(x, W, al, ar, H, D, store_feat=False)
            out = torch.empty(N, H, D, device=cuda)
            act = torch.empty(N, H, D, device=cuda)
            _C.check(_C.lib.stg_gat_fc_out(kernels._ptr(x), kernels._ptr(W), kernels._ptr(out), kernels._ptr(act), N, fin, H, D,
                                           kernels._stream_ptr(cuda)))
            assert torch.equal(el, el2) and torch.equal(er, er2)
            res[knob] = (feat, el, er, out, act)
        finally:
            _C.set_tuning("rowgemm_x3", 0)
    ref = (x.double() @ W.double().t()).view(N, H, D)
    refs = (ref, (ref * al.double()).sum(-1, keepdim=True), (ref * ar.double()).sum(-1, keepdim=True), ref, F.elu(ref))
    for i, name in enumerate(("feat", "el", "er", "out", "elu(out)")):
        scale = float(refs[i].abs().max()) + 1e-30
        e_new = float((res[0][i].double() - refs[i]).abs().max()) / scale
        e_old = float((res[1][i].double() - refs[i]).abs().max()) / scale
        assert e_new <= max(2 * e_old, 2e-6), (name, e_new, e_old)
        torch.testing.assert_close(res[0][i], res[1][i], rtol=1e-5, atol=1e-5 * scale)


@pytest.mark.parametrize("elu", [True, False])
@pytest.mark.parametrize("N", [65536 + 17, 70_003])
def test_prepass_and_heads_products_in_one_pass(cuda, N, elu):
    """stg_gat_bwd_prepass_heads against stg_gat_bwd_prepass + stg_rowgemm_heads_f32 on the same inputs: g_pre and gW bit for bit (the
    same formula per element; the same split product), pack and grad_er to fp32 rounding (the per-head dot products add the same
    terms in another order); a vertex without an in-edge (S = 0) gets grad_er = 0 in both."""
    from stgraph_amd import _C, kernels
    fin, H, D = 64, 8, 64
    gen = torch.Generator(device=cuda).manual_seed(N + elu)
    S = torch.rand(N, H, 1, device=cuda, generator=gen) * 5 + 0.5
    S[7] = 0.0
    out = torch.randn(N, H, D, device=cuda, generator=gen)
    g = torch.randn(N, H, D, device=cuda, generator=gen)
    W = torch.randn(H * D, fin, device=cuda, generator=gen) / 8
    st = kernels._stream_ptr(cuda)
    P = kernels._ptr
    new = lambda *s: torch.full(s, float("nan"), device=cuda)  # noqa: E731
    gp0, pack0, ger0 = (new(N, H, D) if elu else None), new(N, 16), new(N, H, 1)
    _C.check(_C.lib.stg_gat_bwd_prepass(P(S), P(out), P(g), P(gp0), P(pack0), N, H, D, 0.2, P(ger0), st))
    gW0 = new(H, N, fin)
    _C.check(_C.lib.stg_rowgemm_heads_f32(P(gp0 if elu else g), P(W), P(gW0), N, D, fin, H, st))
    assert _C.lib.stg_gat_bwd_prepass_heads_supported(N, H, D, fin)
    gp1, pack1, ger1, gW1 = (new(N, H, D) if elu else None), new(N, 16), new(N, H, 1), new(H, N, fin)
    _C.check(_C.lib.stg_gat_bwd_prepass_heads(P(S), P(out), P(g), P(gp1), P(pack1), P(ger1), P(W), P(gW1), N, H, D, fin, 0.2, st))
    if elu:
        assert torch.equal(gp0, gp1)
    assert torch.equal(gW0, gW1)
    assert torch.equal(pack0[:, :8], pack1[:, :8])
    keep = torch.ones(N, dtype=torch.bool, device=cuda)
    keep[7] = False                                    # S = 0: the P term is inf / nan in both
    torch.testing.assert_close(pack1[keep, 8:], pack0[keep, 8:], rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(ger1[keep], ger0[keep], rtol=1e-5, atol=1e-4)
    assert not ger1[7].any() and not ger0[7].any()
    assert not _C.lib.stg_gat_bwd_prepass_heads_supported(N, 6, D, fin) and not _C.lib.stg_gat_bwd_prepass_heads_supported(N, H, 32, fin)


@pytest.mark.parametrize("H,D,fin", [(8, 64, 64), (1, 16, 512), (4, 32, 64), (2, 64, 33)])
def test_projection_fold_products_in_one_launch(cuda, H, D, fin):
    """stg_gat_attn_fold against the einsums / elementwise launches it replaces (fp64 reference): attention gradients, A_w, and the
    correction added into gw in place."""
    from stgraph_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(H * D + fin)
    W = torch.randn(H * D, fin, device=cuda, generator=gen)
    G = torch.randn(2 * H, fin, device=cuda, generator=gen)
    al, ar = torch.randn(H, D, device=cuda, generator=gen), torch.randn(H, D, device=cuda, generator=gen)
    gw0 = torch.randn(H * D, fin, device=cuda, generator=gen)
    if not kernels.gat_attn_fold_usable(W, H, D, fin):
        assert 4 * (D * fin + 2 * fin + 2 * D) > 64 * 1024
        return
    gw = gw0.clone()
    dal, dar, Aw = kernels.gat_attn_fold(W, G, al, ar, H, D, fin, want_aw=True, gw=gw)
    Wh, Gd, ald, ard = W.double().view(H, D, fin), G.double(), al.double(), ar.double()
    want = (torch.einsum("hdf,hf->hd", Wh, Gd[:H]), torch.einsum("hdf,hf->hd", Wh, Gd[H:]),
            torch.cat([torch.einsum("hdf,hd->hf", Wh, ald), torch.einsum("hdf,hd->hf", Wh, ard)], 0),
            gw0.double() + (ald.unsqueeze(2) * Gd[:H].unsqueeze(1) + ard.unsqueeze(2) * Gd[H:].unsqueeze(1)).reshape(H * D, fin))
    for got, ref, name in zip((dal, dar, Aw, gw), want, ("dattn_l", "dattn_r", "A_w", "gw")):
        torch.testing.assert_close(got.double(), ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()), msg=lambda m, n=name: f"{n}: {m}")
    dal2, dar2, none = kernels.gat_attn_fold(W, G, al, ar, H, D, fin, want_aw=False, gw=None)
    assert none is None and torch.equal(dal, dal2) and torch.equal(dar, dar2)
