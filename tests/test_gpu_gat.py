"""GPU parity of the fused GAT units (stg_gat_fwd_k0/k1, stg_gat_bwd, stg_gat_bwd_er).

Forward (A, S, out) and grad_feat follow the reference's summation order => bit-exact.
grad_el / grad_er are atomicAdd sums in the reference (order undefined); here they are
deterministic in-wave / per-vertex sums => compared within the north star's 1e-4.
"""
import numpy as np
import pytest
import torch

from oracle import stg_oracle as orc
from tests.util import GAT_SHAPES, golden, random_graph

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("compat", [True, False])
@pytest.mark.parametrize("H,D", GAT_SHAPES)
def test_golden_kernel_level(cuda, H, D, compat):
    import stgraph_amd
    from stgraph_amd import kernels
    d = golden("gat.npz")
    n = int(d["num_nodes"])
    g = kernels.build_graph_csr(d["src"], d["dst"], n, cuda)
    tag = f"H{H}_D{D}"
    el, er, feat, R = (_t(d[tag + k], cuda) for k in ("_k_el", "_k_er", "_k_feat", "_R"))
    ref_full = kernels.ref_active_columns(H) == H and kernels.ref_active_columns(H * D) == H * D
    stgraph_amd.set_reference_compat(compat)
    try:
        out, A, S = kernels.gat_fwd(el, er, feat, g.fwd, 0.2)
        gf, gel, ger = kernels.gat_bwd(A, S, out, R, el, er, feat, g.fwd, g.bwd, 0.2)
    finally:
        stgraph_amd.set_reference_compat(False)
    got = {k: v.cpu().numpy() for k, v in dict(A=A, S=S, out=out, gf=gf, gel=gel, ger=ger).items()}
    if compat or ref_full:
        assert np.array_equal(got["A"], d[tag + "_k_A"])
        assert np.array_equal(got["S"], d[tag + "_k_S"])
        assert np.array_equal(got["out"], d[tag + "_out"])
        assert np.array_equal(got["gf"], d[tag + "_k_grad_feat"])
        np.testing.assert_allclose(got["gel"], d[tag + "_k_grad_el"], rtol=TOL, atol=TOL)
        np.testing.assert_allclose(got["ger"], d[tag + "_k_grad_er"], rtol=TOL, atol=TOL)
    else:
        # D1 off: every column is computed; compare with the oracle run on all columns
        og = orc.build_graph(d["src"], d["dst"], n)
        A0, S0 = orc.gat_k0(d[tag + "_k_el"], d[tag + "_k_er"], og.fwd, og.num_edges)
        o0 = orc.gat_k1(A0, S0, d[tag + "_k_feat"], og.fwd)
        gf0, gel0, ger0 = orc.gat_bwd(A0, S0, o0, d[tag + "_R"], d[tag + "_k_el"], d[tag + "_k_er"],
                                      d[tag + "_k_feat"], og.bwd)
        assert np.array_equal(got["out"], o0) and np.array_equal(got["gf"], gf0)
        np.testing.assert_allclose(got["gel"], gel0, rtol=TOL, atol=TOL)
        np.testing.assert_allclose(got["ger"], ger0, rtol=TOL, atol=TOL)


@pytest.mark.parametrize("H,D,nid", [(1, 1, False), (2, 4, True), (4, 16, False), (8, 64, False), (8, 64, True),
                                     (3, 6, False), (1, 256, False), (5, 12, True), (16, 32, False)])
def test_oracle_random_graph(cuda, H, D, nid):
    from stgraph_amd import kernels
    n, e = 2000, 30000
    src, dst = random_graph(H * 100 + D, n, e)
    g = kernels.build_graph_csr(src, dst, n, cuda)
    og = orc.build_graph(src, dst, n)
    rng = np.random.default_rng(D)
    el = rng.standard_normal((n, H, 1)).astype(np.float32)
    er = rng.standard_normal((n, H, 1)).astype(np.float32)
    feat = rng.standard_normal((n, H, D)).astype(np.float32)
    R = rng.standard_normal((n, H, D)).astype(np.float32)
    out, A, S = kernels.gat_fwd(_t(el, cuda), _t(er, cuda), _t(feat, cuda), g.fwd, 0.2, nid)
    gf, gel, ger = kernels.gat_bwd(A, S, out, _t(R, cuda), _t(el, cuda), _t(er, cuda), _t(feat, cuda),
                                   g.fwd, g.bwd, 0.2, nid)
    A0, S0 = orc.gat_k0(el, er, og.fwd, og.num_edges, use_node_ids=nid)
    o0 = orc.gat_k1(A0, S0, feat, og.fwd, use_node_ids=nid)
    gf0, gel0, ger0 = orc.gat_bwd(A0, S0, o0, R, el, er, feat, og.bwd, use_node_ids=nid)
    assert np.array_equal(A.cpu().numpy(), A0) and np.array_equal(S.cpu().numpy(), S0)
    assert np.array_equal(out.cpu().numpy(), o0)
    assert np.array_equal(gf.cpu().numpy(), gf0)
    scale = max(1.0, float(np.abs(gel0).max()), float(np.abs(ger0).max()))
    np.testing.assert_allclose(gel.cpu().numpy(), gel0, rtol=TOL, atol=TOL * scale)
    np.testing.assert_allclose(ger.cpu().numpy(), ger0, rtol=TOL, atol=TOL * scale)
    # literal form of the emitted kernel (stg_gat_bwd) agrees with the default factored form
    kernels.set_gat_factored_backward(False)
    try:
        gfl, gell, gerl = kernels.gat_bwd(A, S, out, _t(R, cuda), _t(el, cuda), _t(er, cuda), _t(feat, cuda),
                                          g.fwd, g.bwd, 0.2, nid)
    finally:
        kernels.set_gat_factored_backward(True)
    assert np.array_equal(gfl.cpu().numpy(), gf0)
    np.testing.assert_allclose(gell.cpu().numpy(), gel0, rtol=TOL, atol=TOL * scale)
    np.testing.assert_allclose(gerl.cpu().numpy(), ger0, rtol=TOL, atol=TOL * scale)
    # grad_er summed from the per-edge terms T (stg_gat_bwd_er) against the default, regrouped per-vertex form: both are
    # rounding noise around zero (the softmax gradient sums to zero over a target's in-edges); grad_feat / grad_el same bits
    kernels.set_gat_regrouped_er(False)
    try:
        gft, gelt, gert = kernels.gat_bwd(A, S, out, _t(R, cuda), _t(el, cuda), _t(er, cuda), _t(feat, cuda),
                                          g.fwd, g.bwd, 0.2, nid)
    finally:
        kernels.set_gat_regrouped_er(True)
    assert torch.equal(gft, gf) and torch.equal(gelt, gel)
    np.testing.assert_allclose(gert.cpu().numpy(), ger0, rtol=TOL, atol=TOL * scale)
    assert float(ger.abs().max()) <= 1e-5 * scale and float(gert.abs().max()) <= 1e-5 * scale
    # determinism: no atomics anywhere
    gf2, gel2, ger2 = kernels.gat_bwd(A, S, out, _t(R, cuda), _t(el, cuda), _t(er, cuda), _t(feat, cuda),
                                      g.fwd, g.bwd, 0.2, nid)
    assert torch.equal(gel, gel2) and torch.equal(ger, ger2) and torch.equal(gf, gf2)


def test_nonfinite_scores_propagate_like_the_reference(cuda):
    """s - s is NaN for s = +-inf/NaN: the literal GIR keeps that."""
    from stgraph_amd import kernels
    src, dst = np.array([0, 1, 2], np.int32), np.array([1, 2, 0], np.int32)
    g = kernels.build_graph_csr(src, dst, 3, cuda)
    el = torch.tensor([[[float("inf")]], [[0.0]], [[1.0]]], device=cuda)
    er = torch.zeros(3, 1, 1, device=cuda)
    feat = torch.ones(3, 1, 4, device=cuda)
    out, A, S = kernels.gat_fwd(el, er, feat, g.fwd, 0.2)
    og = orc.build_graph(src, dst, 3)
    A0, S0 = orc.gat_k0(el.cpu().numpy(), er.cpu().numpy(), og.fwd, 3)
    assert np.array_equal(np.isnan(A.cpu().numpy()), np.isnan(A0)) and np.isnan(A0).sum() == 1


def test_unsupported_head_width_fails_loudly(cuda):
    from stgraph_amd import _C, kernels
    src, dst = random_graph(3, 50, 200)
    g = kernels.build_graph_csr(src, dst, 50, cuda)
    H, D = 1, 1024
    z = lambda *s: torch.zeros(*s, device=cuda)  # noqa: E731
    out, A, S = kernels.gat_fwd(z(50, H, 1), z(50, H, 1), z(50, H, D), g.fwd, 0.2)
    with pytest.raises(_C.StgError):
        kernels.gat_bwd(A, S, out, z(50, H, D), z(50, H, 1), z(50, H, 1), z(50, H, D), g.fwd, g.bwd, 0.2)


def test_full_size_properties(cuda):
    """BASELINE config 3 shape (|V|=256K, |E|=8M, H=8, D=64): uniform attention => mean aggregation."""
    from stgraph_amd import kernels
    n, e, H, D = 256_000, 8_000_000, 8, 64
    gen = torch.Generator(device=cuda).manual_seed(2)
    src = torch.randint(0, n, (e,), generator=gen, device=cuda, dtype=torch.int32)
    dst = torch.randint(0, n, (e,), generator=gen, device=cuda, dtype=torch.int32)
    g = kernels.build_graph_csr(src, dst, n, cuda)
    el = torch.randn(n, H, 1, device=cuda, generator=gen)
    er = torch.randn(n, H, 1, device=cuda, generator=gen)
    feat = torch.randn(n, H, D, device=cuda, generator=gen)
    out, A, S = kernels.gat_fwd(el, er, feat, g.fwd, 0.2)
    assert bool((A == 1).all())                                        # exp(leaky_relu(0)) == 1 exactly
    assert torch.equal(S[:, 0, 0], g.in_degrees.float())               # S == in-degree
    ones = torch.ones(n, 1, device=cuda)
    inv = torch.where(g.in_degrees > 0, 1.0 / g.in_degrees.float(), torch.zeros(n, device=cuda)).unsqueeze(1)
    mean = kernels.gcn_agg(feat.view(n, H * D), inv, ones, g.fwd).view(n, H, D)
    torch.testing.assert_close(out, mean, rtol=1e-4, atol=1e-5)
    R = torch.randn(n, H, D, device=cuda, generator=gen)
    gf, gel, ger = kernels.gat_bwd(A, S, out, R, el, er, feat, g.fwd, g.bwd, 0.2)
    # grad_feat is the adjoint of the mean aggregation
    gf_ref = kernels.gcn_agg(R.view(n, H * D), ones, inv, g.bwd).view(n, H, D)
    torch.testing.assert_close(gf, gf_ref, rtol=1e-4, atol=1e-5)
    # sum_u grad_el[u] == sum_v grad_er[v]: both are the same edge terms, summed by source / by target
    a, b = gel.double().sum(0).view(-1), ger.double().sum(0).view(-1)
    assert torch.allclose(a, b, rtol=1e-6, atol=1e-3)
    assert bool(torch.isfinite(gel).all()) and bool(torch.isfinite(ger).all())


@pytest.mark.parametrize("H,D", [(8, 8), (8, 64)])
def test_golden_cora_shaped(cuda, H, D):
    """Reference-generated vectors on the benchmark's Cora-shaped graph (hub of in-degree 159): A, S in full, out and
    grad_feat on the sampled rows + fp64 column sums (bit-exact), grad_el / grad_er within 1e-4 (reference: atomics)."""
    from stgraph_amd import kernels
    d = golden("gat_cora.npz")
    n = int(d["num_nodes"])
    g = kernels.build_graph_csr(d["src"], d["dst"], n, cuda)
    tag, rows = f"H{H}_D{D}", d["rows"]
    feat = (d[tag + "_x"].astype(np.float64) @ d[tag + "_fc_weight"].astype(np.float64).T).astype(np.float32).reshape(n, H, D)
    assert np.array_equal(feat[rows], d[tag + "_k_feat_rows"])
    rng = np.random.default_rng(int(d[tag + "_seed"]))
    rng.integers(-8, 9, (n, 6)), rng.integers(-16, 17, (H * D, 6))
    R = rng.standard_normal((n, H, D), dtype=np.float32)
    el, er = _t(d[tag + "_k_el"], cuda), _t(d[tag + "_k_er"], cuda)
    out, A, S = kernels.gat_fwd(el, er, _t(feat, cuda), g.fwd, 0.2)
    for factored in (True, False):
        kernels.set_gat_factored_backward(factored)
        try:
            gf, gel, ger = kernels.gat_bwd(A, S, out, _t(R, cuda), el, er, _t(feat, cuda), g.fwd, g.bwd, 0.2)
        finally:
            kernels.set_gat_factored_backward(True)
        gf = gf.cpu().numpy()
        assert np.array_equal(gf[rows], d[tag + "_k_grad_feat_rows"])          # grad_feat: the reference's sums, both forms
        assert np.array_equal(gf.astype(np.float64).sum(0), d[tag + "_k_grad_feat_colsum"])
        np.testing.assert_allclose(gel.cpu().numpy(), d[tag + "_k_grad_el"], rtol=TOL, atol=TOL)
        np.testing.assert_allclose(ger.cpu().numpy(), d[tag + "_k_grad_er"], rtol=TOL, atol=TOL)
    assert np.array_equal(A.cpu().numpy(), d[tag + "_k_A"]) and np.array_equal(S.cpu().numpy(), d[tag + "_k_S"])
    o = out.cpu().numpy()
    assert np.array_equal(o[rows], d[tag + "_out_rows"])
    assert np.array_equal(o.astype(np.float64).sum(0), d[tag + "_out_colsum"])


@pytest.mark.parametrize("H,D", [(8, 64), (2, 16), (8, 8)])
@pytest.mark.parametrize("poison", [False, True])
def test_all_ones_shortcut_is_bit_identical(cuda, H, D, poison, monkeypatch):
    """Finite scores: A == 1.0f and S == in-degree, so the layers skip writing / reading A (device flag from
    stg_gat_score_flag); one inf score: the flag is set and every unit takes the general path.  Same bits either way in the
    forward (and in the backward when the flag is set), through the autograd node the layers use."""
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn import functional as SF
    if not kernels.gat_proj_supported(H, D):
        pytest.skip("the one-node GAT layer does not cover this head shape")
    n, e = 3000, 40000
    src, dst = random_graph(H * D + 1, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    gen = torch.Generator(device=cuda).manual_seed(H + D)
    feat0 = torch.randn(n, H, D, device=cuda, generator=gen)
    al = torch.randn(H, D, device=cuda, generator=gen)
    ar = torch.randn(H, D, device=cuda, generator=gen)
    if poison:
        feat0[17, 0, 0] = float("inf")                     # el[17, 0] = er[17, 0] = +-inf
    R = torch.randn(n, H, D, device=cuda, generator=gen)
    seen = []
    real_bwd = kernels.gat_bwd
    monkeypatch.setattr(kernels, "gat_bwd", lambda A, *a, **k: (seen.append(getattr(A, "_stg_ones", None)), real_bwd(A, *a, **k))[1])
    res = []
    for on in (True, False):
        kernels.set_gat_ones_shortcut(on)
        try:
            feat = feat0.clone().requires_grad_(True)
            a1, a2 = al.clone().requires_grad_(True), ar.clone().requires_grad_(True)
            out = SF._GatLayer.apply(feat, a1, a2, g.csr("fwd"), g.csr("bwd"), False, 0.2)
            out.backward(R)
        finally:
            kernels.set_gat_ones_shortcut(True)
        res.append((out.detach(), feat.grad, a1.grad, a2.grad))
    assert seen[0] is not None and int(seen[0].item()) == int(poison) and seen[1] is None
    for k, (a, b) in enumerate(zip(*res)):
        if poison or k == 0:
            assert torch.equal(torch.nan_to_num(a, nan=7.0, posinf=8.0, neginf=9.0), torch.nan_to_num(b, nan=7.0, posinf=8.0, neginf=9.0))
        else:
            # with every A = 1.0f the backward unit takes grad_el from ONE dot product per row, f . grad_feat[u], instead of one
            # per edge (gat_bwd_fact_kernel's `lite` path): the same sum in another order
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max() + 1), k
    if not poison:
        deg = g.csr("fwd").row_offset[1:] - g.csr("fwd").row_offset[:-1]
        el, er = kernels.gat_proj_fwd(feat0, al, ar)
        _, A, S = kernels.gat_fwd(el, er, feat0, g.csr("fwd"), 0.2)
        assert bool((A == 1).all()) and torch.equal(S.view(n, H), deg.float().view(n, 1).expand(n, H))
