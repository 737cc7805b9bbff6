"""The forward row-local chain of a TGCN step as one launch (csrc/tgcn_cell_fused.hip) against the unfused
stages (three elementwise kernels around three rocBLAS GEMMs) and an fp64 torch restatement of reference
nn/pytorch/temporal/tgcn.py:21-55."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _operands(cuda, N, C, seed):
    g = torch.Generator(device=cuda).manual_seed(seed)
    r = lambda *s: torch.randn(*s, device=cuda, generator=g)  # noqa: E731
    a3, b3, H = r(N, 3 * C), r(3 * C), r(N, C)
    Ws = [r(C, 2 * C) * 0.2 for _ in range(3)]
    bs = [r(C) for _ in range(3)]
    if N > 3:                                  # make the clamp bite
        a3[0, :5] = 3e6
        a3[2, C + 1] = -5e6
    return a3, b3, H, Ws, bs


@pytest.mark.parametrize("N", [1, 31, 32, 33, 4097, 50_000])
@pytest.mark.parametrize("C", [32, 64])
def test_fused_forward_matches_unfused_and_fp64(cuda, N, C):
    from stgraph_amd.nn.pytorch.temporal import cell
    a3, b3, H, (Wz, Wr, Wh), (bz, br, bh) = _operands(cuda, N, C, N + C)
    res = []
    for fused in (True, False):
        cell.set_fused_forward(fused)
        try:
            Hn, extra = cell._cell_forward(a3, b3, H, Wz, bz, Wr, br, Wh, bh)
        finally:
            cell.set_fused_forward(True)
        res.append((Hn, *extra))
    names = ("Hn", "CZ", "CR", "CH", "Z", "R", "Ht")
    for name, a, b in zip(names, *res):
        if name in ("CZ", "CR"):
            assert torch.equal(a, b), name                       # no GEMM involved: identical bits
        else:
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5, msg=lambda m, n=name: f"{n}: {m}")
    # fp64 restatement of tgcn.py:21-55
    d = lambda t: t.double()  # noqa: E731
    h = torch.clamp(d(a3) + d(b3), -1e6, 1e6)
    hz, hr, hh = h[:, :C], h[:, C:2 * C], h[:, 2 * C:]
    Z = torch.sigmoid(torch.cat([hz, d(H)], 1) @ d(Wz).t() + d(bz))
    R = torch.sigmoid(torch.cat([hr, d(H)], 1) @ d(Wr).t() + d(br))
    Ht = torch.tanh(torch.cat([hh, d(H) * R], 1) @ d(Wh).t() + d(bh))
    Hn = Z * d(H) + (1 - Z) * Ht
    torch.testing.assert_close(res[0][0].double(), Hn, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(res[0][5].double(), R, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("N", [1, 33, 4097, 50_000])
@pytest.mark.parametrize("C", [32, 64])
def test_fused_backward_matches_unfused(cuda, N, C):
    from stgraph_amd.nn.pytorch.temporal import cell
    a3, b3, H, (Wz, Wr, Wh), (bz, br, bh) = _operands(cuda, N, C, 7 * N + C)
    cell.set_fused_forward(False)
    try:
        Hn, (CZ, CR, CH, Z, R, Ht) = cell._cell_forward(a3, b3, H, Wz, bz, Wr, br, Wh, bh)
    finally:
        cell.set_fused_forward(True)
    dHn = torch.randn(N, C, device=cuda)
    res = []
    for fused in (True, False):
        cell.set_fused_backward(fused)
        try:
            da3, dH, pairs = cell._cell_backward(dHn, a3, b3, H, Wz, Wr, Wh, CZ, CR, CH, Z, R, Ht)
        finally:
            cell.set_fused_backward(True)
        res.append((da3, dH, pairs[0][0], pairs[1][0], pairs[2][0]))
    for name, a, b in zip(("da3", "dH", "dzl", "drl", "dhl"), *res):
        if name in ("dzl", "dhl"):
            assert torch.equal(a, b), name                       # elementwise only
        else:
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5, msg=lambda m, n=name: f"{n}: {m}")
    if N > 3:                                                    # the clamp mask bit: rows 0 and 2 were pushed outside
        assert float(res[0][0][0, :5].abs().max()) == 0.0 and float(res[0][0][2, C + 1].abs()) == 0.0


def test_tgcn_bptt_with_fused_forward_equals_unfused(cuda):
    """hidden = 64 TGCN, 4 steps of BPTT: loss and every gradient with the fused forward chain == without."""
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn.pytorch.temporal import cell
    from stgraph_amd.nn.pytorch.temporal.tgcn import TGCN
    from tests.util import random_graph
    n, e = 6000, 60000
    src, dst = random_graph(8, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    f = g.csr("fwd")
    deg = (f.row_offset[1:] - f.row_offset[:-1]).float()
    g.set_ndata("norm", torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg)).unsqueeze(1))
    w = torch.rand(e, 1, device=cuda) + 0.5
    xs = torch.randn(4, n, 32, device=cuda)
    res = []
    for fused in (True, False):
        torch.manual_seed(5)
        m = TGCN(32, 64).to(cuda)
        cell.set_fused_forward(fused)
        cell.set_fused_backward(fused)
        try:
            Hs, loss = None, 0
            for t in range(4):
                Hs = m(g, xs[t], w, Hs)
                loss = loss + (Hs ** 2).mean()
            loss.backward()
        finally:
            cell.set_fused_forward(True)
            cell.set_fused_backward(True)
        res.append([loss.detach()] + [p.grad.clone() for p in m.parameters()])
    for a, b in zip(*res):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=1e-5)


@pytest.mark.parametrize("N", [1, 15, 16, 17, 4097, 50_000])
@pytest.mark.parametrize("C", [32, 64])
def test_sixteen_row_tiles_match_thirty_two_row_tiles(cuda, N, C):
    """The v_mfma_f32_16x16x4_f32 variant of the forward chain (`cell_rows` = 16): same outputs as the 32-row kernel
    (operand copies bit-identical, GEMM results to fp32 rounding of a different k order)."""
    from stgraph_amd import _C
    from stgraph_amd.nn.pytorch.temporal import cell
    a3, b3, H, (Wz, Wr, Wh), (bz, br, bh) = _operands(cuda, N, C, N + C)
    res = []
    for rows in (16, 32):
        _C.set_tuning("cell_rows", rows)
        try:
            Hn, extra = cell._cell_forward(a3, b3, H, Wz, bz, Wr, br, Wh, bh)
        finally:
            _C.set_tuning("cell_rows", 0)
        res.append((Hn, *extra))
    names = ("Hn", "CZ", "CR", "CH", "Z", "R", "Ht")
    for name, a, b in zip(names, *res):
        if name in ("CZ", "CR"):
            assert torch.equal(a, b), name
        else:
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5, msg=lambda m, n=name: f"{n}: {m}")


@pytest.mark.parametrize("N", [1, 15, 16, 17, 4097, 50_000])
@pytest.mark.parametrize("C", [32, 64])
def test_sixteen_row_backward_matches_thirty_two_row_backward(cuda, N, C):
    from stgraph_amd import _C
    from stgraph_amd.nn.pytorch.temporal import cell
    a3, b3, H, (Wz, Wr, Wh), (bz, br, bh) = _operands(cuda, N, C, 11 * N + C)
    Hn, (CZ, CR, CH, Z, R, Ht) = cell._cell_forward(a3, b3, H, Wz, bz, Wr, br, Wh, bh)
    dHn = torch.randn(N, C, device=cuda)
    res = []
    for rows in (16, 32):
        _C.set_tuning("cell_rows", rows)
        try:
            da3, dH, pairs = cell._cell_backward(dHn, a3, b3, H, Wz, Wr, Wh, CZ, CR, CH, Z, R, Ht)
        finally:
            _C.set_tuning("cell_rows", 0)
        res.append((da3, dH, pairs[0][0], pairs[1][0], pairs[2][0]))
    for name, a, b in zip(("da3", "dH", "dzl", "drl", "dhl"), *res):
        if name in ("dzl", "dhl"):
            assert torch.equal(a, b), name                       # elementwise only
        else:
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5, msg=lambda m, n=name: f"{n}: {m}")
    if N > 3:
        assert float(res[0][0][0, :5].abs().max()) == 0.0 and float(res[0][0][2, C + 1].abs()) == 0.0


@pytest.mark.parametrize("N", [1, 17, 4097, 50_000])
@pytest.mark.parametrize("C", [32, 64])
def test_backward_with_the_input_gradient_product(cuda, N, C):
    """stg_tgcn_cell_fused_bwd_dx: the same five outputs as the plain 16-row backward (bit for bit) plus
    dx = da3 @ Wcat.T."""
    from stgraph_amd import _C, kernels
    from stgraph_amd.nn.pytorch.temporal import cell
    a3, b3, H, (Wz, Wr, Wh), (bz, br, bh) = _operands(cuda, N, C, 13 * N + C)
    Hn, (CZ, CR, CH, Z, R, Ht) = cell._cell_forward(a3, b3, H, Wz, bz, Wr, br, Wh, bh)
    dHn = torch.randn(N, C, device=cuda)
    Wcat = torch.randn(32, 3 * C, device=cuda) * 0.2
    assert kernels.tgcn_cell_fused_bwd_dx_supported(C, 32) and not kernels.tgcn_cell_fused_bwd_dx_supported(C, 48)
    da3, dH, pairs, dx = cell._cell_backward(dHn, a3, b3, H, Wz, Wr, Wh, CZ, CR, CH, Z, R, Ht, Wcat=Wcat)
    _C.set_tuning("cell_rows", 16)
    try:
        da3_p, dH_p, pairs_p = cell._cell_backward(dHn, a3, b3, H, Wz, Wr, Wh, CZ, CR, CH, Z, R, Ht)
    finally:
        _C.set_tuning("cell_rows", 0)
    assert torch.equal(da3, da3_p) and torch.equal(dH, dH_p)
    for (a, _), (b, _) in zip(pairs, pairs_p):
        assert torch.equal(a, b)
    want = (da3.double() @ Wcat.double().t())
    scale = (da3.double().abs() @ Wcat.double().abs().t())
    assert bool(((dx.double() - want).abs() <= 1e-6 * scale + 1e-6).all())
    cell.set_fused_dx(False)
    try:
        _, _, _, none = cell._cell_backward(dHn, a3, b3, H, Wz, Wr, Wh, CZ, CR, CH, Z, R, Ht, Wcat=Wcat)
    finally:
        cell.set_fused_dx(True)
    assert none is None
