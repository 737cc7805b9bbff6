"""GPU parity of the fused GCN aggregation (stg_gcn_agg, called through the C ABI).

Bar: the kernel promises the reference's summation order with two-rounding
multiply-add, so results must be BIT-IDENTICAL to the golden vectors (produced by
the reference's emitted kernels) and to the oracle; the 1e-4 tolerance of the
north star is asserted as well for documentation.
"""
import numpy as np
import pytest
import torch

from oracle import stg_oracle as orc
from tests.util import GCN_WIDTHS, gcn_norm, golden, random_graph

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _dev_graph(src, dst, n, cuda):
    from stgraph_amd import kernels
    return kernels.build_graph_csr(src, dst, n, cuda)


@pytest.mark.parametrize("gname", ["static", "naive"])
@pytest.mark.parametrize("use_ew", [False, True])
@pytest.mark.parametrize("compat", [True, False])
def test_golden_all_widths(cuda, gname, use_ew, compat):
    from stgraph_amd import kernels
    d = golden("gcn.npz")
    n = int(d["num_nodes"])
    g = _dev_graph(d["src"], d["dst"], n, cuda)
    og = orc.build_graph(d["src"], d["dst"], n)
    norm_np = d[f"{gname}_norm"]
    norm = torch.from_numpy(norm_np).to(cuda)
    ew_np = d["edge_weight_by_eid"] if use_ew else None
    ew = torch.from_numpy(ew_np).to(cuda) if use_ew else None
    nid = gname == "naive"
    for F in GCN_WIDTHS:
        tag = f"{gname}_F{F}_{'ew' if use_ew else 'now'}"
        fa_ref = kernels.ref_active_columns(F)
        fa = fa_ref if compat else F
        x = torch.from_numpy(d[tag + "_x"]).to(cuda)
        R = torch.from_numpy(d[tag + "_R"]).to(cuda)
        out = kernels.gcn_agg(x, norm, norm, g.fwd, ew=ew, use_node_ids=nid, f_active=fa).cpu().numpy()
        gx = kernels.gcn_agg(R, norm, norm, g.bwd, ew=ew, use_node_ids=nid, f_active=fa).cpu().numpy()
        if compat:      # the reference's exact output, zero tail included (defect D1)
            assert np.array_equal(out, d[tag + "_out"]), tag
            assert np.array_equal(gx, d[tag + "_grad_x"]), tag
        else:           # reference columns unchanged; the remaining columns equal the complete math
            assert np.array_equal(out[:, :fa_ref], d[tag + "_out"][:, :fa_ref]), tag
            assert np.array_equal(gx[:, :fa_ref], d[tag + "_grad_x"][:, :fa_ref]), tag
            full = orc.gcn_agg(d[tag + "_x"], norm_np, norm_np, og.fwd, ew=ew_np, use_node_ids=nid)
            assert np.array_equal(out, full), tag
        np.testing.assert_allclose(out[:, :fa_ref], d[tag + "_out"][:, :fa_ref], rtol=TOL, atol=TOL)


@pytest.mark.parametrize("F", [1, 2, 3, 5, 6, 8, 9, 13, 16, 33, 64, 65, 96, 128, 129, 200, 257, 512, 1100])
@pytest.mark.parametrize("use_ew", [False, True])
def test_oracle_random_graph(cuda, F, use_ew):
    from stgraph_amd import kernels
    n, e = 3000, 40000
    src, dst = random_graph(100 + F, n, e)
    g = _dev_graph(src, dst, n, cuda)
    og = orc.build_graph(src, dst, n)
    rng = np.random.default_rng(F)
    x = rng.standard_normal((n, F)).astype(np.float32)
    norm = gcn_norm(og.in_degrees())
    ew = rng.uniform(0.5, 1.5, (len(src), 1)).astype(np.float32) if use_ew else None
    for csr, ocsr in ((g.fwd, og.fwd), (g.bwd, og.bwd)):
        for nid in (False, True):
            got = kernels.gcn_agg(torch.from_numpy(x).to(cuda), torch.from_numpy(norm).to(cuda),
                                  torch.from_numpy(norm).to(cuda), csr,
                                  ew=None if ew is None else torch.from_numpy(ew).to(cuda),
                                  use_node_ids=nid).cpu().numpy()
            want = orc.gcn_agg(x, norm, norm, ocsr, ew=ew, use_node_ids=nid)
            assert np.array_equal(got, want), (F, use_ew, nid)


def test_edge_cases(cuda):
    from stgraph_amd import kernels
    # no edges at all, one vertex, self loop only, isolated vertices
    for n, edges in ((1, []), (1, [(0, 0)]), (5, []), (6, [(2, 2), (3, 4)])):
        src = np.array([a for a, _ in edges], np.int32)
        dst = np.array([b for _, b in edges], np.int32)
        g = _dev_graph(src, dst, n, cuda)
        og = orc.build_graph(src, dst, n)
        x = np.arange(n * 5, dtype=np.float32).reshape(n, 5) + 1
        norm = np.full((n, 1), 0.5, np.float32)
        got = kernels.gcn_agg(torch.from_numpy(x).to(cuda), torch.from_numpy(norm).to(cuda),
                              torch.from_numpy(norm).to(cuda), g.fwd).cpu().numpy()
        assert np.array_equal(got, orc.gcn_agg(x, norm, norm, og.fwd))


def test_tuning_knobs_do_not_change_results(cuda):
    from stgraph_amd import _C, kernels
    n, e, F = 5000, 80000, 128
    src, dst = random_graph(7, n, e)
    g = _dev_graph(src, dst, n, cuda)
    rng = np.random.default_rng(7)
    x = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(cuda)
    norm = torch.from_numpy(rng.uniform(0.1, 1, (n, 1)).astype(np.float32)).to(cuda)
    base = kernels.gcn_agg(x, norm, norm, g.fwd)
    try:
        for lanes in (16, 32, 64):
            for unroll in (2, 4, 8):
                _C.set_tuning("gcn_lanes_per_row", lanes)
                _C.set_tuning("gcn_unroll", unroll)
                assert torch.equal(kernels.gcn_agg(x, norm, norm, g.fwd), base), (lanes, unroll)
    finally:
        _C.set_tuning("gcn_lanes_per_row", 0)
        _C.set_tuning("gcn_unroll", 0)


@pytest.mark.parametrize("F", [7, 16, 32, 100])
def test_launch_mapping_knobs_do_not_change_results(cuda, F):
    """Workgroup size, XCD runs and 32-bit gather offsets only move work around: bit-identical output, on a graph
    large enough (> 2048 workgroups) to take the plain launch and on a small one (merged long-row launch), with and
    without the degree-sorted row order, and against the oracle."""
    from stgraph_amd import _C, kernels
    rng = np.random.default_rng(F)
    for n, e in ((300_000, 1_200_000), (3000, 20000)):
        src, dst = random_graph(F, n, e)
        g = _dev_graph(src, dst, n, cuda)
        x_np = rng.standard_normal((n, F)).astype(np.float32)
        norm_np = rng.uniform(0.1, 1, (n, 1)).astype(np.float32)
        x, norm = torch.from_numpy(x_np).to(cuda), torch.from_numpy(norm_np).to(cuda)
        base = kernels.gcn_agg(x, norm, norm, g.fwd)
        if n <= 3000:
            og = orc.build_graph(src, dst, n)
            assert np.array_equal(base.cpu().numpy(), orc.gcn_agg(x_np, norm_np, norm_np, og.fwd))
        try:
            for block in (64, 128, 256):
                for tile in (1, 3, 64):
                    for a32 in (0, 1):
                        _C.set_tuning("gcn_block", block)
                        _C.set_tuning("gcn_xcd_tile", tile)
                        _C.set_tuning("gcn_addr32", a32)
                        for nid in (False, True):
                            got = kernels.gcn_agg(x, norm, norm, g.fwd, use_node_ids=nid)
                            assert torch.equal(got, base), (n, block, tile, a32, nid)
        finally:
            _C.set_tuning("gcn_block", 0)
            _C.set_tuning("gcn_xcd_tile", 0)
            _C.set_tuning("gcn_addr32", 0)


@pytest.mark.parametrize("n,F", [(1 << 24, 4), (1 << 24, 32), ((1 << 24) + 5, 4), (40_000_000, 16), (30_000_000, 32)])
def test_32_bit_gather_offsets_at_their_limits(cuda, n, F):
    """Narrow rows use 24-bit x 24-bit offset arithmetic when |V| <= 2^24 and the matrix is < 4 GB: the largest
    such shapes, the first ones beyond (64-bit path), edges into the last rows; against the 64-bit path and a
    torch restatement of the touched rows."""
    from stgraph_amd import _C, kernels
    gen = torch.Generator(device=cuda).manual_seed(n % 1000)
    e = 300_000
    src = torch.randint(0, n, (e,), generator=gen, device=cuda, dtype=torch.int32)
    src[:1000] = n - 1 - torch.arange(1000, device=cuda, dtype=torch.int32)          # the last rows as sources
    dst = torch.randint(0, n, (e,), generator=gen, device=cuda, dtype=torch.int32)
    dst[:500] = n - 1                                                                 # ... and a hub at the last row
    keys = torch.unique(src.long() * n + dst.long())
    src, dst = (keys // n).int(), (keys % n).int()
    g = kernels.build_graph_csr(src, dst, n, cuda)
    x = torch.randn(n, F, device=cuda, generator=gen)
    ones = torch.ones(n, 1, device=cuda)
    got = kernels.gcn_agg(x, ones, ones, g.fwd)
    _C.set_tuning("gcn_addr32", 1)
    try:
        wide = kernels.gcn_agg(x, ones, ones, g.fwd)
    finally:
        _C.set_tuning("gcn_addr32", 0)
    assert torch.equal(got, wide)
    want = torch.zeros(n, F, device=cuda, dtype=torch.float64)
    want.index_add_(0, dst.long(), x[src.long()].double())
    rows = torch.unique(dst.long())
    torch.testing.assert_close(got[rows].double(), want[rows], rtol=1e-5, atol=1e-5)
    assert not got[n - 2].any() or (dst == n - 2).any()
    del x, got, wide, want
    torch.cuda.empty_cache()


def test_unaligned_operands(cuda):
    """x / out at a 4-byte (not 16-byte) aligned address, as a slice of a larger buffer can be."""
    from stgraph_amd import kernels
    n, e = 2000, 16000
    src, dst = random_graph(3, n, e)
    g = _dev_graph(src, dst, n, cuda)
    og = orc.build_graph(src, dst, n)
    rng = np.random.default_rng(3)
    norm_np = rng.uniform(0.1, 1, (n, 1)).astype(np.float32)
    norm = torch.from_numpy(norm_np).to(cuda)
    for F in (4, 7, 16, 33):
        x_np = rng.standard_normal((n, F)).astype(np.float32)
        buf = torch.zeros(n * F + 3, device=cuda)
        for shift in (1, 2, 3):
            x = buf[shift:shift + n * F].view(n, F)
            x.copy_(torch.from_numpy(x_np))
            assert x.data_ptr() % 16 != 0
            got = kernels.gcn_agg(x, norm, norm, g.fwd)
            assert np.array_equal(got.cpu().numpy(), orc.gcn_agg(x_np, norm_np, norm_np, og.fwd)), (F, shift)


def test_full_size_properties(cuda):
    """BASELINE config 2 shape (|V|=1M, |E|=16M, F=128): size-independent properties."""
    from stgraph_amd import kernels
    n, e, F = 1_000_000, 16_000_000, 128
    gen = torch.Generator(device=cuda).manual_seed(1)
    src = torch.randint(0, n, (e,), generator=gen, device=cuda, dtype=torch.int32)
    dst = torch.randint(0, n, (e,), generator=gen, device=cuda, dtype=torch.int32)
    g = kernels.build_graph_csr(src, dst, n, cuda)
    ones = torch.ones(n, 1, device=cuda)
    # (1) aggregating all-ones features with unit norms counts in-degrees exactly
    deg = kernels.gcn_agg(torch.ones(n, F, device=cuda), ones, ones, g.fwd)
    assert torch.equal(deg[:, 0], g.in_degrees.float()) and torch.equal(deg[:, 0], deg[:, F - 1])
    # (2) adjoint identity <A x, y> == <x, A^T y> ties the forward and backward CSR together
    x = torch.randn(n, F, device=cuda, generator=gen)
    y = torch.randn(n, F, device=cuda, generator=gen)
    norm = torch.rand(n, 1, device=cuda, generator=gen) + 0.5
    Ax = kernels.gcn_agg(x, norm, norm, g.fwd)
    ATy = kernels.gcn_agg(y, norm, norm, g.bwd)
    lhs, rhs = (Ax.double() * y.double()).sum(), (x.double() * ATy.double()).sum()
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), abs(rhs), 1.0)
    # (3) linearity in x (exact for a power-of-two scale) and run-to-run determinism
    assert torch.equal(kernels.gcn_agg(2 * x, norm, norm, g.fwd), 2 * Ax)
    assert torch.equal(kernels.gcn_agg(x, norm, norm, g.fwd), Ax)
    # (4) a 4096-row sample against the oracle's sequential sum, bit for bit
    rows = torch.randint(0, n, (4096,), generator=gen, device=cuda).cpu().numpy()
    ro = g.fwd.row_offset.cpu().numpy()
    col = g.fwd.column_indices.cpu().numpy()
    xn, nn_ = x.cpu().numpy(), norm.cpu().numpy().reshape(-1)
    Axn = Ax[torch.from_numpy(rows).to(cuda)].cpu().numpy()
    for i, r in enumerate(rows[:512]):
        acc = np.zeros(F, np.float32)
        for c in col[ro[r]:ro[r + 1]]:
            acc = acc + nn_[c] * xn[c]
        assert np.array_equal(acc * nn_[r], Axn[i])


@pytest.mark.parametrize("compat", [True, False])
def test_golden_n200_hub_selfloops_isolated(cuda, compat):
    """Reference-generated vectors on N = 200 with a hub of in-degree >= 90, self-loops, isolated vertices, all widths."""
    from stgraph_amd import kernels
    d = golden("gcn_n200.npz")
    n = int(d["num_nodes"])
    g = _dev_graph(d["src"], d["dst"], n, cuda)
    og = orc.build_graph(d["src"], d["dst"], n)
    norm = torch.from_numpy(d["norm"]).to(cuda)
    for F in GCN_WIDTHS:
        fa_ref = kernels.ref_active_columns(F)
        fa = fa_ref if compat else F
        for use_ew in (False, True):
            tag = f"F{F}_{'ew' if use_ew else 'now'}"
            ew = torch.from_numpy(d["edge_weight_by_eid"]).to(cuda) if use_ew else None
            out = kernels.gcn_agg(torch.from_numpy(d[tag + "_x"]).to(cuda), norm, norm, g.fwd, ew=ew, f_active=fa).cpu().numpy()
            gx = kernels.gcn_agg(torch.from_numpy(d[tag + "_R"]).to(cuda), norm, norm, g.bwd, ew=ew, f_active=fa).cpu().numpy()
            assert np.array_equal(out[:, :fa_ref], d[tag + "_out"][:, :fa_ref]), tag
            assert np.array_equal(gx[:, :fa_ref], d[tag + "_grad_x"][:, :fa_ref]), tag
            if compat:
                assert np.array_equal(out, d[tag + "_out"]) and np.array_equal(gx, d[tag + "_grad_x"]), tag
            else:
                w = d["edge_weight_by_eid"] if use_ew else None
                assert np.array_equal(out, orc.gcn_agg(d[tag + "_x"], d["norm"], d["norm"], og.fwd, ew=w)), tag


def test_golden_cora_shaped(cuda):
    """Reference-generated vectors on the benchmark's Cora-shaped graph (N = 2708, E = 10556, hub of in-degree 159):
    sampled rows in full and the fp64 column sums over all rows."""
    from stgraph_amd import kernels
    d = golden("gcn_cora.npz")
    n = int(d["num_nodes"])
    g = _dev_graph(d["src"], d["dst"], n, cuda)
    norm, rows = torch.from_numpy(d["norm"]).to(cuda), d["rows"]
    for F in (7, 16, 64, 300):
        fa = kernels.ref_active_columns(F)
        for use_ew in (False, True):
            tag = f"F{F}_{'ew' if use_ew else 'now'}"
            rng = np.random.default_rng(int(d[tag + "_seed"]))
            x = rng.standard_normal((n, F), dtype=np.float32)
            R = rng.standard_normal((n, F), dtype=np.float32)
            ew = torch.from_numpy(d["edge_weight_by_eid"]).to(cuda) if use_ew else None
            out = kernels.gcn_agg(torch.from_numpy(x).to(cuda), norm, norm, g.fwd, ew=ew, f_active=fa).cpu().numpy()
            gx = kernels.gcn_agg(torch.from_numpy(R).to(cuda), norm, norm, g.bwd, ew=ew, f_active=fa).cpu().numpy()
            assert np.array_equal(out[rows], d[tag + "_out_rows"]) and np.array_equal(gx[rows], d[tag + "_grad_x_rows"]), tag
            assert np.array_equal(out.astype(np.float64).sum(0), d[tag + "_out_colsum"]), tag
            assert np.array_equal(gx.astype(np.float64).sum(0), d[tag + "_grad_x_colsum"]), tag
