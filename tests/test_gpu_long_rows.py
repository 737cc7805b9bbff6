"""Hub rows (more than 16 edges) of narrow feature rows go through the wave-per-row, LDS-staged launch
(gcn_agg_long_kernel): same additions in the same order, so results stay bit-identical to the oracle's
sequential loop and to the main kernel with the long-row path switched off."""
import numpy as np
import pytest
import torch

from oracle import stg_oracle as orc
from tests.util import gcn_norm

pytestmark = pytest.mark.gpu


def skewed_graph(seed, n, e, hubs=6):
    """Duplicate-free edges with a few hub destinations AND hub sources (degrees 33 ... ~n/3) next to a
    low-degree bulk, plus rows of degree exactly 16 and 17 (the threshold) and 128 / 129 (one full batch at F = 16)."""
    rng = np.random.default_rng(seed)
    pairs = set()
    for h in range(hubs):
        deg = [17, 16, 128, 129, 168, max(40, n // 3)][h % 6]
        deg = min(deg, n - 1)
        others = rng.choice(n, size=deg, replace=False)
        for o in others:
            pairs.add((int(o), h))                 # hub destination h
            pairs.add((hubs + h, int(o)))          # hub source hubs + h
    while len(pairs) < e:
        a, b = rng.integers(0, n, 2)
        pairs.add((int(a), int(b)))
    arr = np.array(sorted(pairs), dtype=np.int32)
    rng.shuffle(arr)
    return arr[:, 0].copy(), arr[:, 1].copy()


@pytest.mark.parametrize("F", [1, 2, 3, 4, 5, 6, 7, 8, 9, 13, 16, 24, 32, 50, 64, 100, 128])
@pytest.mark.parametrize("use_ew", [False, True])
def test_long_rows_bit_exact(cuda, F, use_ew):
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    n, e = 700, 5000
    src, dst = skewed_graph(F, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    og = orc.build_graph(src, dst, n)
    norm_np = gcn_norm(og.in_degrees())
    norm = torch.from_numpy(norm_np).to(cuda)
    rng = np.random.default_rng(1)
    x_np = rng.standard_normal((n, F)).astype(np.float32)
    w_np = (rng.random(len(src)) + 0.5).astype(np.float32)
    x = torch.from_numpy(x_np).to(cuda)
    w = torch.from_numpy(w_np).to(cuda) if use_ew else None
    assert g.csr("fwd").degree_sorted and g.csr("bwd").degree_sorted       # what enables the long-row workgroups
    assert int((g.csr("fwd").row_offset[1:] - g.csr("fwd").row_offset[:-1]).max()) > 32
    assert int((g.csr("bwd").row_offset[1:] - g.csr("bwd").row_offset[:-1]).max()) > 32
    for side, ocsr in (("fwd", og.fwd), ("bwd", og.bwd)):
        for nid in (False, True):
            want = orc.gcn_agg(x_np, norm_np, norm_np, ocsr, ew=w_np if use_ew else None, use_node_ids=nid)
            got = kernels.gcn_agg(x, norm, norm, g.csr(side), ew=w, use_node_ids=nid)
            assert np.array_equal(got.cpu().numpy(), want), (side, nid)
            kernels.set_long_row_path(False)
            try:
                plain = kernels.gcn_agg(x, norm, norm, g.csr(side), ew=w, use_node_ids=nid)
            finally:
                kernels.set_long_row_path(True)
            assert torch.equal(got, plain)
    # with the layer epilogue
    b = torch.randn(F, device=cuda)
    fused = kernels.gcn_agg(x, norm, norm, g.csr("fwd"), ew=w, bias=b, act=kernels.ACT_RELU)
    assert torch.equal(fused, torch.relu(kernels.gcn_agg(x, norm, norm, g.csr("fwd"), ew=w) + b))


def test_long_rows_every_row_long_and_empty_graph(cuda):
    """A complete bipartite-like block (every row has 200 edges) and a graph without edges."""
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    n = 256
    src = np.repeat(np.arange(n, dtype=np.int32), 200)
    dst = ((np.tile(np.arange(200, dtype=np.int32), n) * 7 + np.repeat(np.arange(n, dtype=np.int32), 200)) % n).astype(np.int32)
    pair = np.unique(np.stack([src, dst], 1), axis=0)
    src, dst = pair[:, 0].copy(), pair[:, 1].copy()
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    og = orc.build_graph(src, dst, n)
    norm_np = gcn_norm(og.in_degrees())
    x_np = np.random.default_rng(0).standard_normal((n, 16)).astype(np.float32)
    got = kernels.gcn_agg(torch.from_numpy(x_np).to(cuda), torch.from_numpy(norm_np).to(cuda),
                          torch.from_numpy(norm_np).to(cuda), g.csr("fwd"))
    assert np.array_equal(got.cpu().numpy(), orc.gcn_agg(x_np, norm_np, norm_np, og.fwd))


def test_skewed_graph_at_scale_properties(cuda):
    """Cora-shaped degree skew replicated 512 times (block diagonal, hubs of degree 168): long-row path ==
    main kernel bit for bit, rows resummed sequentially on the host for a sample."""
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    base_n, reps = 2708, 512
    s0, d0 = skewed_graph(3, base_n, 10556)
    off = (np.arange(reps, dtype=np.int64) * base_n)[:, None]
    src = (s0[None, :] + off).reshape(-1).astype(np.int32)
    dst = (d0[None, :] + off).reshape(-1).astype(np.int32)
    n = base_n * reps
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    f = g.csr("fwd")
    deg = (f.row_offset[1:] - f.row_offset[:-1]).float()
    norm = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg)).unsqueeze(1)
    for F in (7, 16):
        x = torch.randn(n, F, device=cuda)
        a = kernels.gcn_agg(x, norm, norm, f)
        kernels.set_long_row_path(False)
        try:
            b = kernels.gcn_agg(x, norm, norm, f)
        finally:
            kernels.set_long_row_path(True)
        assert torch.equal(a, b)
        ro, col = f.row_offset.cpu().numpy(), f.column_indices.cpu().numpy()
        xh, nh, ah = x.cpu().numpy(), norm.cpu().numpy().reshape(-1), a.cpu().numpy()
        for r in list(np.argsort(-deg.cpu().numpy())[:8]) + [5, 1000, n - 1]:
            acc = np.zeros(F, np.float32)
            for e in range(ro[r], ro[r + 1]):
                acc = acc + nh[col[e]] * xh[col[e]]
            assert np.array_equal(ah[r], acc * nh[r])


def test_hand_built_csr_keeps_every_row_on_the_row_group_path(cuda):
    """A DeviceCSR assembled by hand (node_ids in arbitrary order) must not be searched for long rows through it."""
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    n = 500
    src, dst = skewed_graph(9, n, 4000)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    f = g.csr("fwd")
    shuffled = f.node_ids[torch.randperm(n, device=cuda)]
    mine = kernels.DeviceCSR(f.row_offset, f.column_indices, f.eids, shuffled)
    assert not mine.degree_sorted
    x = torch.randn(n, 16, device=cuda)
    norm = torch.rand(n, 1, device=cuda) + 0.5
    assert torch.equal(kernels.gcn_agg(x, norm, norm, mine, use_node_ids=True), kernels.gcn_agg(x, norm, norm, f))


def hub_graph(seed, n, hub_degrees, bulk_edges):
    """Hubs of the given in-degrees (destinations 0, 1, ...) and the same hubs as sources, over a random low-degree bulk."""
    rng = np.random.default_rng(seed)
    src, dst = [], []
    for h, deg in enumerate(hub_degrees):
        others = rng.choice(np.arange(len(hub_degrees), n), size=deg, replace=False).astype(np.int64)
        src += [others, np.full(deg, h, np.int64)]
        dst += [np.full(deg, h, np.int64), others]
    bs, bd = rng.integers(len(hub_degrees), n, bulk_edges), rng.integers(len(hub_degrees), n, bulk_edges)
    keys = np.unique(np.concatenate([np.concatenate(src) * n + np.concatenate(dst), bs * n + bd]))
    rng.shuffle(keys)
    return (keys // n).astype(np.int32), (keys % n).astype(np.int32)


@pytest.mark.parametrize("F,use_ew,epilogue", [(128, False, False), (128, True, False), (128, False, True), (256, True, False),
                                               (64, False, False), (200, False, False), (100, False, False), (300, False, False),
                                               (300, True, True), (131, False, False), (257, True, False), (1030, False, True)])
def test_wide_hub_rows_bit_exact(cuda, F, use_ew, epilogue):
    """Rows of a whole wave and wider (F >= 128) with hubs of 50 000 / 20 000 / 5 000 / 1 500 / 1 025 / 1 024 in-edges:
    the feature-sliced long-row workgroups (gcn_agg_wide_long_kernel: 16 / 4 / 1 slices by row length, columns in super-blocks
    of <= 256 floats, a width that is not a multiple of 4 closed by an overlapping 16-byte window) give the oracle's sequential
    sums bit for bit, forward and backward CSR, with edge weights and with the layer epilogue -- for EVERY width: 300 and
    1030 (several super-blocks), 131 / 257 (not multiples of 4; 257 leaves a 125-float second super-block); F = 64 and 100
    (the one-wave long-row path of narrower rows) ride along."""
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    n = 60_000
    src, dst = hub_graph(F, n, [50_000, 20_000, 5_000, 1_500, 1_025, 1_024], 150_000)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    og = orc.build_graph(src, dst, n)
    norm_np = gcn_norm(og.in_degrees())
    norm = torch.from_numpy(norm_np).to(cuda)
    rng = np.random.default_rng(2)
    x_np = rng.standard_normal((n, F)).astype(np.float32)
    w_np = (rng.random(len(src)) + 0.5).astype(np.float32)
    b_np = rng.standard_normal(F).astype(np.float32)
    x = torch.from_numpy(x_np).to(cuda)
    w = torch.from_numpy(w_np).to(cuda) if use_ew else None
    assert g.csr("fwd").degree_sorted and int(g.csr("fwd").node_ids[0]) == 0
    for side, ocsr in (("fwd", og.fwd), ("bwd", og.bwd)):
        want = orc.gcn_agg(x_np, norm_np, norm_np, ocsr, ew=w_np if use_ew else None, omp=True)
        if epilogue:
            want = np.maximum(want + b_np, np.float32(0))
            got = kernels.gcn_agg(x, norm, norm, g.csr(side), ew=w, bias=torch.from_numpy(b_np).to(cuda), act=kernels.ACT_RELU)
        else:
            got = kernels.gcn_agg(x, norm, norm, g.csr(side), ew=w)
        assert np.array_equal(got.cpu().numpy(), want), side


def test_wide_hub_rows_replayed_from_a_hip_graph(cuda):
    """The hubs' launch runs on a helper stream beside the main launch (fork / join through two events); captured in a HIP
    graph those become graph edges: the replay gives the eager bits."""
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    n, F = 30_000, 128
    src, dst = hub_graph(5, n, [20_000, 3_000, 1_100], 60_000)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    deg = np.bincount(dst, minlength=n)
    norm = torch.from_numpy(gcn_norm(deg)).to(cuda)
    x = torch.randn(n, F, device=cuda)
    csr = g.csr("fwd")
    eager = kernels.gcn_agg(x, norm, norm, csr)
    assert kernels._hub_plan(csr) == (1, 1, 1)
    side = torch.cuda.Stream(device=cuda)
    side.wait_stream(torch.cuda.current_stream(cuda))
    with torch.cuda.stream(side):
        kernels.gcn_agg(x, norm, norm, csr)
    torch.cuda.current_stream(cuda).wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = kernels.gcn_agg(x, norm, norm, csr)
    for _ in range(3):
        out.zero_()
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)
