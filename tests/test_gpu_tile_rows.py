"""The edge-dealt narrow-row kernel (gcn_agg_tile_kernel: a workgroup's rows = one contiguous edge range, edges dealt
to lane groups, per-edge products staged in an LDS tile, per-row sums in CSR order): same additions in the same order,
so bit-identical to the oracle's sequential loop and to the row-group kernel.  Forced (`gcn_tile` = 2) on small graphs
for every supported width, and taken by itself (auto) on a graph larger than one resident grid.  Each case runs the
one-block-per-workgroup form and the persistent, software-pipelined one (gcn_agg_tile_pipe_kernel, `gcn_tile_pipe` = 2:
several row blocks per workgroup, index prefetch across chunk and block boundaries, the ring of row offsets), with
the automatic and a small forced number of rows per block."""
import numpy as np
import pytest
import torch

from oracle import stg_oracle as orc
from tests.test_gpu_long_rows import skewed_graph
from tests.util import gcn_norm, random_graph

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[(1, 0), (2, 0), (2, 24)], ids=["block", "pipe", "pipe_rows24"])
def forced_tile(request):
    from stgraph_amd import _C
    pipe, rows = request.param
    _C.set_tuning("gcn_tile", 2)
    _C.set_tuning("gcn_tile_pipe", pipe)
    _C.set_tuning("gcn_tile_rows", rows)
    yield
    _C.set_tuning("gcn_tile", 0)
    _C.set_tuning("gcn_tile_pipe", 0)
    _C.set_tuning("gcn_tile_rows", 0)


@pytest.mark.parametrize("F", [4, 5, 6, 7, 8, 9, 12, 13, 16, 20, 24, 31, 32])
@pytest.mark.parametrize("use_ew", [False, True])
def test_forced_tile_kernel_bit_exact(cuda, forced_tile, F, use_ew):
    """Hubs spanning many chunks, threshold degrees, empty rows, ragged widths; forward and backward CSR."""
    from stgraph_amd import _C, kernels
    from stgraph_amd.graph import StaticGraph
    n, e = 1500, 9000
    src, dst = skewed_graph(F, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    og = orc.build_graph(src, dst, n)
    norm_np = gcn_norm(og.in_degrees())
    norm = torch.from_numpy(norm_np).to(cuda)
    rng = np.random.default_rng(F)
    x_np = rng.standard_normal((n, F)).astype(np.float32)
    w_np = (rng.random(len(src)) + 0.5).astype(np.float32)
    x = torch.from_numpy(x_np).to(cuda)
    w = torch.from_numpy(w_np).to(cuda) if use_ew else None
    for side, ocsr in (("fwd", og.fwd), ("bwd", og.bwd)):
        want = orc.gcn_agg(x_np, norm_np, norm_np, ocsr, ew=w_np if use_ew else None)
        got = kernels.gcn_agg(x, norm, norm, g.csr(side), ew=w)
        assert np.array_equal(got.cpu().numpy(), want), side
        _C.set_tuning("gcn_tile", 1)
        try:
            rows = kernels.gcn_agg(x, norm, norm, g.csr(side), ew=w)
        finally:
            _C.set_tuning("gcn_tile", 2)
        assert torch.equal(got, rows), side
    # layer epilogue and a partially active width
    b = torch.randn(F, device=cuda)
    fused = kernels.gcn_agg(x, norm, norm, g.csr("fwd"), ew=w, bias=b, act=kernels.ACT_RELU)
    assert torch.equal(fused, torch.relu(kernels.gcn_agg(x, norm, norm, g.csr("fwd"), ew=w) + b))
    if F > 4:
        fa = F - 1
        part = kernels.gcn_agg(x, norm, norm, g.csr("fwd"), ew=w, f_active=fa)
        full = kernels.gcn_agg(x, norm, norm, g.csr("fwd"), ew=w)
        assert torch.equal(part[:, :fa], full[:, :fa]) and not part[:, fa:].any()


def test_forced_tile_kernel_degenerate_graphs(cuda, forced_tile):
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    # no edges at all; a single row holding every edge; fewer rows than one workgroup takes
    for n, src, dst in ((300, np.zeros(0, np.int32), np.zeros(0, np.int32)),
                        (2000, np.arange(1, 2000, dtype=np.int32), np.zeros(1999, np.int32)),
                        (3, np.array([0, 1, 2, 2], np.int32), np.array([1, 2, 0, 1], np.int32))):
        g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
        og = orc.build_graph(src, dst, n)
        x_np = np.random.default_rng(n).standard_normal((n, 7)).astype(np.float32)
        norm_np = np.random.default_rng(n + 1).uniform(0.5, 1.5, (n, 1)).astype(np.float32)
        x, norm = torch.from_numpy(x_np).to(cuda), torch.from_numpy(norm_np).to(cuda)
        for side, ocsr in (("fwd", og.fwd), ("bwd", og.bwd)):
            got = kernels.gcn_agg(x, norm, norm, g.csr(side))
            assert np.array_equal(got.cpu().numpy(), orc.gcn_agg(x_np, norm_np, norm_np, ocsr)), (n, side)


@pytest.mark.parametrize("F", [4, 7, 8])
def test_large_graph_takes_the_tile_kernel_by_itself(cuda, F):
    """> 2048 workgroups and rows of one or two lanes: the default dispatch; against the row-group kernel."""
    from stgraph_amd import _C, kernels
    n, e = 700_000, 2_800_000
    src, dst = random_graph(F, n, e)
    g = kernels.build_graph_csr(src, dst, n, cuda)
    rng = np.random.default_rng(F)
    x = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(cuda)
    norm = torch.from_numpy(rng.uniform(0.1, 1, (n, 1)).astype(np.float32)).to(cuda)
    records = []
    kernels.enable_launch_timing(records)
    try:
        got = kernels.gcn_agg(x, norm, norm, g.fwd)
    finally:
        kernels.enable_launch_timing(None)
    _C.set_tuning("gcn_tile", 1)
    try:
        rows = kernels.gcn_agg(x, norm, norm, g.fwd)
    finally:
        _C.set_tuning("gcn_tile", 0)
    assert torch.equal(got, rows)
    _C.set_tuning("gcn_tile_pipe", 1)
    try:
        block = kernels.gcn_agg(x, norm, norm, g.fwd)
    finally:
        _C.set_tuning("gcn_tile_pipe", 0)
    assert torch.equal(got, block)
    # in-degree property: all-ones features, unit norms
    ones = torch.ones(n, 1, device=cuda)
    deg = kernels.gcn_agg(torch.ones(n, F, device=cuda), ones, ones, g.fwd)
    assert torch.equal(deg[:, 0], g.in_degrees.float()) and torch.equal(deg[:, 0], deg[:, F - 1])
