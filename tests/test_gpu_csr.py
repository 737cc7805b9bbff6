"""GPU parity of the device CSR builder (stg_graph_build_device): bit-exact integers."""
import numpy as np
import pytest
import torch

from oracle import stg_oracle as orc
from tests.util import golden, random_graph

pytestmark = pytest.mark.gpu
KEYS = ("row_offset", "column_indices", "eids")


def _check(g, og):
    for side, c, oc in (("fwd", g.fwd, og.fwd), ("bwd", g.bwd, og.bwd)):
        for k in KEYS + ("node_ids",):
            assert np.array_equal(getattr(c, k).cpu().numpy(), getattr(oc, k)), (side, k)
    assert np.array_equal(g.in_degrees.cpu().numpy(), og.in_degrees())
    assert np.array_equal(g.out_degrees.cpu().numpy(), og.out_degrees())
    assert np.array_equal(g.perm_fwd.cpu().numpy(), og.perm_fwd)


@pytest.mark.parametrize("tag", ["n1", "n5", "n64", "n2708"])
def test_golden(cuda, tag):
    from stgraph_amd import kernels
    d = golden(f"csr_{tag}.npz")
    n = int(d["num_nodes"])
    g = kernels.build_graph_csr(d["src"], d["dst"], n, cuda)
    for side, c in (("fwd", g.fwd), ("bwd", g.bwd)):
        for k in KEYS:
            assert np.array_equal(getattr(c, k).cpu().numpy(), d[f"{side}_{k}"]), (side, k)
        # node_ids: the reference's tie order is unspecified (std::sort) => permutation + monotone degrees
        nid = c.node_ids.cpu().numpy()
        deg = np.diff(c.row_offset.cpu().numpy())
        assert sorted(nid.tolist()) == list(range(n)) and np.all(np.diff(deg[nid]) <= 0)
    assert np.array_equal(g.in_degrees.cpu().numpy(), d["in_degrees"])
    assert np.array_equal(g.out_degrees.cpu().numpy(), d["out_degrees"])
    pair = np.stack([d["src"], d["dst"]], 1)[g.perm_fwd.cpu().numpy()]
    assert np.array_equal(pair, d["sorted_inplace"])


@pytest.mark.parametrize("n,e,dup", [(1, 0, False), (7, 0, False), (2, 4, False), (50, 600, True),
                                     (1000, 20000, False), (70000, 300000, True), (300000, 2000000, False)])
def test_oracle_random(cuda, n, e, dup):
    from stgraph_amd import kernels
    if e == 0:
        src = dst = np.empty(0, np.int32)
    else:
        src, dst = random_graph(n + e, n, e, duplicates=dup, hub=e > 100)
    g = kernels.build_graph_csr(src, dst, n, cuda)
    _check(g, orc.build_graph(src, dst, n))


def test_vertex_out_of_range_is_reported(cuda):
    from stgraph_amd import kernels
    with pytest.raises(ValueError):
        kernels.build_graph_csr(np.array([0, 5], np.int32), np.array([1, 1], np.int32), 3, cuda)


def test_full_size_properties(cuda):
    """16M-edge build: sortedness, permutation and degree checksums instead of a CPU recomputation."""
    from stgraph_amd import kernels
    n, e = 1_000_000, 16_000_000
    gen = torch.Generator(device=cuda).manual_seed(1)
    src = torch.randint(0, n, (e,), generator=gen, device=cuda, dtype=torch.int32)
    dst = torch.randint(0, n, (e,), generator=gen, device=cuda, dtype=torch.int32)
    g = kernels.build_graph_csr(src, dst, n, cuda)
    for csr, rows_in, cols_in in ((g.fwd, dst, src), (g.bwd, src, dst)):
        ro = csr.row_offset.long()
        assert ro[0] == 0 and ro[-1] == e and bool((ro[1:] >= ro[:-1]).all())
        rows = torch.repeat_interleave(torch.arange(n, device=cuda), ro[1:] - ro[:-1])
        key = rows * n + csr.column_indices.long()
        assert bool((key[1:] >= key[:-1]).all())                       # (row, col) sorted
        assert torch.equal(torch.sort(rows_in.long() * n + cols_in.long()).values, key)   # same multiset
    assert torch.equal(g.fwd.eids.long(), torch.arange(e, device=cuda))
    assert torch.equal(torch.sort(g.bwd.eids.long()).values, torch.arange(e, device=cuda))
    # eids of the backward CSR point at the forward position of the same (src, dst) pair
    rows_b = torch.repeat_interleave(torch.arange(n, device=cuda), (g.bwd.row_offset[1:] - g.bwd.row_offset[:-1]).long())
    rows_f = torch.repeat_interleave(torch.arange(n, device=cuda), (g.fwd.row_offset[1:] - g.fwd.row_offset[:-1]).long())
    be = g.bwd.eids.long()
    assert torch.equal(g.fwd.column_indices.long()[be], rows_b) and torch.equal(rows_f[be], g.bwd.column_indices.long())
    assert torch.equal(src.long()[g.perm_fwd], g.fwd.column_indices.long())
    assert int(g.in_degrees.sum()) == e and int(g.out_degrees.sum()) == e
    for csr in (g.fwd, g.bwd):
        deg = (csr.row_offset[1:] - csr.row_offset[:-1])[csr.node_ids.long()]
        assert bool((deg[1:] <= deg[:-1]).all())
        assert torch.equal(torch.sort(csr.node_ids.long()).values, torch.arange(n, device=cuda))


@pytest.mark.parametrize("n,e,dup", [(1, 1, False), (7, 40, True), (300, 5000, True), (25_000, 250_000, False)])
def test_direct_build_equals_sort_build_and_lazy_node_ids(cuda, n, e, dup):
    """The counting build (stg_graph_build_direct_device) and the sort-based build give the same arrays, also with
    duplicate edges / self loops; node_ids left to first use equal the eagerly sorted ones; a row longer than the
    direct path's limit falls back to the sort-based build transparently."""
    from stgraph_amd import kernels
    src, dst = random_graph(n + e, n, e, duplicates=dup, hub=False)       # rows stay under the direct path's limit
    res = []
    for direct in (True, False):
        kernels.set_direct_build(direct)
        try:
            res.append(kernels.build_graph_csr(src, dst, n, cuda))
        finally:
            kernels.set_direct_build(True)
    a, b = res
    for x, y in ((a.fwd, b.fwd), (a.bwd, b.bwd)):
        for k in ("row_offset", "column_indices", "eids", "node_ids"):
            assert torch.equal(getattr(x, k), getattr(y, k)), k
    assert torch.equal(a.perm_fwd, b.perm_fwd) and torch.equal(a.in_degrees, b.in_degrees)
    lazy = kernels.build_graph_csr(src, dst, n, cuda, lazy_node_ids=True)
    assert lazy.fwd.node_ids_if_ready is None and not lazy.fwd.degree_sorted
    assert torch.equal(lazy.fwd.node_ids, a.fwd.node_ids) and torch.equal(lazy.bwd.node_ids, a.bwd.node_ids)
    assert lazy.fwd.degree_sorted and lazy.fwd.node_ids_if_ready is not None


def test_direct_build_falls_back_on_a_long_row(cuda):
    from stgraph_amd import kernels
    n = 5000
    hub = np.arange(1, 3001, dtype=np.int32)                        # 3000 edges into vertex 0: longer than 2048
    src = np.concatenate([hub, np.arange(n - 1, dtype=np.int32)])
    dst = np.concatenate([np.zeros(3000, np.int32), np.arange(1, n, dtype=np.int32)])
    g = kernels.build_graph_csr(src, dst, n, cuda, lazy_node_ids=True)
    og = orc.build_graph(src, dst, n)
    for side, o in ((g.fwd, og.fwd), (g.bwd, og.bwd)):
        for k in ("row_offset", "column_indices", "eids"):
            assert np.array_equal(getattr(side, k).cpu().numpy(), getattr(o, k)), k
    assert g.fwd.degree_sorted and int(g.fwd.node_ids[0]) == 0


def test_rebuilt_snapshots_are_verified_in_bulk(cuda):
    """NaiveGraph(resident=False) rebuilds a validated snapshot without reading the build's status word; the words are
    checked together at the next reset_graph -- so an edge list that changed under it (here: an endpoint pushed out of
    range in place) is still reported."""
    from stgraph_amd.graph import NaiveGraph
    n = 500
    rng = np.random.default_rng(0)
    snaps = []
    for t in range(3):
        keys = rng.choice(n * n, size=4000, replace=False)
        snaps.append((torch.from_numpy((keys // n).astype(np.int32)).to(cuda), torch.from_numpy((keys % n).astype(np.int32)).to(cuda)))
    G = NaiveGraph(snaps, n, device=cuda, sort_inplace=False, resident=False, max_cached=1)
    first = [G.csr("fwd", t).row_offset.clone() for t in range(3)]          # first builds: validated, status read
    G._snapshots.clear()
    again = [G.csr("fwd", t).row_offset.clone() for t in range(3)]          # rebuilds: status read skipped
    assert all(torch.equal(a, b) for a, b in zip(first, again))
    assert 1 <= len(G._pending_status) <= 3          # the fused rebuild reports through ONE sticky word per device
    G.reset_graph()                                                          # bulk check passes (and moves to t = 0)
    G.verify_builds()
    assert not G._pending_status
    G._edges[1][0][7] = n + 3                                                # corrupt snapshot 1 in place
    G._snapshots.clear()
    G.csr("fwd", 1)
    with pytest.raises(ValueError, match="deferred"):
        G.reset_graph()
    G._edges[1][0][7] = 3                                                    # repaired: the sticky word was cleared by the report
    G._snapshots.clear()
    G.csr("fwd", 1)
    G.reset_graph()


def test_a_failed_batched_rebuild_names_its_snapshot(cuda):
    """The sticky status word is one per device: the batched rebuild leaves the id of the FIRST edge list that failed its
    validation beside the code (stg_build_job::id = timestamp + 1), and the bulk check names that timestamp (the failure
    used to be reported without saying whose it was)."""
    from stgraph_amd import kernels
    from stgraph_amd.graph import NaiveGraph
    n = 800
    rng = np.random.default_rng(3)
    snaps = []
    for t in range(6):
        keys = rng.choice(n * n, size=6000, replace=False)
        snaps.append((torch.from_numpy((keys // n).astype(np.int32)).to(cuda), torch.from_numpy((keys % n).astype(np.int32)).to(cuda)))
    G = NaiveGraph(snaps, n, device=cuda, sort_inplace=False, resident=False, max_cached=8)
    for t in range(6):
        G.csr("fwd", t)                                                      # first builds: validated
    G.reset_graph()
    G._snapshots.clear()
    G._edges[4][1][11] = -2                                                  # snapshot 4: an endpoint out of range, in place
    assert G.prebuild(range(6)) == 6
    with pytest.raises(ValueError, match=r"build id 5 = timestamp 4"):
        G.verify_builds()
    assert kernels._build_status_word(cuda).item() == 0                      # reported once, then clean
    G._edges[4][1][11] = 2
    G._snapshots.clear()
    assert G.prebuild(range(6)) == 6
    G.verify_builds()


@pytest.mark.parametrize("lds_count", [0, 1, 2])
@pytest.mark.parametrize("n,e", [(1, 1), (64, 400), (2708, 10556), (25_000, 250_000), (40_960, 300_000), (70_000, 300_000)])
def test_fused_rebuild_equals_the_first_build(cuda, n, e, lds_count):
    """(``lds_count``: the histogram pass with its counters in LDS -- 0 as the library chooses, 1 wherever |V| fits, 2 never.)
    stg_graph_build_direct2_device (five launches, one atomic pass, norm + per-edge norm on the side; what a
    NaiveGraph(resident=False) re-runs per snapshot and epoch) against stg_graph_build_direct_device on the same edges: every
    CSR array bit for bit, norm == degree_norm, the per-edge gathers == norm[col]; the shared counters are zero afterwards,
    so the build can be repeated (also from a HIP graph)."""
    from stgraph_amd import _C, kernels
    src, dst = random_graph(n + e, n, e, hub=e < 16000)        # (a hub of e / 8 edges: rows past 2048 go to the sort-based build)
    s, d = torch.from_numpy(src).to(cuda), torch.from_numpy(dst).to(cuda)
    first = kernels.build_graph_csr(s, d, n, cuda, lazy_node_ids=True)
    assert first.built_by == "direct"
    _C.set_tuning("build_lds_count", lds_count)
    try:
        _fused_rebuild_checks(cuda, kernels, first, s, d, n)
    finally:
        _C.set_tuning("build_lds_count", 0)


def _fused_rebuild_checks(cuda, kernels, first, s, d, n):
    for rep in range(3):
        again = kernels.build_graph_csr(s, d, n, cuda, lazy_node_ids=True, known_path="direct")
        assert again.norm_in is not None
        for side in ("fwd", "bwd"):
            a, b = getattr(first, side), getattr(again, side)
            for k in ("row_offset", "column_indices", "eids"):
                assert torch.equal(getattr(a, k), getattr(b, k)), (side, k, rep)
            nc = b._edge_cache["norm"][2]
            assert torch.equal(nc, again.norm_in.view(-1)[b.column_indices.long()]), side
        assert torch.equal(first.in_degrees, again.in_degrees) and torch.equal(first.out_degrees, again.out_degrees)
        assert torch.equal(first.perm_fwd, again.perm_fwd)
        assert torch.equal(again.norm_in.view(-1), kernels.degree_norm(degrees=first.in_degrees).view(-1))
        counters, sticky = kernels._build_counters(cuda, n)
        assert int(counters.abs().sum()) == 0 and int(sticky) == 0
    # replayed from a HIP graph
    side_stream = torch.cuda.Stream(device=cuda)
    side_stream.wait_stream(torch.cuda.current_stream(cuda))
    with torch.cuda.stream(side_stream):
        kernels.build_graph_csr(s, d, n, cuda, lazy_node_ids=True, known_path="direct")
    torch.cuda.current_stream(cuda).wait_stream(side_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap = kernels.build_graph_csr(s, d, n, cuda, lazy_node_ids=True, known_path="direct")
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(cap.fwd.column_indices, first.fwd.column_indices) and torch.equal(cap.bwd.eids, first.bwd.eids)
    assert int(kernels._build_counters(cuda, n)[0].abs().sum()) == 0


def _same_graph(a, b, tag):
    for side in ("fwd", "bwd"):
        x, y = getattr(a, side), getattr(b, side)
        for k in ("row_offset", "column_indices", "eids"):
            assert torch.equal(getattr(x, k), getattr(y, k)), (tag, side, k)
        assert torch.equal(x._edge_cache["norm"][2], y._edge_cache["norm"][2]), (tag, side, "norm_col")
    assert torch.equal(a.in_degrees, b.in_degrees) and torch.equal(a.out_degrees, b.out_degrees), tag
    assert torch.equal(a.perm_fwd, b.perm_fwd) and torch.equal(a.norm_in, b.norm_in), tag


@pytest.mark.parametrize("lds_count", [0, 1, 2])
@pytest.mark.parametrize("n,sizes", [(64, [400, 1, 90]), (2708, [10556, 9000, 12000, 300]),
                                     (25_000, [250_000, 240_000, 40_000, 260_000, 250_001, 249_999, 8, 250_000]),
                                     (40_960, [300_000] * 2 + [90_000]), (70_000, [300_000, 280_000]),
                                     (5_000, [60_000 + 17 * i for i in range(16)])])
def test_a_window_of_rebuilds_in_the_launches_of_one(cuda, n, sizes, lds_count):
    """stg_graph_build_direct2_batch_device: the snapshots of a window (different |E|, some counted in LDS and some not in
    the same batch) against one stg_graph_build_direct2_device call each -- every array bit for bit; counters of every
    slot zero afterwards; replayable from a HIP graph."""
    from stgraph_amd import _C, kernels
    lists = []
    for i, e in enumerate(sizes):
        src, dst = random_graph(n + e + i, n, e, hub=False)
        lists.append((torch.from_numpy(src).to(cuda), torch.from_numpy(dst).to(cuda)))
        assert kernels.build_graph_csr(*lists[-1], n, cuda, lazy_node_ids=True).built_by == "direct"      # validated
    _C.set_tuning("build_lds_count", lds_count)
    try:
        singles = [kernels.build_graph_csr(s, d, n, cuda, lazy_node_ids=True, known_path="direct") for s, d in lists]
        for rep in range(2):
            batch = kernels.build_graph_csr_batch(lists, n, cuda)
            assert len(batch) == len(lists)
            for i, (a, b) in enumerate(zip(singles, batch)):
                _same_graph(a, b, (rep, i))
            for slot in range(len(lists)):
                counters, sticky = kernels._build_counters(cuda, n, slot)
                assert int(counters.abs().sum()) == 0 and int(sticky) == 0
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cap = kernels.build_graph_csr_batch(lists, n, cuda)
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(singles, cap)):
            _same_graph(a, b, ("graph", i))
    finally:
        _C.set_tuning("build_lds_count", 0)


def test_batched_rebuild_argument_checks(cuda):
    from stgraph_amd import _C, kernels
    n = 100
    src, dst = random_graph(3, n, 500, hub=False)
    s, d = torch.from_numpy(src).to(cuda), torch.from_numpy(dst).to(cuda)
    with pytest.raises(ValueError):
        kernels.build_graph_csr_batch([(s, d)] * (_C.BUILD_BATCH_MAX + 1), n, cuda)
    with pytest.raises(ValueError):
        kernels.build_graph_csr_batch([], n, cuda)
    with pytest.raises(ValueError):
        kernels.build_graph_csr_batch([(s, d), (s[:0], d[:0])], n, cuda)
    sticky = torch.zeros(1, dtype=torch.int32, device=cuda)
    jobs = (_C.BuildJob * 2)()
    assert _C.lib.stg_graph_build_direct2_batch_device(jobs, _C.BUILD_BATCH_MAX + 1, n, sticky.data_ptr(), None) == _C.STG_ERR_UNSUPPORTED
    assert _C.lib.stg_graph_build_direct2_batch_device(jobs, 2, n, sticky.data_ptr(), None) == _C.STG_ERR_INVALID_ARGUMENT    # NULL arrays
    assert _C.lib.stg_graph_build_direct2_batch_device(jobs, 0, n, sticky.data_ptr(), None) == 0
    # a corrupted list inside a batch reports through the sticky word, as a single rebuild does
    good = kernels.build_graph_csr(s, d, n, cuda, lazy_node_ids=True)
    bad = s.clone()
    bad[5] = n + 1
    kernels.build_graph_csr_batch([(s, d), (bad, d)], n, cuda)
    word = kernels._build_counters(cuda, n, 0)[1]
    assert int(word) != 0
    word.zero_()
    for slot in range(2):
        kernels._build_counters(cuda, n, slot)[0].zero_()
    again = kernels.build_graph_csr_batch([(s, d), (s, d)], n, cuda)
    _same_graph(kernels.build_graph_csr(s, d, n, cuda, lazy_node_ids=True, known_path="direct"), again[1], "after")
    assert torch.equal(good.fwd.column_indices, again[0].fwd.column_indices)


def test_naive_graph_prebuilds_a_window(cuda):
    from stgraph_amd.graph import NaiveGraph
    n = 3000
    rng = np.random.default_rng(1)
    snaps = []
    for t in range(6):
        keys = rng.choice(n * n, size=20_000 + 100 * t, replace=False)
        snaps.append((torch.from_numpy((keys // n).astype(np.int32)).to(cuda), torch.from_numpy((keys % n).astype(np.int32)).to(cuda)))
    G = NaiveGraph(snaps, n, device=cuda, sort_inplace=False, resident=False, max_cached=5)
    first = [(G.csr("fwd", t).column_indices.clone(), G.csr("bwd", t).eids.clone()) for t in range(6)]
    G._snapshots.clear()
    builds = G.build_count
    assert G.prebuild(range(0, 4)) == 4 and G.build_count == builds + 4
    assert G.prebuild(range(0, 4)) == 0                                  # all cached
    for t in range(4):
        assert torch.equal(G.csr("fwd", t).column_indices, first[t][0]) and torch.equal(G.csr("bwd", t).eids, first[t][1])
    assert G.build_count == builds + 4                                   # served from the batch
    G._snapshots.clear()
    assert G.prebuild(range(0, 6)) == 5                                  # max_cached bounds a batch
    G.reset_graph()                                                      # deferred status words: clean


def test_build_counter_buffers_are_bounded_and_pinnable(cuda):
    """kernels._build_counters: buffers of at most BUILD_COUNTER_SIZES_KEPT distinct |V| stay cached unless an owner of captured graphs
    pinned theirs (raw pointers inside HIP graphs); a dead owner unpins."""
    import gc
    from stgraph_amd import kernels

    class Owner:
        pass
    kernels._BUILD_COUNTERS.clear()
    o = Owner()
    kernels.pin_build_counters(o, cuda, 1001)
    pinned = kernels._build_counters(cuda, 1001, 0)[0]
    for n in range(2000, 2000 + 3 * kernels.BUILD_COUNTER_SIZES_KEPT):
        c, st = kernels._build_counters(cuda, n, 0)
        assert c.numel() >= 2 * n and int(c.abs().sum()) == 0
    sizes = {(k[0], k[1]) for k in kernels._BUILD_COUNTERS}
    assert (str(cuda), 1001) in sizes and len(sizes) <= kernels.BUILD_COUNTER_SIZES_KEPT + 1
    assert kernels._build_counters(cuda, 1001, 0)[0].data_ptr() == pinned.data_ptr()
    del o
    gc.collect()
    assert (str(cuda), 1001) not in kernels._BUILD_COUNTER_PINS
    kernels._BUILD_COUNTERS.clear()
