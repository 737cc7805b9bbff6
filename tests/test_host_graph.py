"""Host-side logic (no GPU): host CSR builders, StaticGraph / NaiveGraph / DynamicGraph protocol."""
import copy

import numpy as np
import pytest
import torch

from oracle import stg_oracle as orc
from tests.util import golden, random_graph

CSR_KEYS = ("row_offset", "column_indices", "eids", "node_ids")


def _eq(csr, ocsr):
    for k in CSR_KEYS:
        assert np.array_equal(getattr(csr, k).cpu().numpy(), getattr(ocsr, k)), k


@pytest.mark.parametrize("n,e,dup", [(1, 0, False), (1, 1, True), (9, 0, False), (40, 300, True),
                                     (500, 6000, False), (20000, 150000, True)])
def test_graph_build_host_matches_oracle(n, e, dup):
    from stgraph_amd import kernels
    if e == 0:
        src = dst = np.empty(0, np.int32)
    elif n == 1:
        src = dst = np.zeros(e, np.int32)
    else:
        src, dst = random_graph(n * 7 + e, n, e, duplicates=dup, hub=e > 100)
    g = kernels.build_graph_csr(src, dst, n, "cpu")
    og = orc.build_graph(src, dst, n)
    _eq(g.fwd, og.fwd)
    _eq(g.bwd, og.bwd)
    assert np.array_equal(g.in_degrees.numpy(), og.in_degrees())
    assert np.array_equal(g.out_degrees.numpy(), og.out_degrees())
    assert np.array_equal(g.perm_fwd.numpy(), og.perm_fwd)


@pytest.mark.parametrize("tag", ["n1", "n5", "n64", "n2708"])
def test_static_graph_matches_reference_golden(tag):
    from stgraph_amd.graph import StaticGraph
    d = golden(f"csr_{tag}.npz")
    n = int(d["num_nodes"])
    el = [(int(a), int(b)) for a, b in zip(d["src"], d["dst"])]
    g = StaticGraph(el, d["weights_by_eid"].tolist(), n, device="cpu")
    for side in ("fwd", "bwd"):
        for k in CSR_KEYS[:3]:
            assert np.array_equal(getattr(g.csr(side), k).numpy(), d[f"{side}_{k}"]), (side, k)
    assert np.array_equal(g.in_degrees(), d["in_degrees"]) and g.in_degrees().dtype == np.int32
    assert np.array_equal(g.out_degrees(), d["out_degrees"])
    assert np.array_equal(g.weighted_in_degrees(), d["weighted_in_degrees"])
    assert g.get_num_nodes() == n and g.get_num_edges() == int(d["num_edges"])
    assert g.graph_type() == "csr_unsorted"
    # the reference sorts the CALLER's list in place by (dst, src)  (static_graph.py:66-67)
    assert np.array_equal(np.array(el, np.int32).reshape(-1, 2), d["sorted_inplace"].reshape(-1, 2))
    # raw-pointer surface of STGraphBase + get_array
    from stgraph_amd.graph.static.csr import get_array
    assert get_array(g.fwd_row_offset_ptr, n + 1) == d["fwd_row_offset"].tolist()
    assert get_array(g.bwd_eids_ptr, len(el)) == d["bwd_eids"].tolist()
    g.set_ndata("norm", torch.ones(n, 1))
    assert g.get_ndata("norm").shape == (n, 1) and g.get_ndata("missing") is None


def test_static_graph_accepts_arrays_and_counts_distinct_edges():
    from stgraph_amd.graph import StaticGraph
    src = np.array([0, 0, 1, 2, 0], np.int32)
    dst = np.array([1, 1, 2, 0, 1], np.int32)
    for edges in (np.stack([src, dst], 1), (src, dst), torch.from_numpy(np.stack([src, dst], 1))):
        g = StaticGraph(edges, None, 3, device="cpu", sort_inplace=False)
        assert g.get_num_edges() == 3                      # len(set(edge_list)), static_graph.py:48
        assert g.csr("fwd").num_edges == 5                 # the CSR keeps duplicates
        og = orc.build_graph(src, dst, 3)
        _eq(g.csr("fwd"), og.fwd)
        _eq(g.csr("bwd"), og.bwd)


def test_vertex_out_of_range_and_bad_shapes_raise():
    from stgraph_amd import _C, kernels
    from stgraph_amd.graph import StaticGraph
    with pytest.raises(_C.StgError) as ei:
        kernels.build_graph_csr(np.array([0, 3], np.int32), np.array([1, 1], np.int32), 3, "cpu")
    assert ei.value.code == _C.STG_ERR_VERTEX_RANGE
    with pytest.raises(ValueError):
        StaticGraph(np.zeros((3,), np.int32), None, 5, device="cpu")
    with pytest.raises(ValueError):
        StaticGraph([(-1, 0)], None, 5, device="cpu")


def test_csr_class_matches_pybind_surface():
    from stgraph_amd.graph.static.csr import CSR, get_array
    fwd = [(1, 0, 0), (0, 1, 1), (2, 1, 2), (0, 2, 3), (3, 2, 4)]
    c = CSR(fwd, [1.0, 2.0, 3.0, 4.0, 5.0], 4, is_edge_reverse=True, device="cpu")
    o = orc.csr_ctor([t[0] for t in fwd], [t[1] for t in fwd], [t[2] for t in fwd], [1, 2, 3, 4, 5], 4, True)
    assert get_array(c.row_offset_ptr, 5) == o.row_offset.tolist()
    assert get_array(c.column_indices_ptr, 5) == o.column_indices.tolist()
    assert get_array(c.eids_ptr, 5) == o.eids.tolist()
    assert c.out_degrees == o.out_degrees.tolist() and c.in_degrees == o.in_degrees.tolist()
    assert c.weighted_out_degrees == o.weighted_out_degrees.tolist()
    c2 = copy.deepcopy(c)
    assert c2.row_offset_ptr == c.row_offset_ptr          # copies share the device arrays (csr.cu:193-199)
    from stgraph_amd import _C
    with pytest.raises(_C.StgError):                      # rows must arrive grouped, as the Python callers guarantee
        CSR([(0, 2, 0), (0, 1, 1)], [1.0, 1.0], 3, is_edge_reverse=True, device="cpu")


def _naive_fixture():
    d = golden("naive_tgcn.npz")
    n, T = int(d["num_nodes"]), int(d["T"])
    snaps = [[(int(a), int(b)) for a, b in zip(d[f"t{t}_src"], d[f"t{t}_dst"])] for t in range(T)]
    return d, n, T, snaps


@pytest.mark.parametrize("resident", [True, False])
def test_naive_graph_snapshots_and_timestamp_protocol(resident):
    from stgraph_amd.graph import DynamicGraph, NaiveGraph
    d, n, T, snaps = _naive_fixture()
    G = NaiveGraph(snaps, n, device="cpu", resident=resident, max_cached=None if resident else 2)
    assert isinstance(G, DynamicGraph) and G.graph_type() == "csr"
    assert G.build_count == (T if resident else 1)
    for t in range(T):
        for side in ("fwd", "bwd"):
            for k in CSR_KEYS[:3]:
                assert np.array_equal(getattr(G.csr(side, t), k).numpy(), d[f"t{t}_{side}_{k}"]), (t, side, k)
            nid = G.csr(side, t).node_ids.numpy()
            deg = np.diff(G.csr(side, t).row_offset.numpy())
            assert sorted(nid.tolist()) == list(range(n)) and np.all(np.diff(deg[nid]) <= 0)
    G.reset_graph()
    assert G.current_timestamp == 0
    seen = []
    for t in range(T):
        G.get_graph(t)
        assert G.current_timestamp == t and G.get_num_nodes() == n
        assert G.get_num_edges() == len(set(snaps[t]))
        assert G.fwd_row_offset_ptr == G.csr("fwd", t).row_offset_ptr
        G.set_ndata("norm", torch.full((n, 1), float(t)))
        seen.append(G.in_degrees().copy())
        assert np.array_equal(G.in_degrees(), np.diff(d[f"t{t}_fwd_row_offset"]))
    with pytest.raises(RuntimeError):
        G.get_graph(T - 2)                                 # forward view cannot go back in time
    for t in reversed(range(T)):                           # BPTT order
        G.get_backward_graph(t)
        assert G.current_timestamp == t
        assert G.bwd_row_offset_ptr == G.csr("bwd", t).row_offset_ptr
        assert float(G.get_ndata("norm")[0, 0]) == float(t)   # node data is per timestamp
    with pytest.raises(RuntimeError):
        G.get_backward_graph(2)                            # backward view cannot go forward
    G.get_graph(1)                                         # next window resumes forward from t=0
    assert G.current_timestamp == 1
    with pytest.raises(RuntimeError):
        G.get_graph(T)                                     # past the last snapshot
    upd = G.graph_updates                                   # add/delete lists, sorted by (dst, src)
    assert set(upd["1"]["add"]) == set(snaps[1]) - set(snaps[0])
    assert set(upd["1"]["delete"]) == set(snaps[0]) - set(snaps[1])
    assert upd["2"]["add"] == sorted(upd["2"]["add"], key=lambda x: (x[1], x[0]))


def test_host_builder_property_based():
    """hypothesis: arbitrary multigraphs (duplicates, self loops, isolated vertices, any order) --
    the host builder equals the oracle's restatement of the reference pipeline, and the CSR
    invariants the kernels rely on hold."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    from stgraph_amd import kernels

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 40).flatmap(lambda n: st.tuples(
        st.just(n), st.lists(st.tuples(st.integers(0, n - 1), st.integers(0, n - 1)), max_size=200))))
    def check(case):
        n, edges = case
        src = np.array([a for a, _ in edges], np.int32)
        dst = np.array([b for _, b in edges], np.int32)
        g = kernels.build_graph_csr(src, dst, n, "cpu")
        og = orc.build_graph(src, dst, n)
        _eq(g.fwd, og.fwd)
        _eq(g.bwd, og.bwd)
        e = len(edges)
        ro = g.fwd.row_offset.numpy()
        assert ro[0] == 0 and ro[-1] == e and np.all(np.diff(ro) >= 0)
        assert g.fwd.eids.numpy().tolist() == list(range(e))                  # forward eids are the identity
        assert sorted(g.bwd.eids.numpy().tolist()) == list(range(e))          # backward eids: a permutation
        # every backward entry points at the forward position of the same (src, dst) pair
        frows = np.repeat(np.arange(n), np.diff(ro))
        brows = np.repeat(np.arange(n), np.diff(g.bwd.row_offset.numpy()))
        be = g.bwd.eids.numpy()
        assert np.array_equal(g.fwd.column_indices.numpy()[be], brows)
        assert np.array_equal(frows[be], g.bwd.column_indices.numpy())

    check()
