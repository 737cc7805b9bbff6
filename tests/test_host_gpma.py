"""GPMA counterpart on host arrays (no GPU): the ``gpma`` module functions and ``GPMAGraph`` against the
oracle's restatement of the reference's label / reverse-CSR / kernel contract on a gapped array
(oracle/stg_gpma_oracle.c; PARITY UNPINNED: the reference ships no gpma.so) and against ``NaiveGraph``."""
import copy

import numpy as np
import pytest
import torch

from oracle import stg_oracle as orc
from oracle.stg_gpma_oracle import OracleGPMA
from tests.util import gcn_norm, random_graph


def snapshots(seed, n, e0, churn, T):
    """T duplicate-free snapshots as lists of (src, dst): a sliding window over a random edge stream."""
    rng = np.random.default_rng(seed)
    stream = rng.choice(n * n, size=e0 + churn * T, replace=False)
    out = []
    for t in range(T):
        k = stream[t * churn: t * churn + e0]
        out.append([(int(a), int(b)) for a, b in zip(k // n, k % n)])
    return out


def updates_of(snaps):
    """``DynamicGraph.graph_updates`` format (dynamic_graph.py:56-79)."""
    key = lambda x: (x[1], x[0])  # noqa: E731
    upd = {"0": {"add": sorted(snaps[0], key=key), "delete": []}}
    for t in range(1, len(snaps)):
        a, b = set(snaps[t]), set(snaps[t - 1])
        upd[str(t)] = {"add": sorted(a - b, key=key), "delete": sorted(b - a, key=key)}
    return upd


def check_module_against_oracle(device):
    from stgraph_amd.graph.dynamic.gpma import gpma as M
    from stgraph_amd.graph.static.csr import get_array
    n, T = 40, 5
    snaps = snapshots(5, n, 300, 40, T)
    g = M.GPMA(device=device)
    M.init_gpma(g, n)
    M.init_graph_updates(g, updates_of(snaps), reverse_edges=True)
    cur = -1                                                             # forward, then revert back to t = 0
    plan = [("fwd", t) for t in range(T)] + [("rev", t) for t in range(T - 1, 0, -1)]
    for kind, t in plan:
        times = M.edge_update_t(g, t, revert_update=(kind == "rev"))
        assert len(times) == 2
        M.label_edges(g)
        cur = t if kind == "fwd" else t - 1
        edges = snaps[cur]
        # the reference's pipeline on a gapped image of the same set; rows = destinations (reverse_edges=True),
        # a few tombstones: the edges the NEXT snapshot no longer has... any absent pair will do
        dead = [(d, s) for s, d in snaps[(cur + 2) % T] if (s, d) not in set(edges)][:25]
        o = OracleGPMA(n, [(d, s) for s, d in edges], dead=dead, hole_pct=35, seed=cur + 1)
        o.label_edges()
        assert M.get_graph_attr(g) == (n, len(edges)) and g.edge_count == o.edge_count
        assert M.get_gpma_edge_list(g) == set(o.live_edges())
        assert M.get_out_degrees(g) == o.out_degree.tolist() and M.get_in_degrees(g) == o.in_degree.tolist()
        # forward pointers: uint32 offsets, uint64 keys, 1-based labels -- the dense image of the same rows
        ro_p, col_p, eid_p, nid_p = M.get_csr_ptrs(g)
        E = len(edges)
        ro, keys, lab = get_array(ro_p, n + 1), get_array(col_p, E), get_array(eid_p, E)
        assert lab == list(range(1, E + 1))
        assert [(k >> 32, k & 0xFFFFFFFF, l) for k, l in zip(keys, lab)] == o.live_edges()
        assert ro == np.concatenate([[0], np.cumsum(o.out_degree)]).tolist()
        nid = np.array(get_array(nid_p, n))
        assert sorted(nid.tolist()) == list(range(n)) and np.array_equal(o.out_degree[nid], o.out_degree[o.node_ids()])
        # reverse CSR: same offsets, same (row, col, label) content per row; in-row order is undefined upstream
        with pytest.raises(RuntimeError):
            M.free_backward_csr(g)
            M.get_csr_ptrs(g, is_backward=True)
        assert len(M.build_backward_csr(g)) == 3
        want = o.build_backward_csr()
        bro_p, bcol_p, beid_p, bnid_p = M.get_csr_ptrs(g, is_backward=True)
        bro, bkeys, blab = get_array(bro_p, n + 1), get_array(bcol_p, E), get_array(beid_p, E)
        assert bro == want["row_offset"].tolist()
        for r in range(n):
            a, b = bro[r], bro[r + 1]
            assert sorted(zip(bkeys[a:b], blab[a:b])) == sorted(zip(want["keys"][a:b].tolist(), want["values"][a:b].tolist()))
            assert bkeys[a:b] == sorted(bkeys[a:b])                                   # ascending here
        assert M.get_reverse_csr_edge_list(g) == {(int(k >> 32), int(k & 0xFFFFFFFF), int(l))
                                                  for k, l in zip(want["keys"].tolist(), want["values"].tolist())}
        bn = np.array(get_array(bnid_p, n))
        assert np.array_equal(o.in_degree[bn], o.in_degree[o.node_ids(backward=True)])
        M.free_backward_csr(g)
    g.check()
    c = copy.deepcopy(g)
    assert M.get_gpma_edge_list(c) == M.get_gpma_edge_list(g) and isinstance(c, M.GPMA)


def test_gpma_module_host():
    check_module_against_oracle("cpu")


def check_emit_equals_static_builder(device, n, e, seed=3):
    """Key-order emission of an edge set == the static builder's CSR of the same edges (both directions):
    what makes a GPMAGraph snapshot interchangeable with a NaiveGraph snapshot."""
    from stgraph_amd import kernels
    src, dst = random_graph(seed, n, e)
    s, d = torch.from_numpy(src).to(device), torch.from_numpy(dst).to(device)
    es = kernels.edgeset_update(kernels.edgeset_empty(n, device), s, d)
    kernels.edgeset_check(es)
    g = kernels.build_graph_csr(s, d, n, device)
    for rev, side in ((False, g.fwd), (True, g.bwd)):
        c = kernels.edgeset_emit_csr(es, rev, key_order=True)
        assert torch.equal(c.row_offset, side.row_offset) and torch.equal(c.column_indices, side.column_indices)
        assert torch.equal(c.eids, side.eids) and torch.equal(c.eids1, side.eids + 1)
        assert torch.equal(c.keys >> 32, torch.repeat_interleave(
            torch.arange(n, device=c.keys.device), (side.row_offset[1:] - side.row_offset[:-1]).long()))
        assert torch.equal((c.keys & 0xFFFFFFFF).int(), side.column_indices)
        deg = c.degrees[c.node_ids.long()]
        assert bool((deg[1:] <= deg[:-1]).all())
    with pytest.raises(ValueError):
        kernels.edgeset_emit_csr(es, False).keys          # the PCSR layout is not in key order


@pytest.mark.parametrize("n,e", [(1, 1), (5, 12), (64, 400), (2708, 10556)])
def test_key_order_emission_equals_static_csr_host(n, e):
    check_emit_equals_static_builder("cpu", n, e)


def check_graph_protocol(device):
    """GPMAGraph driven like the training loop (windows of B, BPTT walk back, two epochs): every timestamp
    publishes the CSR a NaiveGraph holds for that snapshot."""
    from stgraph_amd.graph import DynamicGraph, GPMAGraph, NaiveGraph
    from stgraph_amd.graph.static.csr import get_array
    n, T, B = 30, 7, 3
    snaps = snapshots(9, n, 200, 30, T)
    G = GPMAGraph([list(s) for s in snaps], n, device=device)
    R = NaiveGraph([list(s) for s in snaps], n, device=device)
    assert isinstance(G, DynamicGraph) and G.graph_type() == "gpma"

    def same(direction, t):
        a, b = G.csr(direction), R.csr(direction, t)
        for k in ("row_offset", "column_indices", "eids"):
            assert torch.equal(getattr(a, k), getattr(b, k)), (direction, t, k)
        deg = (a.row_offset[1:] - a.row_offset[:-1])[a.node_ids.long()]
        assert bool((deg[1:] <= deg[:-1]).all())

    for epoch in range(2):
        G.reset_graph()
        for w0 in range(0, T, B):
            ts = list(range(w0, min(w0 + B, T)))
            G.get_graph(w0)
            for t in ts:
                G.get_graph(t)
                assert G.current_timestamp == t and G.get_num_edges() == len(snaps[t]) and G.get_num_nodes() == n
                same("fwd", t)
                indeg = np.bincount([d for _, d in snaps[t]], minlength=n)
                assert np.array_equal(G.in_degrees(), indeg) and G.in_degrees().dtype == np.int32
                assert np.array_equal(G.out_degrees(), np.bincount([s for s, _ in snaps[t]], minlength=n))
                E = len(snaps[t])
                keys = get_array(G.fwd_column_indices_ptr, E)
                assert keys == sorted((d << 32) | s for s, d in snaps[t])
                assert get_array(G.fwd_eids_ptr, E) == list(range(1, E + 1))
                G.set_ndata("norm", torch.full((n, 1), float(t)))
            for t in reversed(ts):
                G.get_backward_graph(t)
                assert G.current_timestamp == t
                same("bwd", t)
                assert float(G.get_ndata("norm")[0, 0]) == float(t)
                bk = get_array(G.bwd_column_indices_ptr, len(snaps[t]))
                assert bk == sorted((s << 32) | d for s, d in snaps[t])
    G.check()
    with pytest.raises(RuntimeError):
        G.get_graph(T)                       # past the last timestamp (gpma_graph.py:121-124)


def test_gpma_graph_protocol_host():
    check_graph_protocol("cpu")


def test_oracle_kernel_contract_on_gapped_array():
    """The reference's GPMA kernel loop over a gapped array (holes, walls, tombstones) with the intended
    predicate == the plain CSR loop over the dense arrays this build emits, bit for bit (same in-row order);
    the template's literal predicate (defect D17) is not: it drops the edge labelled 1 and counts tombstones."""
    from stgraph_amd import kernels
    n, F = 50, 12
    src, dst = random_graph(21, n, 600)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((n, F)).astype(np.float32)
    w = (rng.random(600) + 0.5).astype(np.float32)
    live = set(zip(dst.tolist(), src.tolist()))
    dead = [(int(a), int(b)) for a, b in zip(rng.integers(0, n, 80), rng.integers(0, n, 80)) if (int(a), int(b)) not in live]
    o = OracleGPMA(n, list(live), dead=dead, hole_pct=40, seed=7)
    o.label_edges()
    o.build_backward_csr()
    es = kernels.edgeset_update(kernels.edgeset_empty(n, "cpu"), torch.from_numpy(src), torch.from_numpy(dst))
    fwd, bwd = (kernels.edgeset_emit_csr(es, r, key_order=True) for r in (False, True))
    norm = gcn_norm(o.out_degree.astype(np.int64))
    as_orc = lambda c: orc.OracleCSR(*(getattr(c, k).numpy() for k in ("row_offset", "column_indices", "eids", "node_ids")),  # noqa: E731
                                     None, None, None)
    for ew in (None, w):
        want = orc.gcn_agg(x, norm, norm, as_orc(fwd), ew=ew, use_node_ids=True)
        assert np.array_equal(o.gcn_agg(x, norm, norm, ew), want)
        back = orc.gcn_agg(x, norm, norm, as_orc(bwd), ew=ew, use_node_ids=True)
        np.testing.assert_allclose(o.gcn_agg(x, norm, norm, ew, backward=True), back, rtol=1e-5, atol=1e-5)
    lit = o.gcn_agg(x, norm, norm, None, literal=True)
    assert not np.array_equal(lit, orc.gcn_agg(x, norm, norm, as_orc(fwd), use_node_ids=True))
