"""PCSR store on host arrays (no GPU): the C ABI's *_host entry points behind ``PCSR`` / ``PCSRGraph``
against the reference-recorded fixtures and the oracle's packed-memory array."""
import copy

import numpy as np
import pytest
import torch

from oracle.stg_pcsr_oracle import OraclePCSR
from tests.util import golden

KEYS = ("row_offset", "column_indices", "eids", "node_ids")


def check_published(p, want, n, tag=""):
    """The four arrays behind ``get_csr_ptrs()`` == the reference's (node_ids: same degree sequence)."""
    from stgraph_amd.graph.static.csr import get_array
    ro_p, col_p, eid_p, nid_p = p.get_csr_ptrs()
    e = len(want["column_indices"])
    assert get_array(ro_p, n + 1) == want["row_offset"].tolist(), tag
    assert get_array(col_p, e) == want["column_indices"].tolist(), tag
    assert get_array(eid_p, e) == want["eids"].tolist(), tag
    nid = np.array(get_array(nid_p, n))
    deg = np.diff(want["row_offset"].astype(np.int64))
    assert sorted(nid.tolist()) == list(range(n)) and np.array_equal(deg[nid], deg[want["node_ids"]]), tag


def replay_streams(device):
    from stgraph_amd.graph.dynamic.pcsr.pcsr import PCSR
    d = golden("pcsr_streams.npz")
    for tag in d["tags"]:
        n = int(d[f"{tag}_num_nodes"])
        p = PCSR(n, int(d[f"{tag}_max_edges"]), device=device)
        for s in range(int(d[f"{tag}_steps"])):
            pre = f"{tag}_step{s}_"
            p.edge_update_list(d[pre + "add"], is_reverse_edge=True)
            p.edge_update_list(d[pre + "delete"], is_delete=True, is_reverse_edge=True)
            p.label_edges()
            p.check()
            for kind, build in (("fwd", p.build_csr), ("bwd", p.build_reverse_csr)):
                assert build() == 0.0
                check_published(p, {k: d[f"{pre}{kind}_{k}"] for k in KEYS}, n, pre + kind)
            assert np.array_equal(p.in_degrees, d[pre + "in_degrees"]) and np.array_equal(p.out_degrees, d[pre + "out_degrees"])
            assert p.edge_count == len(d[pre + "fwd_column_indices"]) and p.get_n() == n
            # 0-based view for the launch wrappers
            assert np.array_equal(p.csr(False).eids.cpu().numpy() + 1, d[pre + "fwd_eids"])
        assert p.update_count == sum(1 for s in range(int(d[f"{tag}_steps"]))
                                     if len(d[f"{tag}_step{s}_add"]) + len(d[f"{tag}_step{s}_delete"]))


def test_pcsr_streams_host():
    replay_streams("cpu")


def replay_protocol(device):
    """PCSRGraph protocol as the training loop drives it (windows of B, cache/restore at the boundary)."""
    from stgraph_amd.graph import DynamicGraph, PCSRGraph
    from stgraph_amd.graph.static.csr import get_array
    d = golden("pcsr_tgcn.npz")
    n, T, B = int(d["num_nodes"]), int(d["T"]), int(d["B"])
    snaps = [[(int(a), int(b)) for a, b in zip(d[f"t{t}_src"], d[f"t{t}_dst"])] for t in range(T)]
    G = PCSRGraph(snaps, n, device=device)
    assert isinstance(G, DynamicGraph) and G.graph_type() == "pcsr" and G.max_num_edges == int(d["max_num_edges"])
    for epoch in range(2):                                     # the second epoch starts from the restored base graph
        G.reset_graph()
        for w0 in range(0, T, B):
            ts = list(range(w0, min(w0 + B, T)))
            G.get_graph(w0)
            for t in ts:
                G.get_graph(t)
                assert G.current_timestamp == t and G.get_num_edges() == int(d[f"t{t}_num_edges"])
                check_published(G._forward_graph, {k: d[f"t{t}_fwd_{k}"] for k in KEYS}, n, f"fwd t{t}")
                assert get_array(G.fwd_row_offset_ptr, n + 1) == d[f"t{t}_fwd_row_offset"].tolist()
                assert np.array_equal(G.in_degrees(), d[f"t{t}_in_degrees"]) and G.in_degrees().dtype == np.int32
                assert np.array_equal(G.in_degrees_tensor().cpu().numpy(), d[f"t{t}_in_degrees"])
                f = G.csr("fwd")
                assert np.array_equal(f.eids.cpu().numpy() + 1, d[f"t{t}_fwd_eids"])
                G.set_ndata("norm", torch.full((n, 1), float(t)))
            for t in reversed(ts):                             # BPTT
                G.get_backward_graph(t)
                assert G.current_timestamp == t
                check_published(G._forward_graph, {k: d[f"t{t}_bwd_{k}"] for k in KEYS}, n, f"bwd t{t}")
                assert get_array(G.bwd_eids_ptr, len(d[f"t{t}_bwd_eids"])) == d[f"t{t}_bwd_eids"].tolist()
                assert float(G.get_ndata("norm")[0, 0]) == float(t)
                assert np.array_equal(G.csr("bwd").column_indices.cpu().numpy(), d[f"t{t}_bwd_column_indices"])
                # the forward CSR of the same timestamp stays reachable during backprop (GAT backward reads both)
                assert np.array_equal(G.csr("fwd").column_indices.cpu().numpy(), d[f"t{t}_fwd_column_indices"])
            G.check()
    with pytest.raises(RuntimeError):
        G.get_graph(T)
    with pytest.raises(RuntimeError):
        G.get_graph(0)                                         # forward view cannot go back in time
    upd = G.graph_updates                                      # the reference's Python-list view, on demand
    assert set(upd["1"]["add"]) == set(snaps[1]) - set(snaps[0])


def test_pcsr_graph_protocol_host():
    replay_protocol("cpu")


def test_pcsr_matches_oracle_pma_on_random_streams():
    from stgraph_amd.graph.dynamic.pcsr.pcsr import PCSR
    rng = np.random.default_rng(5)
    for trial in range(25):
        n = int(rng.integers(2, 60))
        uni = [(int(a), int(b)) for a in range(n) for b in range(n)]
        rng.shuffle(uni)
        uni = uni[: int(rng.integers(1, min(len(uni), 500) + 1))]
        p, o, cur = PCSR(n, len(uni), device="cpu"), OraclePCSR(n, len(uni)), set()
        for step in range(6):
            cand = [e for e in uni if e not in cur]
            add = [cand[i] for i in rng.permutation(len(cand))[: int(rng.integers(0, len(cand) + 1))]]
            cl = sorted(cur)
            dele = [cl[i] for i in rng.permutation(len(cl))[: int(rng.integers(0, len(cl) + 1))]] if step else []
            for obj in (p, o):
                obj.edge_update_list(add, False, True)
                obj.edge_update_list(dele, True, True)
                obj.label_edges()
            cur |= set(add)
            cur -= set(dele)
            assert p.edge_count == len(cur)
            if o.edge_count != len(cur) or len(o.get_edges()) != len(cur):
                break                   # the reference's PMA hid an edge of the last vertex: no oracle
            for m, rev in (("build_csr", False), ("build_reverse_csr", True)):
                want = getattr(o, m)()
                c = p.csr(rev)
                assert np.array_equal(c.row_offset.numpy(), want["row_offset"].astype(np.int32))
                assert np.array_equal(c.column_indices.numpy(), want["column_indices"].astype(np.int32))
                assert np.array_equal(p.labels(rev).numpy(), want["eids"].astype(np.int32))
            assert [tuple(e) for e in o.get_edges().tolist()] == p.get_edges()


def test_pcsr_rejects_invalid_streams_and_copies_are_independent():
    from stgraph_amd.graph.dynamic.pcsr.pcsr import PCSR
    p = PCSR(4, 8, device="cpu")
    p.edge_update_list([(0, 1), (2, 1), (1, 3)])
    p.label_edges()
    p.check()
    q = copy.deepcopy(p)
    q.edge_update_list([(3, 0)])
    assert q.edge_count == 4 and p.edge_count == 3              # the copy moved on, the original did not
    assert p.get_edges() == [(0, 1, 1), (1, 3, 2), (2, 1, 3)]
    for bad, kw in (([(0, 1)], {}), ([(3, 3)], {"is_delete": True}), ([(0, 9)], {})):
        r = copy.copy(p)
        r.edge_update_list(bad, **kw)
        with pytest.raises(ValueError):
            r.check()
    r = copy.copy(p)
    r.edge_update_list([(0, 1)], is_delete=True)
    r.edge_update_list([(0, 1)])                                # delete then re-add: two passes, valid
    r.check()
    assert r.get_edges() == p.get_edges() and r.update_count == p.update_count + 2
