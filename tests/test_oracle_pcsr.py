"""Pin the PCSR oracle (oracle/stg_pcsr_oracle.c) and, where /root/reference exists, check both
oracles against the reference's own compiled code (csr.so / pcsr.so through oracle/_ref).  CPU only."""
import numpy as np
import pytest

from oracle import ref_shim
from oracle import stg_oracle as orc
from oracle.stg_pcsr_oracle import OraclePCSR
from tests.util import golden

KEYS = ("row_offset", "column_indices", "eids")
needs_ref = pytest.mark.skipif(not ref_shim.available(), reason="reference binaries (csr.so/pcsr.so) not on this machine")


def _same_rows(a, b):
    """node_ids: the reference's std::sort leaves ties unspecified => same degree sequence, a permutation."""
    deg = np.diff(a["row_offset"].astype(np.int64))
    na, nb = a["node_ids"].astype(np.int64), b["node_ids"].astype(np.int64)
    return np.array_equal(deg[na], deg[nb]) and np.all(np.diff(deg[na]) <= 0) and sorted(na) == sorted(nb)


def test_pcsr_streams_golden():
    """Replay the recorded update streams: PMA internals, counters and both CSRs equal the reference's, bit for bit."""
    d = golden("pcsr_streams.npz")
    for tag in d["tags"]:
        n, steps = int(d[f"{tag}_num_nodes"]), int(d[f"{tag}_steps"])
        p = OraclePCSR(n, int(d[f"{tag}_max_edges"]))
        for s in range(steps):
            pre = f"{tag}_step{s}_"
            p.edge_update_list(d[pre + "add"], False, True)
            p.edge_update_list(d[pre + "delete"], True, True)
            p.label_edges()
            st = p.state()
            assert [st["N"], st["H"], st["logN"]] == d[pre + "dims"].tolist(), pre
            assert np.array_equal(st["items"], d[pre + "items"]) and np.array_equal(st["nodes"], d[pre + "nodes"]), pre
            ind, outd = p.degrees()
            assert np.array_equal(ind, d[pre + "in_degrees"]) and np.array_equal(outd, d[pre + "out_degrees"])
            for kind, out in (("fwd", p.build_csr()), ("bwd", p.build_reverse_csr())):
                want = {k: d[f"{pre}{kind}_{k}"] for k in KEYS + ("node_ids",)}
                for k in KEYS:
                    assert np.array_equal(out[k].astype(np.int32), want[k]), (pre, kind, k)
                assert _same_rows(out, want)


def test_pcsr_csr_is_the_static_csr_with_reversed_rows_and_one_based_eids():
    """What the PMA's output is, as a closed form: for a valid stream the arrays depend on the edge SET only."""
    d = golden("pcsr_streams.npz")
    tag = "s2"
    n, cur = int(d[f"{tag}_num_nodes"]), set()
    for s in range(int(d[f"{tag}_steps"])):
        pre = f"{tag}_step{s}_"
        cur |= {tuple(e) for e in d[pre + "add"].tolist()}
        cur -= {tuple(e) for e in d[pre + "delete"].tolist()}
        src = np.array([a for a, _ in cur], np.int32)
        dst = np.array([b for _, b in cur], np.int32)
        g = orc.build_graph(src, dst, n)
        for kind, c in (("fwd", g.fwd), ("bwd", g.bwd)):
            ro = c.row_offset
            assert np.array_equal(ro, d[f"{pre}{kind}_row_offset"])
            col, eid = c.column_indices.copy(), c.eids.copy()
            for r in range(n):
                col[ro[r]:ro[r + 1]] = col[ro[r]:ro[r + 1]][::-1]
                eid[ro[r]:ro[r + 1]] = eid[ro[r]:ro[r + 1]][::-1]
            assert np.array_equal(col, d[f"{pre}{kind}_column_indices"])
            assert np.array_equal(eid + 1, d[f"{pre}{kind}_eids"])


def _pcsr_csr(arrs, n):
    """OracleCSR over recorded PCSR arrays; tpl_fa_pcsr.jinja:32-34 subtracts 1 from every eid."""
    z = np.zeros(n, np.int32)
    return orc.OracleCSR(arrs["row_offset"].astype(np.int32), arrs["column_indices"].astype(np.int32),
                         arrs["eids"].astype(np.int32) - 1, arrs["node_ids"].astype(np.int32), z, z, z.astype(np.float32))


def test_pcsr_gcn_golden():
    """The emitted 'pcsr' kernels (reference code generator, tpl_fa_pcsr) == oracle aggregation over the PCSR arrays."""
    d = golden("pcsr_gcn.npz")
    n = int(d["num_nodes"])
    fwd = _pcsr_csr({k: d[f"fwd_{k}"] for k in KEYS + ("node_ids",)}, n)
    bwd = _pcsr_csr({k: d[f"bwd_{k}"] for k in KEYS + ("node_ids",)}, n)
    # the store was fed this edge list: same arrays from the oracle's PMA
    p = OraclePCSR(n, len(d["src"]))
    p.edge_update_list(sorted(zip(d["src"].tolist(), d["dst"].tolist()), key=lambda x: (x[1], x[0])), False, True)
    p.label_edges()
    for k in KEYS:
        assert np.array_equal(p.build_csr()[k].astype(np.int32), d[f"fwd_{k}"])
        assert np.array_equal(p.build_reverse_csr()[k].astype(np.int32), d[f"bwd_{k}"])
    for F in (7, 16, 64):
        fa = orc.ref_active_columns(F)
        for use_ew in (False, True):
            tag = f"F{F}_{'ew' if use_ew else 'now'}"
            w = d["edge_weight_by_eid"] if use_ew else None
            out = orc.gcn_agg(d[tag + "_x"], d["norm"], d["norm"], fwd, ew=w, use_node_ids=True, f_active=fa)
            gx = orc.gcn_agg(d[tag + "_R"], d["norm"], d["norm"], bwd, ew=w, use_node_ids=True, f_active=fa)
            assert np.array_equal(out, d[tag + "_out"]), tag
            assert np.array_equal(gx, d[tag + "_grad_x"]), tag


def test_pcsr_tgcn_golden_arrays():
    """PCSRGraph protocol (pcsr_graph.py:46-166) replayed on the oracle: forward arrays at every get_graph(t),
    reverse arrays at every backward step."""
    d = golden("pcsr_tgcn.npz")
    n, T, B = int(d["num_nodes"]), int(d["T"]), int(d["B"])
    sets = [set(zip(d[f"t{t}_src"].tolist(), d[f"t{t}_dst"].tolist())) for t in range(T)]
    key = lambda x: (x[1], x[0])  # noqa: E731
    p = OraclePCSR(n, int(d["max_num_edges"]))
    p.edge_update_list(sorted(sets[0], key=key), False, True)
    p.label_edges()
    cur = 0

    def move(to):
        nonlocal cur
        while cur != to:
            nxt = cur + 1 if to > cur else cur - 1
            hi = max(cur, nxt)
            add, dele = sorted(sets[hi] - sets[hi - 1], key=key), sorted(sets[hi - 1] - sets[hi], key=key)
            if nxt < cur:
                add, dele = dele, add
            p.edge_update_list(add, False, True)
            p.edge_update_list(dele, True, True)
            p.label_edges()
            cur = nxt

    for w0 in range(0, T, B):
        ts = list(range(w0, min(w0 + B, T)))
        for t in ts:
            move(t)
            out = p.build_csr()
            for k in KEYS:
                assert np.array_equal(out[k].astype(np.int32), d[f"t{t}_fwd_{k}"]), (t, k)
            assert np.array_equal(p.out_degrees.astype(np.int32), d[f"t{t}_in_degrees"])   # pcsr_graph.py:101-103
            assert p.edge_count == int(d[f"t{t}_num_edges"])
        for t in reversed(ts):
            move(t)
            out = p.build_reverse_csr()
            for k in KEYS:
                assert np.array_equal(out[k].astype(np.int32), d[f"t{t}_bwd_{k}"]), (t, k)
        move(ts[-1])


@needs_ref
def test_csr_oracle_equals_reference_binary():
    """orc_csr_ctor vs the reference's compiled CSR::CSR (csr.so) on random multigraphs."""
    rng = np.random.default_rng(0)
    for _ in range(120):
        n, e = int(rng.integers(1, 60)), int(rng.integers(1, 400))
        src, dst = rng.integers(0, n, e).astype(np.int32), rng.integers(0, n, e).astype(np.int32)
        w = rng.random(e).astype(np.float32)
        _, f, b = orc.prepare_edge_lists(src, dst)
        for (x, y, eid), rev in ((f, True), (b, False)):
            r, o = ref_shim.csr_ctor(x, y, eid, w, n, rev), orc.csr_ctor(x, y, eid, w, n, rev)
            for k in KEYS + ("in_degrees", "out_degrees", "weighted_out_degrees"):
                assert np.array_equal(r[k], getattr(o, k)), k
            assert _same_rows(r, {"node_ids": o.node_ids})


@needs_ref
def test_pcsr_oracle_equals_reference_binary_state_for_state():
    """Random update streams (valid ones, and ones that re-add present / delete absent edges): after every step
    the PMA internals and counters of the restatement equal those of the reference's compiled PCSR (pcsr.so)."""
    rng = np.random.default_rng(7)
    for trial in range(60):
        n = int(rng.integers(2, 50))
        uni = [(int(a), int(b)) for a in range(n) for b in range(n)]
        rng.shuffle(uni)
        uni = uni[: int(rng.integers(1, min(len(uni), 400) + 1))]
        r, o = ref_shim.RefPCSR(n, len(uni)), OraclePCSR(n, len(uni))
        cur, rev, valid = set(), bool(rng.integers(0, 2)), bool(rng.integers(0, 4))
        for step in range(6):
            cand = [e for e in uni if e not in cur] if valid else list(uni)
            add = [cand[i] for i in rng.permutation(len(cand))[: int(rng.integers(0, len(cand) + 1))]]
            cl = sorted(cur) if valid else [e for e in uni if e in cur or rng.integers(0, 8) == 0]
            dele = [cl[i] for i in rng.permutation(len(cl))[: int(rng.integers(0, len(cl) + 1))]] if step else []
            for obj in (r, o):
                obj.edge_update_list(add, False, rev)
                obj.edge_update_list(dele, True, rev)
                obj.label_edges()
            cur |= set(add)
            cur -= set(dele)
            sr, so = r.state(), o.state()
            assert (sr["N"], sr["H"], sr["logN"]) == (so["N"], so["H"], so["logN"])
            assert np.array_equal(sr["items"], so["items"]) and np.array_equal(sr["nodes"], so["nodes"])
            assert r.edge_count == o.edge_count
            assert all(np.array_equal(x, y) for x, y in zip(r.degrees(), o.degrees()))
            consistent = valid and r.edge_count == len(cur) and len(o.get_edges()) == len(cur)
            if consistent:                 # (the reference can lose an edge of the last vertex: DESIGN.md D13)
                for m in ("build_csr", "build_reverse_csr"):
                    x, y = getattr(r, m)(), getattr(o, m)()
                    for k in KEYS:
                        assert np.array_equal(x[k], y[k]), (m, k)
                    assert _same_rows(x, y)
            if not valid and any(int(v) > 1 << 30 for v in r.degrees()[0]):
                break                       # counters wrapped below zero: nothing meaningful left to compare
