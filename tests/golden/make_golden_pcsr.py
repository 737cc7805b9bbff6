"""Golden vectors for the PCSR dynamic-graph store, recorded from the reference itself.

BUILD CONTAINER ONLY.  The reference's compiled ``PCSR`` class (pcsr.so) is driven through
oracle/_ref/libref_shim.so, and ``PCSRGraph`` / ``DynamicGraph`` / the compiler stack / TGCN run
unmodified on top of it (tests/golden/ref_harness.py).  Usage:  python tests/golden/make_golden_pcsr.py

  pcsr_streams.npz  raw update streams (s0..): per step the add / delete lists handed to
                    ``edge_update_list(..., is_reverse_edge=True)``, then ``label_edges``; recorded after every
                    step: ``build_csr`` and ``build_reverse_csr`` arrays, degree counters, edge_count and the
                    PMA internals (N, H, logN, items, nodes).  Only streams on which the reference stays
                    self-consistent are kept (defect D13).
  pcsr_gcn.npz      GCNConv on a PCSRGraph at t=0, F in {7, 16, 64}, with and without edge weights
                    (1-based eids, rows walked in the PMA's back-to-front order): out + grad_x
  pcsr_tgcn.npz     PCSRGraph with 5 snapshots, TGCN BPTT in windows of 2 (cache/restore at the window
                    boundary): the CSR arrays of every forward and backward step, hidden states,
                    window losses and parameter gradients
"""
from __future__ import annotations

import numpy as np
import torch

import make_golden as mg           # loads the reference through the harness
import ref_harness as rh
from oracle import ref_shim
from stgraph.graph import PCSRGraph  # noqa: E402

assert ref_shim.available(), "needs oracle/_ref/libref_shim.so (make -C oracle ref)"
KEYS = ("row_offset", "column_indices", "eids", "node_ids")


def visible_edges(p) -> set:
    st = p.state()
    out = set()
    for i, (b, e, _, _) in enumerate(st["nodes"]):
        for j in range(int(b) + 1, int(e)):
            if st["items"][j, 1] != 0:
                out.add((i, int(st["items"][j, 0])))
    return out


def gen_streams():
    d = {}
    specs = [  # (tag, n, universe size, steps, seed)
        ("s0", 5, 12, 4, 1), ("s1", 3, 9, 4, 2), ("s2", 48, 400, 8, 3), ("s3", 300, 3000, 5, 4), ("s4", 2, 4, 6, 5),
    ]
    for tag, n, usize, steps, seed in specs:
        while True:
            rng = np.random.default_rng(seed)
            uni = [(a, b) for a in range(n) for b in range(n)]
            rng.shuffle(uni)
            uni = [tuple(map(int, x)) for x in uni[:usize]]
            p = ref_shim.RefPCSR(n, len(uni))
            cur, rec, ok = set(), {}, True
            for s in range(steps):
                cand = [e for e in uni if e not in cur]
                add = [cand[i] for i in rng.permutation(len(cand))[: int(rng.integers(0, len(cand) + 1))]]
                if s == 0 and not add:
                    add = cand[:1]
                cl = sorted(cur)
                dele = [cl[i] for i in rng.permutation(len(cl))[: int(rng.integers(0, len(cl) + 1))]] if s else []
                add.sort(key=lambda x: (x[1], x[0]))          # as DynamicGraph presorts them
                dele.sort(key=lambda x: (x[1], x[0]))
                p.edge_update_list(add, False, True)
                p.edge_update_list(dele, True, True)
                p.label_edges()
                cur |= set(add)
                cur -= set(dele)
                if p.edge_count != len(cur) or visible_edges(p) != {(b, a) for a, b in cur}:
                    ok = False                                  # the reference lost an edge (D13): not a fixture
                    break
                rec[f"{tag}_step{s}_add"] = np.array(add, np.int32).reshape(-1, 2)
                rec[f"{tag}_step{s}_delete"] = np.array(dele, np.int32).reshape(-1, 2)
                for kind, out in (("fwd", p.build_csr()), ("bwd", p.build_reverse_csr())):
                    for k in KEYS:
                        rec[f"{tag}_step{s}_{kind}_{k}"] = out[k].astype(np.int32)
                ind, outd = p.degrees()
                rec[f"{tag}_step{s}_in_degrees"], rec[f"{tag}_step{s}_out_degrees"] = ind, outd
                st = p.state()
                rec[f"{tag}_step{s}_dims"] = np.array([st["N"], st["H"], st["logN"]], np.int32)
                rec[f"{tag}_step{s}_items"], rec[f"{tag}_step{s}_nodes"] = st["items"], st["nodes"]
            if ok:
                break
            seed += 100
        d.update(rec)
        d[f"{tag}_num_nodes"], d[f"{tag}_max_edges"], d[f"{tag}_steps"] = n, len(uni), steps
    d["tags"] = np.array([s[0] for s in specs])
    mg.save("pcsr_streams.npz", d)


def snapshots(rng, n, e0, T, churn, isolated):
    cur = set(mg.random_edges(rng, n, e0, hub=1, isolated=(isolated,)))
    snaps = []
    for t in range(T):
        if t > 0:
            lst = sorted(cur)
            rng.shuffle(lst)
            for p in lst[: int(e0 * churn)]:
                cur.discard(tuple(p))
            while len(cur) < e0:
                s, dd = int(rng.integers(n)), int(rng.integers(n))
                if s != isolated and dd != isolated:
                    cur.add((s, dd))
        lst = [tuple(map(int, p)) for p in cur]
        rng.shuffle(lst)
        snaps.append([(int(a), int(b)) for a, b in lst])
    return snaps


def device_arrays(G, n, side):
    e = G._forward_graph.edge_count
    get = lambda p, k: np.array(mg.get_array(p, k), np.int32)  # noqa: E731
    pre = "bwd" if side == "bwd" else "fwd"
    return dict(row_offset=get(getattr(G, pre + "_row_offset_ptr"), n + 1),
                column_indices=get(getattr(G, pre + "_column_indices_ptr"), e),
                eids=get(getattr(G, pre + "_eids_ptr"), e), node_ids=get(getattr(G, pre + "_node_ids_ptr"), n))


def gen_gcn():
    n, e = 24, 120
    rng = np.random.default_rng(21)
    el = mg.random_edges(rng, n, e, hub=3, isolated=(9,))
    given = np.array(el, np.int32)
    d = dict(num_nodes=n, src=given[:, 0], dst=given[:, 1])
    w_eid = torch.from_numpy(rng.uniform(0.5, 1.5, (e, 1)).astype(np.float32))
    d["edge_weight_by_eid"] = w_eid              # indexed by eid-1 = rank in (dst, src) order
    for F in (7, 16, 64):
        for use_ew in (False, True):
            G = PCSRGraph([list(el)], n)         # a fresh store per case: its arrays are single-buffered
            assert G.graph_type() == "pcsr"
            G.get_graph(0)
            norm = mg.norm_of(G)
            G.set_ndata("norm", norm)
            d["norm"] = norm
            for k, v in device_arrays(G, n, "fwd").items():
                d[f"fwd_{k}"] = v
            torch.manual_seed(5000 + F)
            conv = mg.GCNConv(F, F, bias=False)
            with torch.no_grad():
                conv.weight.copy_(torch.eye(F))
            x = torch.randn(n, F, requires_grad=True)
            R = torch.randn(n, F)
            out = conv(G, x, edge_weight=w_eid if use_ew else None)
            (out * R).sum().backward()
            for k, v in device_arrays(G, n, "bwd").items():
                d[f"bwd_{k}"] = v
            tag = f"F{F}_{'ew' if use_ew else 'now'}"
            d[tag + "_x"], d[tag + "_R"] = x.detach(), R
            d[tag + "_out"], d[tag + "_grad_x"] = out.detach(), x.grad.detach()
    mg.save("pcsr_gcn.npz", d)


def gen_tgcn():
    n, e0, fin, hid, T, B = 40, 200, 8, 16, 5, 2
    rng = np.random.default_rng(22)
    snaps = snapshots(rng, n, e0, T, 0.1, 11)
    d = dict(num_nodes=n, T=T, B=B)
    for t in range(T):
        arr = np.array(snaps[t], np.int32)
        d[f"t{t}_src"], d[f"t{t}_dst"] = arr[:, 0], arr[:, 1]
    G = PCSRGraph([list(s) for s in snaps], n)
    d["max_num_edges"] = G.max_num_edges
    feats = torch.from_numpy(rng.standard_normal((T, n, fin)).astype(np.float32))
    targets = torch.from_numpy(rng.standard_normal((T, n, 1)).astype(np.float32))
    d["feats"], d["targets"] = feats, targets
    torch.manual_seed(6000)
    model = mg.RefTGCNModel(fin, hid, 1)
    d.update(mg.params_dict(model, "param_"))
    G.reset_graph()
    hs, costs = [], []
    for w0 in range(0, T, B):
        model.zero_grad()
        hidden, cost = None, 0
        ts = list(range(w0, min(w0 + B, T)))
        G.get_graph(w0)                      # dynamic-temporal-tgcn/seastar/train.py:194
        for t in ts:
            G.get_graph(t)
            if G.get_ndata("norm") is None:
                G.set_ndata("norm", mg.norm_of(G))
            d[f"t{t}_norm"] = G.get_ndata("norm")
            d[f"t{t}_in_degrees"] = np.asarray(G.in_degrees())
            d[f"t{t}_num_edges"] = G.get_num_edges()
            for k, v in device_arrays(G, n, "fwd").items():
                d[f"t{t}_fwd_{k}"] = v
            y, hidden = model(G, feats[t], None, hidden)
            cost = cost + torch.mean((y - targets[t]) ** 2)
            hs.append(hidden.detach().clone())
        cost = cost / (B + 1)
        rh._RefPCSR.BUILD_LOG.clear()
        cost.backward()
        # every reverse build of the backward walk, newest timestamp first
        rev = [a for kind, a in rh._RefPCSR.BUILD_LOG if kind == "bwd"]
        assert len(rev) == len(ts), (len(rev), ts)
        for t, a in zip(reversed(ts), rev):
            for k in KEYS:
                d[f"t{t}_bwd_{k}"] = a[k].astype(np.int32)
        assert G.current_timestamp == w0
        costs.append(cost.detach().clone())
        d.update({f"w{w0}_grad_{k}": p.grad.detach().clone() for k, p in model.named_parameters()})
    d["hidden"], d["cost"] = torch.stack(hs), torch.stack(costs)
    mg.save("pcsr_tgcn.npz", d)


if __name__ == "__main__":
    gen_streams()
    gen_gcn()
    gen_tgcn()
